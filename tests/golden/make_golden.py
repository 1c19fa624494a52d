"""Regenerates the golden fixtures from the oracle (run from the repo root: python tests/golden/make_golden.py).

The reference holds no expected pixel, hit record or sampler value for this path, so
these vectors are produced by the pinned oracle (tests/test_oracle_kat.py) and replayed against the
GPU on the GPU box, where neither the reference nor a rebuild of the oracle is required.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc  # noqa: E402
from rustracer_amd.scenes import cornell_box  # noqa: E402
from util import random_rays  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    d = cornell_box(32, 32, 16)
    o = orc.OracleScene(d)
    film, st = o.render(mode=1, n_threads=1)
    bvh = o.bvh()
    rays = random_rays(4096, [0, 0, 0], [555, 548, 559], seed=7)
    hc = o.trace(rays)
    rays_any = rays.copy()
    rays_any[:, 3] = np.random.default_rng(8).uniform(50, 600, rays.shape[0]).astype(np.float32)
    ha = o.trace(rays_any, any_hit=True)
    ld = o.light_distrib(max_voxels=512)
    np.savez_compressed(
        os.path.join(OUT, "cornell_32x32_16spp.npz"),
        film_xyzw=film, stats=np.array([st[k] for k in ("camera_rays", "rays_closest", "rays_shadow", "rays_mis")], np.int64),
        bvh_bounds=bvh["bounds"], bvh_offset=bvh["offset"], bvh_n_prims=bvh["n_prims"], bvh_axis=bvh["axis"], bvh_ordered=bvh["ordered"],
        rays=rays, hit_t=hc["t"], hit_prim=hc["prim"], hit_b0=hc["b0"], hit_b1=hc["b1"], hit_nodes=hc["nodes"], hit_tris=hc["tris"],
        rays_any=rays_any, occluded=ha["occluded"], any_nodes=ha["nodes"], any_tris=ha["tris"],
        ld_n_voxels=ld["n_voxels"], ld_func=ld["func"], ld_cdf=ld["cdf"], ld_int=ld["func_int"])
    tabs = {}
    for spp in (16, 64):
        for px in (0, 1, 777):
            t1, t2, _ = orc.sampler_tables(spp, 4, 1, px)
            tabs[f"t1_{spp}_{px}"] = t1
            tabs[f"t2_{spp}_{px}"] = t2
    np.savez_compressed(os.path.join(OUT, "sampler_tables_keyed.npz"), **tabs)
    # Keyed pixels whose start_pixel stream contains a bounded-draw retry (rc/rng.rs:32-40): ~1e-5 of all pixels.
    # The GPU sampler cuts the stream at fixed positions and must fall back for exactly these.
    np.savez_compressed(os.path.join(OUT, "sampler_retry_pixels.npz"),
                        spp1024=orc.sampler_retry_scan(1024, 4, 0, 300000), spp16384=orc.sampler_retry_scan(16384, 4, 0, 12000))
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
