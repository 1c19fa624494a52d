"""Randomised parity of the stackless LDS walks (pruned link tables, general primitives, mid-size scenes) against the oracle: random soups of every size class with
random leaf sizes, some degenerate, some with spheres / a disk / a cylinder among the triangles; rays inside and outside the scene, axis-parallel, with zero and
denormal direction components, origins on vertices; closest hit (primitive and the bits of t, b0, b1) and occlusion. GPU box, repo root:
    python scripts/fuzz_lds_walks.py [n_scenes=40] [seed=1]
Prints one line per scene and the number of mismatching rays in all (expected: 0). The oracle is the checker here, as in tests/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rustracer_amd import host  # noqa: E402
from rustracer_amd.scenes import random_soup  # noqa: E402
from oracle import orc  # noqa: E402  (the checker, as in tests/conftest.py)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rays_for(rng, lo, hi, verts, n):
    ext = hi - lo
    org = rng.uniform(lo - 0.4 * ext, hi + 0.4 * ext, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = np.arange(n)
    d[k % 11 == 0, rng.integers(0, 3)] = 0.0
    d[k % 13 == 0, rng.integers(0, 3)] = -0.0
    d[k % 17 == 0] = np.float32([0, 0, 1])
    d[k % 19 == 0, rng.integers(0, 3)] = 1e-39          # a denormal component: its reciprocal overflows
    sel = k % 7 == 0
    org[sel] = verts[rng.integers(0, len(verts), int(sel.sum()))]   # origins on box bounds
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = org; rays[:, 3] = np.inf; rays[:, 4:7] = d
    return rays


def main():
    orc.build()
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad_total, t0 = 0, time.time()
    for k in range(n_scenes):
        cls = k % 4
        n_tris = int(rng.integers(3, 127)) if cls < 2 else (int(rng.integers(130, 700)) if cls == 2 else int(rng.integers(700, 1400)))
        max_prims = int(rng.choice([1, 2, 4, 8]))
        degenerate = bool(rng.random() < 0.15)
        d = random_soup(n_tris, seed=int(rng.integers(1 << 30)), max_prims=max_prims, degenerate=degenerate)
        general = cls == 1
        if general:                                       # quadrics among the triangles: the GENERAL kernels of an LDS-resident scene
            for _ in range(int(rng.integers(1, 4))):
                d.add_sphere(tuple(rng.uniform(10, 90, 3)), float(rng.uniform(3, 15)), d.matte((0.5, 0.5, 0.5)))
        verts = np.asarray(d.arrays()[0], np.float32)
        lo, hi = verts.min(0), verts.max(0)
        o, h = orc.OracleScene(d), host.HostScene(d)
        h.upload(0)
        nn = h.bvh_sizes()[0]
        kind = "lds" if h.lds_resident() else ("mid" if host.lib().rtxh_scene_query(h.h, 2) == 1 else "hbm")
        rays = rays_for(rng, lo, hi, verts, 40000)
        ro, rh = o.trace(rays), h.trace(rays, count=False)
        bad = int((ro["prim"] != rh["prim"]).sum())
        for f in (("t",) if general else ("t", "b0", "b1")):
            bad += int((bits(ro[f]) != bits(rh[f])).sum())
        rays[:, 3] = rng.uniform(0.05, 1.2, len(rays)).astype(np.float32) * float(np.linalg.norm(hi - lo))
        bad += int((o.trace(rays, True)["occluded"] != h.trace(rays, True, count=False)["occluded"]).sum())
        bad_total += bad
        print(f"scene {k:3d}: {n_tris:5d} triangles{' + quadrics' if general else ''}, leaves <= {max_prims}{', degenerate' if degenerate else ''}, {nn:5d} nodes ({kind}, "
              f"{host.lib().rtxh_scene_query(h.h, 1)} tested), hits {float((ro['prim'] >= 0).mean()):.2f}: {bad} mismatches", flush=True)
    print(f"{n_scenes} scenes, {bad_total} mismatching values, {time.time() - t0:.0f} s")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
