"""The device's linear BVH builder (rt_bvh_build) on random scenes: random soups of every size class (some degenerate, leaves of 1 / 2 / 4 / 8) and the rooms of
scripts/fuzz_shading.py that hold plain triangles - a valid tree in the reference's flattened layout, the root box of the host's SAH tree, and the same closest hits (the
bits of t, the source triangle, b0) and occlusion answers as the SAH tree, production and counting kernels (tests/test_gpu_bvh_build.py _same_hits). Soups with
zero-area / coincident triangles are counted apart: there the reference's own hits depend on the tree.
GPU box: python scripts/fuzz_device_bvh.py [n_scenes=60] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from rustracer_amd import host  # noqa: E402
from rustracer_amd.scenes import random_soup  # noqa: E402
import fuzz_shading  # noqa: E402
from test_gpu_bvh_build import _same_hits  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad, skipped = 0, 0
    for k in range(n):
        if k % 2 == 0:
            n_tris = int(rng.choice([3, 17, 100, 600, 1300, 5000, 40000]))
            degenerate = bool(rng.random() < 0.2)
            d = random_soup(n_tris, seed=int(rng.integers(1 << 30)), max_prims=int(rng.choice([1, 2, 4, 8])), degenerate=degenerate)
            what = f"soup of {n_tris}"
        else:
            d = fuzz_shading.make_scene(rng)[0]; degenerate = False
            what = f"room of {d.n_tris}"
            if d.spheres or (d.alpha_ids() is not None and np.asarray(d.alpha_ids()).max(initial=-1) >= 0):
                skipped += 1
                continue
        try:
            if degenerate:
                # zero-area and coincident triangles: Bounds3::intersect_p_fast carries no (1 + 2 gamma_3) (bounds.rs:127-157, kept), so whether a flat box is entered at all
                # depends on the box - on the tree. Counted, not required to be zero: the ORACLE's own hits differ between two SAH trees of such a soup (5000 triangles, leaves <= 1
                # against <= 4: 21 of 30 000 rays end on another t; a soup without degenerate triangles: 0), and a linear tree is further from either than they are from each other.
                from util import bits, random_rays
                sah, lin = host.HostScene(d), host.HostScene(d, device_bvh=True)
                lo, hi = sah.bvh()["bounds"][0, :3], sah.bvh()["bounds"][0, 3:]
                assert np.array_equal(lin.bvh()["bounds"][0], sah.bvh()["bounds"][0])
                rays = random_rays(30000, lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), int(rng.integers(1 << 30)))
                a, b = sah.trace(rays, count=False), lin.trace(rays, count=False)
                n_diff = int((bits(a["t"]) != bits(b["t"])).sum())
                assert n_diff <= 450, f"{n_diff} of 30000 rays"
                print(f"scene {k:3d}: {what}, leaves <= {d.max_prims_per_node}, degenerate triangles: {n_diff} of 30000 rays end on another t", flush=True)
                continue
            _same_hits(host, d, 30000, int(rng.integers(1 << 30)), ties=True)
            print(f"scene {k:3d}: {what}, leaves <= {d.max_prims_per_node}: ok", flush=True)
        except AssertionError as e:
            bad += 1
            import traceback
            fr = traceback.extract_tb(e.__traceback__)[-1]
            print(f"scene {k:3d}: {what}, leaves <= {d.max_prims_per_node}{', degenerate triangles' if degenerate else ''}: DIFFERENT at {fr.name}:{fr.lineno}: {fr.line} {str(e)[:160]}", flush=True)
        except host.BackendError as e:
            bad += 1
            print(f"scene {k:3d}: {what}: refused: {e}", flush=True)
    print(f"{n} scenes ({skipped} rooms with quadrics or masks left out), {bad} different")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
