"""Many instances of one mesh: the reference's two-level form (one tree per object, general kernels) against the same scene written out (single-level fast
kernels). Prints Msamples/s, memory-relevant counts and the film difference. Usage (GPU box): python scripts/exp_instances.py [n_side] [level] [spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import host
from rustracer_amd.scene_desc import SceneDesc
from rustracer_amd.scenes import forest


if __name__ == "__main__":
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    films = {}
    for two_level in (True, False):
        t0 = time.time(); d = forest(n_side, level, spp, two_level); h = host.HostScene(d); t1 = time.time()
        h.render()
        t2 = time.time(); f, st = h.render(); t3 = time.time()
        films[two_level] = f
        ms = (t3 - t2) * 1e3
        print(f"{'two-level' if two_level else 'written out'}: {n_side * n_side} instances x {20 * 4 ** level} triangles, build {t1 - t0:.1f} s, frame {ms:.1f} ms, "
              f"{d.film.xres * d.film.yres * spp / ms / 1e3:.1f} Msamples/s, top-level primitives {len(h.bvh()['ordered'])}", flush=True)
    a, b = host.film_to_rgb(films[True]), host.film_to_rgb(films[False])
    # two roundings of one scene: at a few samples per pixel single paths that flip at a grazing hit dominate the L2 norm, the means agree
    print(f"weights equal {np.array_equal(films[True][..., 3], films[False][..., 3])}, rel L2 two-level vs written out {np.linalg.norm(a - b) / np.linalg.norm(b):.2e}, "
          f"mean radiance {a.mean():.5f} vs {b.mean():.5f}")
