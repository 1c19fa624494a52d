"""The reference's own sampler stream (RT_FLAG_REF_STREAM: one PCG32 per 16 x 16 tile, consumed in order, renderer.rs:83-84) on random rooms - the scenes of
scripts/fuzz_shading.py at 48 x 36 x 8 spp = nine tiles - against the oracle's SAMPLER_REF frame. The stream is shared by a tile: one draw more or less (a path that ends a
bounce early, a last-bit difference at a Russian-roulette threshold) and every later sample of the tile is another sample of the same image, so the comparison is per TILE: a
tile whose samples all took the same draws is inside 1e-3 of the oracle's, and most tiles of most scenes are. Prints, per scene, the tiles inside the gate and the film's
rel-L2; filter weights have to be equal wherever every tile took the same draws. GPU box: python scripts/fuzz_ref_stream.py [n_scenes=40] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rustracer_amd import host  # noqa: E402
from oracle import orc  # noqa: E402
import fuzz_shading as F  # noqa: E402


def main():
    orc.build()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad_weights, tiles_in, tiles_all, whole, refused = 0, 0, 0, 0, 0
    for k in range(n):
        d = F.make_scene(rng)[0]
        try:
            fo, so = orc.OracleScene(d).render(mode=0)
            fh, sh = host.HostScene(d).render(ref_stream=True)
        except (host.BackendError, RuntimeError) as e:
            refused += 1   # (the mode takes scenes of plain triangles: quadrics, masks and instances are refused by name)
            if "plain triangles" not in str(e):
                print(f"scene {k:3d}: refused: {e}"); bad_weights += 1
            continue
        w_ok = np.array_equal(fo[..., 3], fh[..., 3]) if (d.film.filter_kind == 0 and max(d.film.filter_params[:2]) <= 0.5) else np.allclose(fo[..., 3], fh[..., 3], rtol=1e-4, atol=1e-3)
        a, b = host.film_to_rgb(fh).astype(np.float64), orc.film_to_rgb(fo).astype(np.float64)
        H, W = a.shape[:2]
        inside = []
        for y0 in range(0, H, 16):
            for x0 in range(0, W, 16):
                ta, tb = a[y0:y0 + 16, x0:x0 + 16], b[y0:y0 + 16, x0:x0 + 16]
                inside.append(np.linalg.norm(ta - tb) <= 1e-3 * max(np.linalg.norm(tb), 1e-30))
        # (under a wide filter a tile whose stream went another way has other film positions, so other weights)
        bad_weights += int(not w_ok and all(inside)); tiles_in += sum(inside); tiles_all += len(inside); whole += int(all(inside))
        print(f"scene {k:3d}: weights {'equal' if w_ok else 'DIFFER'}, {sum(inside)} of {len(inside)} tiles inside 1e-3, film rel-L2 {F.rel_l2(a, b):.1e}, rays {sh['rays_closest']} / {so['rays_closest']}", flush=True)
    print(f"{n} scenes, {refused} with quadrics or masks (not taken by this mode): weights differ in {bad_weights}; {tiles_in} of {tiles_all} tiles inside 1e-3; {whole} scenes with every tile inside")
    return 1 if bad_weights else 0


if __name__ == "__main__":
    sys.exit(main())
