"""One bump-mapped, mirror-sharp plate under an area light, oracle against device: at one and at five bounces: where a lobe 1e-3 rad wide (roughness 0.001, not remapped) stops agreeing - the first vertex agrees, later ones differ like noise, bump map or none (MEASUREMENTS R6)
(GPU box). python scripts/exp_sharp_lobes.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import host
from rustracer_amd import scene_desc as sd
from oracle import orc
orc.build()
rng = np.random.default_rng(3)
img = rng.uniform(0.02, 1.0, (16, 16, 3)).astype(np.float32) ** 2


def scene(mat_kind, bump_kind, rough=0.001, remap=False, spp=64, depth=1, scale=0.05, uv=True):
    s = sd.SceneDesc()
    if mat_kind == "plastic":
        m = s.plastic((0.3, 0.3, 0.3), (0.7, 0.6, 0.3), rough, remap)
    elif mat_kind == "metal":
        m = s.metal(roughness=rough, remap=remap)
    elif mat_kind == "substrate":
        m = s.substrate((0.3, 0.3, 0.3), (0.5, 0.5, 0.5), rough, 0.02, remap)
    else:
        m = s.matte((0.5, 0.5, 0.5))
    if bump_kind == "fbm":
        b = s.fbm_tex(0.5, 4)
    elif bump_kind == "image_tri":
        b = s.image_tex(s.add_mip(img, trilinear=True), 3.0, 3.0)
    elif bump_kind == "image_ewa":
        b = s.image_tex(s.add_mip(img, trilinear=False, max_aniso=8.0), 3.0, 3.0)
    elif bump_kind == "checker":
        b = s.checker_tex(1.0, 0.0, 4.0, 4.0)
    elif bump_kind == "uv":
        b = s.uv_tex(3.0, 2.0)
    else:
        b = None
    if b is not None:
        s.set_bump(m, s.scale_tex(b, s.const_tex(scale)) if scale is not None else b)
    R = 4.0
    s.add_mesh(np.float32([(-R, 0, -R), (-R, 0, R), (R, 0, R), (R, 0, -R)]), [[0, 1, 2], [0, 2, 3]], m, UV=np.float32([(0, 0), (1, 0), (1, 1), (0, 1)]) if uv else None)
    grey = s.matte((0.5, 0.5, 0.5))
    if depth > 1:  # later vertices on the same material: a back wall and a side wall (no ray differentials there: du = dv = 0.0005, point lookups)
        s.add_mesh(np.float32([(-R, 0, R), (-R, 2 * R, R), (R, 2 * R, R), (R, 0, R)]), [[0, 1, 2], [0, 2, 3]], m, UV=np.float32([(0, 0), (1, 0), (1, 1), (0, 1)]) if uv else None)
        s.add_mesh(np.float32([(-R, 0, -R), (-R, 2 * R, -R), (-R, 2 * R, R), (-R, 0, R)]), [[0, 1, 2], [0, 2, 3]], grey)
    s.add_mesh(np.float32([(-2, 5, 0), (2, 5, 0), (2, 5, 3), (-2, 5, 3)]), [[0, 1, 2], [0, 2, 3]], grey, emission=(20.0, 20.0, 20.0), two_sided=True)
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 2.5, -3.7), (0.0, 0.0, 1.0), 60.0
    s.film.xres, s.film.yres = 48, 36
    s.sampler.spp = spp
    s.integrator.max_depth = depth
    return s


def compare(d):
    fo, so = orc.OracleScene(d).render(mode=1); fh, sh = host.HostScene(d).render()
    a, b = host.film_to_rgb(fh).astype(np.float64), orc.film_to_rgb(fo).astype(np.float64)
    off = float((np.abs(a - b).max(axis=-1) > 1e-3 * np.maximum(np.abs(b).max(axis=-1), b.mean())).mean())
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())), off


def main():
    for mat in ("plastic", "metal", "substrate", "matte"):  # (prints one line per case; depth 5 adds two more walls of the material)
        for bump in ("none", "fbm", "image_tri", "image_ewa", "checker", "uv"):
            for rough, remap in ((0.001, False), (0.02, False), (0.1, True)):
                if mat == "matte" and rough != 0.1:
                    continue
                for depth in (1, 5):
                    e, off = compare(scene(mat, bump, rough, remap, depth=depth))
                    print(f"{mat:9s} bump {bump:9s} roughness {rough} remap {remap} depth {depth}: rel-L2 {e:.1e}, pixels off {off:.3f}", flush=True)


if __name__ == "__main__":
    main()
