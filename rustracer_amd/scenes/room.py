"""S4 "room-env" (BASELINE config 5, living-room-class): a box room lit through a window opening by an
InfiniteAreaLight with a procedural lat-long HDR (sun + sky), imagemap-textured matte/plastic/uber/substrate
surfaces (EWA filtered, power-of-two procedural images standing in for decoded PNGs), a glass pane, a mirror,
subdivided noise-displaced furniture blocks. `detail` scales the triangle count: 5 blocks x 6 faces x (4 detail)^2 x 2 triangles
(detail=1: 504, detail=8: 61 K, the default 14.5: 201,864 - the ~200 K of SURVEY.md §8d)."""
from __future__ import annotations

import numpy as np

from ..scene_desc import SceneDesc, WRAP_REPEAT
from .procedural import box_mesh, checker_fbm_image, sky_image


def room_env(xres: int = 1920, yres: int = 1080, spp: int = 1024, detail: float = 14.5, tex_size: int = 1024, env_size: int = 2048) -> SceneDesc:
    s = SceneDesc()
    s.name = "room-env"
    imgs = [s.add_mip(checker_fbm_image(tex_size, 42 + i, c0, c1, cells), max_aniso=8.0, wrap=WRAP_REPEAT)
            for i, (c0, c1, cells) in enumerate([((0.75, 0.6, 0.45), (0.5, 0.35, 0.25), 8), ((0.8, 0.8, 0.78), (0.7, 0.7, 0.72), 4),
                                                 ((0.7, 0.2, 0.15), (0.85, 0.8, 0.7), 16), ((0.2, 0.35, 0.6), (0.8, 0.8, 0.85), 8),
                                                 ((0.9, 0.9, 0.9), (0.1, 0.1, 0.1), 32), ((0.3, 0.5, 0.3), (0.6, 0.7, 0.5), 8)])]
    t = [s.image_tex(m, su=su, sv=sv) for m, (su, sv) in zip(imgs, [(4, 4), (2, 2), (3, 3), (2, 2), (1, 1), (2, 2)])]
    floor = s.plastic(t[0], (0.15, 0.15, 0.15), 0.15)
    walls = s.matte(t[1])
    rug = s.matte(t[2], sigma=25.0)
    sofa = s.uber(kd=t[3], ks=(0.2, 0.2, 0.2), roughness=0.3)
    table = s.substrate(kd=t[5], ks=(0.06, 0.06, 0.06), urough=0.05, vrough=0.08)
    rough_lamp = s.translucent(kd=(0.6, 0.55, 0.4), ks=(0.1, 0.1, 0.1), reflect=0.4, transmit=0.6, roughness=0.3)
    glass = s.glass(index=1.5)
    mirror = s.mirror(0.9)
    mixm = s.mix(s.matte(s.scale_tex(t[4], s.const_tex((0.9, 0.7, 0.4)))), s.metal(roughness=0.1), s.mix_tex(s.const_tex(0.2), s.const_tex(0.8), s.const_tex(0.5)))
    W, H, D = 8.0, 3.2, 6.0

    def quad(p0, p1, p2, p3, m, uv=None, **kw):
        uvs = [(0, 0), (1, 0), (1, 1), (0, 1)] if uv is None else uv
        s.add_mesh([p0, p1, p2, p3], [[0, 1, 2], [0, 2, 3]], m, UV=uvs, **kw)
    quad((0, 0, 0), (W, 0, 0), (W, 0, D), (0, 0, D), floor)
    quad((0, H, 0), (0, H, D), (W, H, D), (W, H, 0), walls)
    quad((0, 0, D), (W, 0, D), (W, H, D), (0, H, D), walls)            # back wall
    quad((0, 0, 0), (0, 0, D), (0, H, D), (0, H, 0), walls)            # left wall
    quad((0, 0, 0), (0, H, 0), (W, H, 0), (W, 0, 0), walls)            # wall behind the camera
    # right wall with a window opening (x = W): four strips around [1.0, 2.6] x [1.5, 4.5]
    y0, y1, z0, z1 = 1.0, 2.6, 1.5, 4.5
    quad((W, 0, 0), (W, H, 0), (W, H, z0), (W, 0, z0), walls)
    quad((W, 0, z1), (W, H, z1), (W, H, D), (W, 0, D), walls)
    quad((W, 0, z0), (W, y0, z0), (W, y0, z1), (W, 0, z1), walls)
    quad((W, y1, z0), (W, H, z0), (W, H, z1), (W, y1, z1), walls)
    quad((W - 0.02, y0, z0), (W - 0.02, y1, z0), (W - 0.02, y1, z1), (W - 0.02, y0, z1), glass)  # pane
    quad((0.01, 0.8, 2.0), (0.01, 0.8, 4.0), (0.01, 2.4, 4.0), (0.01, 2.4, 2.0), mirror)
    quad((2.0, 0.005, 1.5), (6.0, 0.005, 1.5), (6.0, 0.005, 4.5), (2.0, 0.005, 4.5), rug)
    sub = max(1, int(round(4 * detail)))
    for lo, hi, m, amp, sd in [((1.0, 0.0, 4.6), (5.0, 0.9, 5.8), sofa, 0.04, 1), ((1.0, 0.9, 5.4), (5.0, 1.6, 5.8), sofa, 0.05, 2),
                               ((3.0, 0.0, 2.4), (5.0, 0.45, 3.6), table, 0.0, 3), ((6.4, 0.0, 4.8), (7.2, 1.8, 5.6), mixm, 0.02, 4),
                               ((3.7, 0.45, 2.8), (4.3, 0.9, 3.2), rough_lamp, 0.03, 5)]:
        P, I, UV = box_mesh(lo, hi, sub, amp, seed=100 + sd)
        s.add_mesh(P, I, m, UV=UV)
    sun_cos = min(0.9995, float(np.cos(1.5 * 2 * np.pi / env_size)))  # at least ~a texel of sun at small test sizes
    sun_rad = 9.0 / (2 * np.pi * (1 - sun_cos))                       # constant sun irradiance ~9
    env = s.add_mip(sky_image(env_size, env_size // 2, (0.8, -0.25, 0.5), sun_rad, sun_cos), trilinear=False, max_aniso=0.0, wrap=WRAP_REPEAT)
    # light-to-world: the map's +z (theta = 0) points up (+y in world)
    l2w = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    s.infinite_light(env, l2w)
    s.point_light((4.0, 2.9, 3.0), (2.0, 1.8, 1.6))
    s.distant_light((0.0, 0.0, 0.0), (0.2, 1.0, 0.1), (0.15, 0.15, 0.2))
    s.camera.pos = (1.2, 1.6, 0.4)
    s.camera.look = (5.5, 1.1, 4.2)
    s.camera.fov = 62.0
    s.film.xres, s.film.yres = xres, yres
    s.sampler.spp = spp
    return s
