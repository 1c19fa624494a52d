"""S3 "mis-plates" (BASELINE config 4, veach-mis-class): four tilted plates with Trowbridge-Reitz roughness
0.005 / 0.02 / 0.05 / 0.1 (metal and plastic), four emitters of radius 0.03 / 0.1 / 0.3 / 0.9 with equal power - tessellated as
icospheres (1280 emitter triangles: the many-lights workload the BASELINE config is benchmarked on) or, with `analytic_spheres=True`, as four
Shape "sphere" emitters the way veach-mis.pbrt has them - and a dim ceiling light. Exercises microfacet BSDFs, MIS and the
spatial light distribution."""
from __future__ import annotations

import numpy as np

from ..scene_desc import SceneDesc
from .procedural import icosphere


def mis_plates(xres: int = 1280, yres: int = 720, spp: int = 512, sphere_level: int = 2, analytic_spheres: bool = False) -> SceneDesc:
    s = SceneDesc()
    s.name = "mis-plates"
    floor = s.matte((0.4, 0.4, 0.4))
    wall = s.plastic((0.3, 0.32, 0.35), (0.2, 0.2, 0.2), 0.2)
    mats = [s.metal(roughness=0.005), s.plastic((0.05, 0.05, 0.05), (0.8, 0.8, 0.8), 0.02), s.metal(roughness=0.05), s.plastic((0.1, 0.1, 0.1), (0.7, 0.7, 0.7), 0.1)]
    lm = s.matte((0.0, 0.0, 0.0))
    s.add_quad((-12, 0, -6), (-12, 0, 14), (12, 0, 14), (12, 0, -6), floor)
    s.add_quad((-12, 0, 14), (-12, 12, 14), (12, 12, 14), (12, 0, 14), wall)
    # plates: tilted about x, stacked towards the back
    for i, m in enumerate(mats):
        z0, y0 = 1.0 + 2.2 * i, 0.3 + 0.55 * i
        ang = np.radians(12.0 + 9.0 * i)
        dz, dy = 1.9 * np.cos(ang), 1.9 * np.sin(ang)
        s.add_quad((-4.5, y0, z0), (-4.5, y0 + dy, z0 + dz), (4.5, y0 + dy, z0 + dz), (4.5, y0, z0), m)
    radii = [0.03, 0.1, 0.3, 0.9]
    xs = [-3.6, -1.3, 1.0, 3.6]
    n_em = 0
    for r, x in zip(radii, xs):
        power = 800.0  # equal power: L = P / (pi * area), area ~ 4 pi r^2
        L = power / (np.pi * 4 * np.pi * r * r)
        if analytic_spheres:
            s.add_sphere((x, 6.0, 6.0), r, lm, emission=(L, L * 0.9, L * 0.8))
            n_em += 1
            continue
        P, F = icosphere(sphere_level, (x, 6.0, 6.0), r)
        s.add_mesh(P, F, lm, emission=(L, L * 0.9, L * 0.8))
        n_em += F.shape[0]
    s.add_quad((-10, 10.0, 0), (10, 10.0, 0), (10, 10.0, 10), (-10, 10.0, 10), lm, emission=(0.4, 0.4, 0.45))
    s.camera.pos = (0.0, 3.2, -10.5)
    s.camera.look = (0.0, 2.0, 4.0)
    s.camera.fov = 36.0
    s.film.xres, s.film.yres = xres, yres
    s.sampler.spp = spp
    return s
