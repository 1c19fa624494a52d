"""S1 "cornell": the classic 555-unit Cornell box.

5 wall quads + short block + tall block (5 quads each) + a 130x105 ceiling emitter = 32 triangles,
all matte; 2 DiffuseAreaLights (one per emitter triangle) so the spatial light distribution is
exercised; box filter radius 0.5, 02sequence sampler, PathIntegrator maxdepth 5.
"""
from __future__ import annotations

from ..scene_desc import SceneDesc


def cornell_box(xres: int = 400, yres: int = 400, spp: int = 64, max_depth: int = 5) -> SceneDesc:
    s = SceneDesc()
    s.name = "cornell"
    white = s.matte((0.73, 0.73, 0.73))
    red = s.matte((0.65, 0.05, 0.05))
    green = s.matte((0.12, 0.45, 0.15))
    light_mat = s.matte((0.78, 0.78, 0.78))
    # walls (normals face into the room; matte is two-sided so orientation only matters for the light)
    s.add_quad((552.8, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 559.2), (549.6, 0.0, 559.2), white)        # floor
    s.add_quad((556.0, 548.8, 0.0), (556.0, 548.8, 559.2), (0.0, 548.8, 559.2), (0.0, 548.8, 0.0), white)  # ceiling
    s.add_quad((549.6, 0.0, 559.2), (0.0, 0.0, 559.2), (0.0, 548.8, 559.2), (556.0, 548.8, 559.2), white)  # back
    s.add_quad((0.0, 0.0, 559.2), (0.0, 0.0, 0.0), (0.0, 548.8, 0.0), (0.0, 548.8, 559.2), green)          # right
    s.add_quad((552.8, 0.0, 0.0), (549.6, 0.0, 559.2), (556.0, 548.8, 559.2), (556.0, 548.8, 0.0), red)    # left
    # short block
    sb = [
        ((130.0, 165.0, 65.0), (82.0, 165.0, 225.0), (240.0, 165.0, 272.0), (290.0, 165.0, 114.0)),
        ((290.0, 0.0, 114.0), (290.0, 165.0, 114.0), (240.0, 165.0, 272.0), (240.0, 0.0, 272.0)),
        ((130.0, 0.0, 65.0), (130.0, 165.0, 65.0), (290.0, 165.0, 114.0), (290.0, 0.0, 114.0)),
        ((82.0, 0.0, 225.0), (82.0, 165.0, 225.0), (130.0, 165.0, 65.0), (130.0, 0.0, 65.0)),
        ((240.0, 0.0, 272.0), (240.0, 165.0, 272.0), (82.0, 165.0, 225.0), (82.0, 0.0, 225.0)),
    ]
    for q in sb:
        s.add_quad(*q, white)
    # tall block
    tb = [
        ((423.0, 330.0, 247.0), (265.0, 330.0, 296.0), (314.0, 330.0, 456.0), (472.0, 330.0, 406.0)),
        ((423.0, 0.0, 247.0), (423.0, 330.0, 247.0), (472.0, 330.0, 406.0), (472.0, 0.0, 406.0)),
        ((472.0, 0.0, 406.0), (472.0, 330.0, 406.0), (314.0, 330.0, 456.0), (314.0, 0.0, 456.0)),
        ((314.0, 0.0, 456.0), (314.0, 330.0, 456.0), (265.0, 330.0, 296.0), (265.0, 0.0, 296.0)),
        ((265.0, 0.0, 296.0), (265.0, 330.0, 296.0), (423.0, 330.0, 247.0), (423.0, 0.0, 247.0)),
    ]
    for q in tb:
        s.add_quad(*q, white)
    # emitter just below the ceiling; winding makes the geometric normal point down (-y)
    s.add_quad((343.0, 548.0, 227.0), (343.0, 548.0, 332.0), (213.0, 548.0, 332.0), (213.0, 548.0, 227.0), light_mat,
               emission=(17.0, 12.0, 4.0))
    assert s.n_tris == 32 and len(s.lights) == 2
    s.camera.pos = (278.0, 273.0, -800.0)
    s.camera.look = (278.0, 273.0, 0.0)
    s.camera.up = (0.0, 1.0, 0.0)
    s.camera.fov = 39.3
    s.film.xres, s.film.yres = xres, yres
    s.sampler.spp = spp
    s.integrator.max_depth = max_depth
    return s
