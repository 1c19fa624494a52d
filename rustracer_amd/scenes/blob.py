"""S2 "blob-1M" (BASELINE config 3, ganesha-class): a noise-displaced sphere of exactly 2*nu*nv triangles
(1,048,576 at nu=1024, nv=512) with per-vertex normals, matte Kd 0.6, on a ground quad, one quad area light.
Traversal-bound: ~1-2 M BVH nodes (32-64 MB) + 48 MB of triangle records spill every L2."""
from __future__ import annotations

from ..scene_desc import SceneDesc
from .procedural import displaced_sphere


def blob_scene(nu: int = 1024, nv: int = 512, xres: int = 1280, yres: int = 720, spp: int = 256, seed: int = 1234, via_ply: str = None) -> SceneDesc:
    """via_ply: path; the mesh is written there as binary little-endian PLY and read back through the host's PLY reader."""
    s = SceneDesc()
    s.name = f"blob-{2 * nu * nv}"
    body = s.matte((0.6, 0.6, 0.6))
    ground = s.matte((0.4, 0.38, 0.35))
    lm = s.matte((0.5, 0.5, 0.5))
    P, idx, N, UV = displaced_sphere(nu, nv, (0.0, 1.1, 0.0), 1.0, 0.18, seed)
    if via_ply:
        from ..ingest import write_ply
        write_ply(via_ply, P, idx, N, UV)
        s.add_ply(via_ply, body)
    else:
        s.add_mesh(P, idx, body, N=N, UV=UV)
    s.add_quad((-6.0, 0.0, -6.0), (-6.0, 0.0, 6.0), (6.0, 0.0, 6.0), (6.0, 0.0, -6.0), ground)
    # light above, facing down
    s.add_quad((-1.5, 4.5, -1.5), (1.5, 4.5, -1.5), (1.5, 4.5, 1.5), (-1.5, 4.5, 1.5), lm, emission=(18.0, 17.0, 15.0))
    assert s.n_tris == 2 * nu * nv + 4
    s.camera.pos = (0.0, 2.2, -4.2)
    s.camera.look = (0.0, 1.0, 0.0)
    s.camera.fov = 40.0
    s.film.xres, s.film.yres = xres, yres
    s.sampler.spp = spp
    return s
