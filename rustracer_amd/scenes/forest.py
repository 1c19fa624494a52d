""""instances" workload: one mesh (an icosphere of 20 * 4^level triangles with per-vertex normals) placed n_side^2 times with rotations and non-uniform
scales over a ground quad, under a distant and a point light. `two_level=True` keeps the reference's form (ObjectBegin / ObjectInstance: one tree per
object, every placement a TransformedPrimitive, rc/primitive.rs:79-118); False writes every placement out as world-space triangles."""
from __future__ import annotations

import numpy as np

from ..scene_desc import SceneDesc
from .procedural import icosphere


def forest(n_side: int = 100, level: int = 3, spp: int = 64, two_level: bool = True, res=(1280, 720)) -> SceneDesc:
    s = SceneDesc()
    s.name = f"instances-{n_side * n_side}"
    ground, leaf = s.matte((0.45, 0.4, 0.3)), s.plastic((0.2, 0.5, 0.25), (0.2, 0.2, 0.2), 0.3)
    e = float(n_side)
    s.add_quad((-e, 0, -e), (-e, 0, e), (e, 0, e), (e, 0, -e), ground)
    P, F = icosphere(level, (0, 0, 0), 0.5)
    n = (P / np.float32(0.5)).astype(np.float32)
    rng = np.random.default_rng(3)
    obj = s.add_object([dict(P=P, idx=F, material=leaf, N=n)]) if two_level else None
    for i in range(n_side):
        for j in range(n_side):
            a, sc = rng.uniform(0, 2 * np.pi), rng.uniform(0.6, 1.4, 3)
            m = np.eye(4)
            c, sn = np.cos(a), np.sin(a)
            m[:3, :3] = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]]) @ np.diag(sc)
            m[:3, 3] = (2.0 * (i - n_side / 2) + rng.uniform(-0.4, 0.4), 0.5 * sc[1], 2.0 * (j - n_side / 2) + rng.uniform(-0.4, 0.4))
            m = m.astype(np.float32)
            if two_level:
                s.add_instance(obj, m)
            else:
                m64 = m.astype(np.float64)
                s.add_mesh((P @ m64[:3, :3].T + m64[:3, 3]).astype(np.float32), F, leaf, N=(n @ np.linalg.inv(m64[:3, :3])).astype(np.float32))
    s.distant_light((0, 0, 0), (0.3, 1.0, -0.4), (3.0, 2.8, 2.5))
    s.point_light((0.0, 12.0, 0.0), (400.0, 400.0, 420.0))
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 9.0, -1.1 * e), (0.0, 0.5, 0.0), 40.0
    s.film.xres, s.film.yres = res
    s.sampler.spp = spp
    return s
