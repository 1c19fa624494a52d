"""Random triangle soups for BVH / traversal parity tests (not a BASELINE config)."""
from __future__ import annotations

import numpy as np

from ..scene_desc import SceneDesc


def random_soup(n_tris: int, seed: int = 0, max_prims: int = 4, degenerate: bool = False, extent: float = 100.0) -> SceneDesc:
    rng = np.random.default_rng(seed)
    s = SceneDesc()
    s.name = f"soup{n_tris}"
    m = s.matte((0.6, 0.6, 0.6))
    c = rng.uniform(0, extent, (n_tris, 1, 3))
    size = extent * 0.5 / max(n_tris, 1) ** (1 / 3)
    P = (c + rng.normal(0, size, (n_tris, 3, 3))).astype(np.float32)
    if degenerate:  # coincident centroids / zero-extent centroid bounds exercise the leaf fall-backs (bvh/mod.rs:172-179)
        P[: n_tris // 2] = P[0]
        P[n_tris // 2:, :, 1] = 7.0
    idx = np.arange(3 * n_tris, dtype=np.int32).reshape(-1, 3)
    s.add_mesh(P.reshape(-1, 3), idx, m)
    s.add_quad((0, extent * 1.2, 0), (0, extent * 1.2, extent), (extent, extent * 1.2, extent), (extent, extent * 1.2, 0), m, emission=(5, 5, 5))
    s.max_prims_per_node = max_prims
    s.camera.pos = (extent / 2, extent / 2, -extent * 1.5)
    s.camera.look = (extent / 2, extent / 2, extent / 2)
    s.camera.fov = 45.0
    s.film.xres, s.film.yres = 32, 32
    s.sampler.spp = 4
    return s
