"""Procedural building blocks shared by the synthetic scenes: value noise, grids, icospheres, images.
Everything is deterministic from explicit seeds and computed in float32."""
from __future__ import annotations

import numpy as np


def value_noise3(p: np.ndarray, seed: int, octaves: int = 5) -> np.ndarray:
    """Multi-octave value noise on points p (n,3) -> (n,) in roughly [-1, 1]."""
    p = np.asarray(p, np.float64)
    out = np.zeros(p.shape[0])
    amp, freq, norm = 1.0, 1.0, 0.0
    for o in range(octaves):
        q = p * freq
        i0 = np.floor(q).astype(np.int64)
        f = q - i0
        f = f * f * (3 - 2 * f)
        acc = 0.0
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    ix, iy, iz = i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz
                    h = (ix * 73856093) ^ (iy * 19349663) ^ (iz * 83492791) ^ (seed * 2654435761 + o * 97)
                    h = (h ^ (h >> 13)) * 1274126177
                    h = (h ^ (h >> 16)) & 0xffffff
                    v = h / float(0xffffff) * 2 - 1
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    acc = acc + v * w
        out += amp * acc
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return (out / norm).astype(np.float32)


def grid_indices(nu: int, nv: int) -> np.ndarray:
    """Two triangles per cell of an (nv+1) x (nu+1) vertex grid (row-major, u fastest)."""
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).reshape(-1)
    b, c, d = a + 1, a + nu + 1, a + nu + 2
    return np.stack([np.stack([a, b, d], 1), np.stack([a, d, c], 1)], 1).reshape(-1, 3).astype(np.int32)


def displaced_sphere(nu: int, nv: int, center, radius: float, amp: float, seed: int):
    """Lat-long grid sphere with radial value-noise displacement: exactly 2*nu*nv triangles.
    Returns (P, idx, N, UV); N = per-vertex normals from the displaced surface."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.02, 0.98, nv + 1)  # keep clear of the poles: no zero-area triangles
    uu, vv = np.meshgrid(u, v)
    phi, theta = uu * 2 * np.pi, vv * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1).reshape(-1, 3)
    r = radius * (1.0 + amp * value_noise3(d * 2.5 + 11.0, seed).astype(np.float64))
    P = (np.asarray(center, np.float64) + d * r[:, None])
    idx = grid_indices(nu, nv)
    # vertex normals: area-weighted face normals
    fn = np.cross(P[idx[:, 1]] - P[idx[:, 0]], P[idx[:, 2]] - P[idx[:, 0]])
    N = np.zeros_like(P)
    for k in range(3):
        np.add.at(N, idx[:, k], fn)
    N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-30)
    # orient outward
    flip = np.einsum("ij,ij->i", N, d) < 0
    N[flip] *= -1
    UV = np.stack([uu, vv], -1).reshape(-1, 2)
    return P.astype(np.float32), idx, N.astype(np.float32), UV.astype(np.float32)


def icosphere(level: int, center, radius: float):
    """Icosphere with 20 * 4**level triangles, outward winding."""
    t = (1.0 + 5 ** 0.5) / 2.0
    V = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(v, np.float64) / np.linalg.norm(v) for v in V]
    for _ in range(level):
        cache, F2 = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = V[a] + V[b]
                V.append(m / np.linalg.norm(m))
                cache[key] = len(V) - 1
            return cache[key]
        for a, b, c in F:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            F2 += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        F = F2
    P = (np.asarray(center, np.float64) + np.array(V) * radius).astype(np.float32)
    return P, np.array(F, np.int32)


def box_mesh(lo, hi, subdiv: int = 1, noise_amp: float = 0.0, seed: int = 0):
    """Axis-aligned box as 6 subdivided faces (12 * subdiv^2 triangles), optionally noise-displaced along face normals.
    Returns (P, idx, UV)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    Ps, Is, UVs, off = [], [], [], 0
    for axis in range(3):
        for side in (0, 1):
            a, b = (axis + 1) % 3, (axis + 2) % 3
            s = np.linspace(0, 1, subdiv + 1)
            uu, vv = np.meshgrid(s, s)
            p = np.zeros(uu.shape + (3,))
            p[..., axis] = hi[axis] if side else lo[axis]
            p[..., a] = lo[a] + (hi[a] - lo[a]) * (uu if side else vv)
            p[..., b] = lo[b] + (hi[b] - lo[b]) * (vv if side else uu)
            p = p.reshape(-1, 3)
            if noise_amp > 0:
                n = np.zeros(3)
                n[axis] = 1.0 if side else -1.0
                p = p + n * (noise_amp * value_noise3(p * 0.05 + 3.0, seed + axis * 2 + side).astype(np.float64))[:, None]
            Ps.append(p)
            Is.append(grid_indices(subdiv, subdiv) + off)
            UVs.append(np.stack([uu, vv], -1).reshape(-1, 2))
            off += p.shape[0]
    return np.concatenate(Ps).astype(np.float32), np.concatenate(Is).astype(np.int32), np.concatenate(UVs).astype(np.float32)


def checker_fbm_image(size: int, seed: int, c0=(0.8, 0.8, 0.8), c1=(0.2, 0.2, 0.25), cells: int = 8) -> np.ndarray:
    """(size, size, 3) float32 texture: checkerboard modulated by value noise (stands in for a decoded PNG)."""
    y, x = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    chk = ((x * cells // size) + (y * cells // size)) % 2
    p = np.stack([x / size * 6.0, y / size * 6.0, np.full_like(x, 0.5, dtype=np.float64)], -1).reshape(-1, 3)
    n = (0.75 + 0.25 * value_noise3(p, seed, 4).astype(np.float64)).reshape(size, size, 1)
    img = np.where(chk[..., None] == 0, np.array(c0), np.array(c1)) * n
    return np.clip(img, 0.0, None).astype(np.float32)


def sky_image(width: int, height: int, sun_dir=(0.3, 0.8, 0.5), sun_radiance: float = 5e4, sun_cos: float = 0.9995) -> np.ndarray:
    """(height, width, 3) lat-long HDR: sky gradient + sun disc (stands in for a decoded PFM/EXR).
    Texel (u, v): phi = 2 pi u, theta = pi v, direction (sin t cos p, sin t sin p, cos t) as InfiniteAreaLight uses."""
    v, u = np.meshgrid((np.arange(height) + 0.5) / height, (np.arange(width) + 0.5) / width, indexing="ij")
    theta, phi = v * np.pi, u * 2 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], -1)
    s = np.asarray(sun_dir, np.float64)
    s = s / np.linalg.norm(s)
    up = np.clip(d[..., 2], -1, 1)
    sky = np.where(up[..., None] > 0, (1 - up[..., None]) * np.array([0.9, 0.95, 1.0]) + up[..., None] * np.array([0.25, 0.45, 0.9]), np.array([0.12, 0.11, 0.1]))
    img = sky * 1.5
    img = img + (np.einsum("ijk,k->ij", d, s) > sun_cos)[..., None] * sun_radiance * np.array([1.0, 0.95, 0.85])
    return img.astype(np.float32)
