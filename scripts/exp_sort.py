"""Measurement: how much a closest-hit launch gains from the ORDER of its rays. Second-bounce-like rays (origins on the surfaces camera rays reach, directions over
the hemisphere back toward the viewer side) traced by the production kernels (rt_trace_closest_device) in pixel order, shuffled, and sorted by direction octant,
by origin cell, and by both. Usage (GPU box): python scripts/exp_sort.py [cornell|blob|room ...]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rustracer_amd import host
from rustracer_amd import scenes as S


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    names = sys.argv[1:] or ["cornell", "blob", "room"]
    rng = np.random.default_rng(5)
    for name in names:
        d = {"cornell": lambda: S.cornell_box(1024, 1024, 4), "blob": lambda: S.blob_scene(spp=4), "room": lambda: S.room_env(spp=4), "mis": lambda: S.mis_plates(spp=4)}[name]()
        h = host.HostScene(d)
        W, H = d.film.xres, d.film.yres
        # camera rays through pixel centres, 2 per pixel -> n rays
        cam = np.asarray(d.camera.pos, np.float64); look = np.asarray(d.camera.look, np.float64); up = np.asarray(d.camera.up, np.float64)
        f = look - cam; f /= np.linalg.norm(f); r = np.cross(up / np.linalg.norm(up), f); r /= np.linalg.norm(r); u = np.cross(f, r)
        t = np.tan(np.radians(d.camera.fov) / 2); asp = W / H
        sx, sy = (t * asp, t) if asp > 1 else (t, t / asp)  # fov over the shorter axis
        ys, xs = np.mgrid[0:H, 0:W]
        px = np.tile(((xs.ravel() + 0.5) / W * 2 - 1), 2); py = np.tile((1 - (ys.ravel() + 0.5) / H * 2), 2)
        dirs = f[None, :] + px[:, None] * sx * r[None, :] + py[:, None] * sy * u[None, :]
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        n = len(dirs)
        rays = np.zeros((n, 8), np.float32); rays[:, 0:3] = cam; rays[:, 3] = np.inf; rays[:, 4:7] = dirs
        hit = h.trace(rays, count=False)
        ok = hit["prim"] >= 0
        print(name, "camera rays", n, "hit", ok.mean())
        o2 = (cam[None, :] + hit["t"][:, None].astype(np.float64) * dirs)[ok]
        din = dirs[ok]
        v = rng.normal(size=(len(o2), 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        v[np.sum(v * din, axis=1) > 0] *= -1  # back toward the side the ray came from
        o2 = o2 + 1e-3 * np.abs(o2).max() * v
        m = len(o2)
        lo, hi = o2.min(0), o2.max(0)
        cell = np.clip(((o2 - lo) / (hi - lo + 1e-9) * 1024).astype(np.int64), 0, 1023)
        mort = morton3(cell)
        octant = ((v[:, 0] < 0).astype(np.uint64) | ((v[:, 1] < 0).astype(np.uint64) << 1) | ((v[:, 2] < 0).astype(np.uint64) << 2))
        orders = {
            "pixel order": np.arange(m),
            "shuffled": rng.permutation(m),
            "octant": np.argsort(octant, kind="stable"),
            "octant of shuffled": None,
            "origin cell (morton 30 bit)": np.argsort(mort, kind="stable"),
            "octant + origin cell": np.argsort((octant << np.uint64(30)) | mort, kind="stable"),
            "origin cell (morton 9 bit) + octant": np.argsort(((mort >> np.uint64(21)) << np.uint64(3)) | octant, kind="stable"),
        }
        sh = orders["shuffled"]; orders["octant of shuffled"] = sh[np.argsort(octant[sh], kind="stable")]
        hits = torch.empty((m, 4), dtype=torch.float32, device="cuda")
        base = None
        for label, idx in orders.items():
            planar = np.zeros((2, m, 4), np.float32)
            planar[0, :, 0:3] = o2[idx]; planar[0, :, 3] = np.inf; planar[1, :, 0:3] = v[idx]
            dr = torch.from_numpy(planar).cuda()
            h.trace_device(dr.data_ptr(), m, hits.data_ptr(), reps=2)
            ms = min(h.trace_device(dr.data_ptr(), m, hits.data_ptr(), reps=10) for _ in range(3))
            base = base or ms
            print(f"  {label:38s} {ms:8.3f} ms/launch  {m / ms / 1e6:7.2f} Grays/s  x{base / ms:5.2f}")


main()
