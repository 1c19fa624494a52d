"""Build-time comparison of the two BVH builders on the 1 M-triangle mesh of S2.

    python scripts/bench_bvh_build.py [nu nv]      -> one JSON line

ms_device: HIP-event time of rt_bvh_build's kernels (keys, radix sort, hierarchy, fit, emit); ms_device_call: the whole
rtxh_scene_commit_device_bvh call (gather + 36 B/triangle upload + build + download + flatten); ms_host_sah: rtxh_scene_commit
(the reference's SAH recursion on host threads). nodes_per_ray: closest-hit node visits of the same 2^20 camera-like rays."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import blob_scene

nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
d = blob_scene(nu=nu, nv=nv, xres=64, yres=64, spp=1)
t0 = time.time(); sah = host.HostScene(d); t_sah = time.time() - t0
host.HostScene(d, device_bvh=True)            # warm-up: module load, allocator
t0 = time.time(); lin = host.HostScene(d, device_bvh=True); t_lin = time.time() - t0
rng = np.random.default_rng(1)
b = sah.bvh()["bounds"][0]
c, r = 0.5 * (b[:3] + b[3:]), 0.5 * float(np.linalg.norm(b[3:] - b[:3]))
n = 1 << 20
o = rng.normal(size=(n, 3)); o = c + 3.0 * r * o / np.linalg.norm(o, axis=1, keepdims=True)
tgt = c + 0.6 * r * rng.uniform(-1, 1, (n, 3))
dirs = tgt - o; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32); rays[:, :3] = o; rays[:, 3] = np.inf; rays[:, 4:7] = dirs
a, l = sah.trace(rays), lin.trace(rays)
assert np.array_equal(a["t"].view(np.uint32), l["t"].view(np.uint32))
print(json.dumps({"n_tris": d.n_tris, "ms_device": round(lin.bvh_build_ms, 3), "ms_device_call": round(1e3 * t_lin, 1), "ms_host_sah": round(1e3 * t_sah, 1),
                  "mtris_per_s_device": round(d.n_tris / lin.bvh_build_ms / 1e3, 1), "n_nodes_sah": int(len(sah.bvh()["offset"])), "n_nodes_linear": int(len(lin.bvh()["offset"])),
                  "nodes_per_ray_sah": round(a["nodes"] / n, 2), "nodes_per_ray_linear": round(l["nodes"] / n, 2), "tris_per_ray_sah": round(a["tris"] / n, 2),
                  "tris_per_ray_linear": round(l["tris"] / n, 2), "hits_equal": True}))
