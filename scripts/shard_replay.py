"""The static film split of an N-GPU run, replayed on ONE GPU at the configuration's full size: every rank's shard of the frame (rt_shard{r, N}: the bands
t = r mod N of RT_SHARD_ROWS(H, N) sample rows - what rank r of `bench.py --gpus N` renders before the gather) is rendered and timed by itself. Prints per-rank
ms, per-rank ms / mean (the imbalance the 8-GPU frame will see: the slowest rank sets the frame time) and sum / whole (per-rank fixed cost), and writes the table
as JSON. No scaling claim: one device, one shard at a time. Usage (GPU box): python scripts/shard_replay.py <cornell|blob|mis|room> [world=8] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rustracer_amd import host
from rustracer_amd.distributed import shard_rows
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "room"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join("gpurun_out", f"shard_replay_{name}_{W}.json")
d, workload = bench.make_desc(name, bench.SCENE_SPP[name])
h = host.HostScene(d); h.upload(0)
st0 = h.setup()
cr, sb = st0["cropped"], st0["sample_bounds"]
film = torch.zeros((int(cr[3] - cr[1]), int(cr[2] - cr[0]), 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream


def timed(rank, world, reps):
    h.render(rank=rank, world_size=world, device_out=film, stream=stream); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        _, st = h.render(rank=rank, world_size=world, device_out=film, stream=stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, st["camera_rays"]


whole, n_whole = timed(0, 1, 1)
rows = [timed(r, W, 2) for r in range(W)]
ms = np.array([r[0] for r in rows])
res = {"workload": workload, "world": W, "band_rows": shard_rows(int(sb[3] - sb[1]), W), "whole_frame_ms": round(whole, 2), "per_rank_ms": [round(float(x), 2) for x in ms],
       "per_rank_over_mean": [round(float(x / ms.mean()), 4) for x in ms], "max_over_mean": round(float(ms.max() / ms.mean()), 4),
       "sum_over_whole": round(float(ms.sum() / whole), 4), "ideal_speedup_bound": round(float(whole / ms.max()), 3),
       "camera_samples": [int(r[1]) for r in rows], "camera_samples_whole": int(n_whole), "kernel_source_sha": bench.source_sha(),
       "note": "one GPU, one shard at a time: the split's imbalance and per-rank fixed cost, not a scaling measurement (the gather is not in it)"}
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res))
