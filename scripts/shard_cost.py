"""Where a shard's time goes beyond its share of the frame (GPU box): one rank of an N-way split of a bench scene rendered with and without per-stage timing;
prints wall ms, the stage sums and what is left (per-frame fixed cost: table builds that nothing overlaps, launch gaps, memsets, the stats copy).
Usage: python scripts/shard_cost.py <scene> [world=8] [rank=3]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rustracer_amd import host
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d, _ = bench.make_desc(name, bench.SCENE_SPP[name])
h = host.HostScene(d); h.upload(0)
st0 = h.setup(); cr = st0["cropped"]
film = torch.zeros((int(cr[3] - cr[1]), int(cr[2] - cr[0]), 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for rank, world in ((0, 1), (R, W)):
    for tk in (False, True):
        h.render(rank=rank, world_size=world, device_out=film, stream=stream, time_kernels=tk); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            _, st = h.render(rank=rank, world_size=world, device_out=film, stream=stream, time_kernels=tk)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / 3 * 1e3
        keys = ("ms_sampler", "ms_raygen", "ms_trace_closest", "ms_trace_any", "ms_trace_mis", "ms_shade", "ms_resolve", "ms_film")
        print(f"{name} rank {rank}/{world} time_kernels={int(tk)} wall {wall:8.2f} ms  ms_total {st['ms_total']:8.2f}  " + " ".join(f"{k[3:]} {st[k]:.2f}" for k in keys) +
              (f"  | path stages {sum(st[k] for k in keys[1:]):.2f}" if tk else ""), flush=True)
