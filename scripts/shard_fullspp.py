"""Measurement on ONE GPU: rank 0's shard of the full S1 frame (1024 x 1024 x 1024 spp) at world sizes 1, 2, 4, 8 - what one rank of `bench.py --gpus N` renders
before the gather. Prints ms per shard and shard x N / whole (1.0 = no per-rank fixed cost)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rustracer_amd import host
from rustracer_amd import scenes as S
d = S.cornell_box(1024, 1024, 1024)
h = host.HostScene(d); h.upload(0)
cr = h.setup()["cropped"]
film = torch.zeros((int(cr[3] - cr[1]), int(cr[2] - cr[0]), 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
whole = None
for W in (1, 2, 4, 8):
    h.render(rank=0, world_size=W, device_out=film, stream=stream); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        h.render(rank=0, world_size=W, device_out=film, stream=stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 3 * 1e3
    whole = whole or ms
    print(f"world {W}: rank 0's shard {ms:8.2f} ms   x {W} / whole = {ms * W / whole:.3f}", flush=True)
