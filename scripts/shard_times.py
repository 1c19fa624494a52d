"""Time rank 0's shard of the S1 frame at world sizes 1/2/4/8 on ONE GPU (how evenly the interleaved tile rows divide the work)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rustracer_amd import host
from rustracer_amd.scenes import cornell_box
d = cornell_box(1024, 1024, 1024)
h = host.HostScene(d); h.upload(0)
st0 = h.setup(); cr = st0["cropped"]
film = torch.zeros((int(cr[3]-cr[1]), int(cr[2]-cr[0]), 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for world in (1, 2, 4, 8):
    h.render(rank=0, world_size=world, device_out=film, stream=stream); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        _, st = h.render(rank=0, world_size=world, device_out=film, stream=stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
    print(f"world {world}: rank-0 shard {dt*1e3:.1f} ms, ideal {1.0/world:.3f} of the frame", flush=True)
