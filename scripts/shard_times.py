"""How evenly the interleaved tile rows divide a frame, measured on ONE GPU: every chunk of an n-chunk split (rt_shard{c, n}: tile rows t = c mod n, what
rt_multi_render hands out) is rendered and timed by itself; the dynamic chunk queue of rt_multi_render (a device that finishes takes the next chunk) is then
replayed for W devices with k chunks per device. Prints, per (scene, W, k): the slowest device's time, the ideal (whole frame / W) and their ratio.
Usage (GPU box): python scripts/shard_times.py [cornell|room ...]   (room at 64 spp: the split is per pixel row, not per sample)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rustracer_amd import host
from rustracer_amd import scenes as S


def chunk_times(h, film, stream, n):
    out = []
    for c in range(n):
        h.render(rank=c, world_size=n, device_out=film, stream=stream); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2):
            h.render(rank=c, world_size=n, device_out=film, stream=stream)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 2)
    return out


def replay(times, W):  # chunks in order, each to the device that is free first
    free = [0.0] * W
    for t in times:
        k = min(range(W), key=lambda i: free[i]); free[k] += t
    return max(free)


for name in (sys.argv[1:] or ["cornell", "room"]):
    d = S.cornell_box(1024, 1024, 128) if name == "cornell" else S.room_env(spp=64)
    h = host.HostScene(d); h.upload(0)
    cr = h.setup()["cropped"]
    film = torch.zeros((int(cr[3] - cr[1]), int(cr[2] - cr[0]), 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    whole = chunk_times(h, film, stream, 1)[0]
    print(f"{name}: whole frame {whole * 1e3:.1f} ms ({(d.film.yres + 15) // 16} tile rows)", flush=True)
    cache = {}
    for W in (2, 4, 8):
        for k in (1, 2, 4):
            n = W * k
            if n not in cache: cache[n] = chunk_times(h, film, stream, n)
            ts = cache[n]
            span = replay(ts, W)
            print(f"  W={W} k={k}: chunks {min(ts) * 1e3:7.1f} .. {max(ts) * 1e3:7.1f} ms, sum {sum(ts) * 1e3:7.1f} ({sum(ts) / whole:5.3f} of the whole frame), "
                  f"slowest device {span * 1e3:7.1f} ms = {span / (whole / W):5.3f} x ideal", flush=True)
