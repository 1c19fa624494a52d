"""Measurement: wall time of one HostScene.render() call against rt_render's own ms_total (what is spent outside the C call), per step."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from rustracer_amd import host
name = sys.argv[1] if len(sys.argv) > 1 else "instances-10k"
d, wl = bench.make_desc(name, bench.SCENE_SPP[name])
r = bench.Runner("single", host.HostScene(d))
for timed in (False, True):
    r.step(timed); torch.cuda.synchronize()
    rows = []
    for _ in range(8):
        t0 = time.perf_counter(); st = r.step(timed); torch.cuda.synchronize(); t1 = time.perf_counter()
        rows.append((round((t1 - t0) * 1e3, 2), round(st["ms_total"], 2)))
    print(name, "time_kernels", timed, rows)
