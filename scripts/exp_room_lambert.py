"""Ablation: S4 with every material constant matte and the environment light replaced by a point light, through k_shade<0>."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import host, scene_desc as sd
from rustracer_amd.scenes import room_env
d = room_env(spp=64, tex_size=64, env_size=64)
grey = d.const_tex((0.5, 0.5, 0.5)); zero = d.const_tex(0.0)
for m in d.materials:
    m.kind, m.params, m.bump = sd.MAT_MATTE, {"kd": grey, "sigma": zero}, -1
keep_env = len(sys.argv) > 1 and sys.argv[1] == "env"
if not keep_env:
    for i, l in enumerate(d.lights):
        if l.kind == sd.LIGHT_INFINITE:
            d.lights[i] = sd.Light(sd.LIGHT_POINT, rgb=(30.0, 30.0, 30.0), vec=(0.5, 2.0, 0.5))
h = host.HostScene(d); h.upload(); h.render(time_kernels=True)
f, st = h.render(time_kernels=True)
print(f"{'env' if keep_env else 'point':6s} total {st['ms_total']:8.1f} ms  shade {st['ms_shade']:8.1f}  closest {st['ms_trace_closest']:8.1f}  any {st['ms_trace_any']:7.1f}  mis {st['ms_trace_mis']:7.1f} resolve {st['ms_resolve']:6.1f}", flush=True)
