"""Where does the generic shade kernel's time go on S4? Same scene with (a) EWA image maps, (b) trilinear image maps, (c) constant colours."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import room_env
from rustracer_amd import scene_desc as sd

def run(tag, d):
    h = host.HostScene(d)
    h.upload()
    h.render(time_kernels=True)
    f, st = h.render(time_kernels=True)
    print(f"{tag:12s} total {st['ms_total']:8.1f} ms  shade {st['ms_shade']:8.1f}  closest {st['ms_trace_closest']:8.1f}  any {st['ms_trace_any']:7.1f}  mis {st['ms_trace_mis']:7.1f}  resolve {st['ms_resolve']:6.1f}  sampler {st['ms_sampler']:6.1f}", flush=True)

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = room_env(spp=spp, tex_size=256, env_size=512)
run("ewa", d)
for m in d.mipmaps[:-1]:
    m.trilinear = True
run("trilinear", d)
for t in d.textures:
    if t.kind == sd.TEX_IMAGE:
        t.kind = sd.TEX_CONST; t.value = (0.5, 0.4, 0.3); t.mip = -1
run("const", d)
for m in d.materials:
    if m.kind != sd.MAT_MIX:
        kd = m.params.get("kd", d.const_tex((0.5, 0.5, 0.5)))
        m.kind, m.params, m.bump = sd.MAT_MATTE, {"kd": kd if d.textures[kd].kind == sd.TEX_CONST else d.const_tex((0.5, 0.5, 0.5)), "sigma": d.const_tex(0.0)}, -1
for m in d.materials:
    if m.kind == sd.MAT_MIX:
        m.kind, m.params = sd.MAT_MATTE, {"kd": d.const_tex((0.5, 0.5, 0.5)), "sigma": d.const_tex(0.0)}
run("all matte", d)
for i, l in enumerate(d.lights):
    if l.kind == sd.LIGHT_INFINITE:
        d.lights[i] = sd.Light(sd.LIGHT_POINT, rgb=(30.0, 30.0, 30.0), vec=(0.5, 2.0, 0.5))
run("matte+point", d)
