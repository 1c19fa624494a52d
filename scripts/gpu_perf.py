import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from rustracer_amd import host
from rustracer_amd.scenes import cornell_box
res = int(sys.argv[1]); spp = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
d = cornell_box(res, res, spp)
h = host.HostScene(d)
for i in range(reps):
    f, s = h.render(time_kernels=(i == reps - 1))
    print(i, 'ms', round(s['ms_total'], 2), 'Msamples/s', round(s['camera_rays'] / s['ms_total'] / 1e3, 2), 'Mrays/s', round((s['rays_closest'] + s['rays_shadow'] + s['rays_mis']) / s['ms_total'] / 1e3, 1))
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in s.items()}))
