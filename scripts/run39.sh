#!/bin/bash
cat > /tmp/p.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import mis_plates
from oracle import orc
d = mis_plates(160, 90, 32, analytic_spheres=True)
fo, so = orc.OracleScene(d).render(mode=1)
h = host.HostScene(d)
fh, sh = h.render()
fc, sc = h.render(count_traversal=True)
ro, rh, rc = orc.film_to_rgb(fo), host.film_to_rgb(fh), host.film_to_rgb(fc)
for nm, r in (("production", rh), ("counting", rc)):
    err = np.abs(r - ro).max(-1); thr = 1e-3 * (np.abs(ro).max(-1) + 1e-3)
    bad = err > thr
    print(nm, "bad", int(bad.sum()), "max abs err", float(err.max()), "at radiance", float(np.abs(ro).max(-1)[np.unravel_index(err.argmax(), err.shape)]), "max rel on bad", float((err[bad] / (np.abs(ro).max(-1)[bad] + 1e-12)).max()) if bad.any() else 0, "median radiance of bad", float(np.median(np.abs(ro).max(-1)[bad])) if bad.any() else 0)
print("production == counting:", bool(np.array_equal(fh, fc)), float(np.abs(rh - rc).max()))
PY
python /tmp/p.py 2>&1 | grep -v amdgpu.ids
RTX_MIS_REACH=0 python /tmp/p.py 2>&1 | grep -v amdgpu.ids
