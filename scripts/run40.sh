#!/bin/bash
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig.so; cp rustracer_amd/csrc/_build/strict.so $LIB
cat > /tmp/p.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import mis_plates
from oracle import orc
for spp in (32, 128):
    d = mis_plates(160, 90, spp, analytic_spheres=True)
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = host.HostScene(d).render()
    ro, rh = orc.film_to_rgb(fo), host.film_to_rgb(fh)
    err = np.abs(rh - ro).max(-1); thr = 1e-3 * (np.abs(ro).max(-1) + 1e-3)
    bad = err > thr
    print("spp", spp, "bad", int(bad.sum()), "max abs err", float(err.max()), "rel_l2", float(np.linalg.norm(rh - ro) / np.linalg.norm(ro)))
    ys, xs = np.nonzero(bad)
    print("   rows of bad pixels: min/max", ys.min() if len(ys) else None, ys.max() if len(ys) else None, "cols", xs.min() if len(xs) else None, xs.max() if len(xs) else None)
PY
echo strict; python /tmp/p.py 2>&1 | grep -v amdgpu.ids
cp /tmp/orig.so $LIB
echo default; python /tmp/p.py 2>&1 | grep -v amdgpu.ids
