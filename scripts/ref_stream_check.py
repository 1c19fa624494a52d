import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import cornell_box
from oracle import orc
from util import rel_l2
d = cornell_box(400, 400, 64)
t=time.time(); fo, so = orc.OracleScene(d).render(mode=0); to=time.time()-t
h = host.HostScene(d)
h.render(ref_stream=True)
t=time.time(); fr, sr = h.render(ref_stream=True); tr=time.time()-t
fk, sk = h.render()
ro, rr, rk = orc.film_to_rgb(fo), host.film_to_rgb(fr), host.film_to_rgb(fk)
print("weights equal", np.array_equal(fo[...,3], fr[...,3]), "rel_l2 ref-stream vs oracle mode 0:", rel_l2(rr, ro), "keyed vs oracle mode 0:", rel_l2(rk, ro))
print("pixels off by > 1e-3:", int((np.abs(rr-ro).max(-1) > 1e-3*(np.abs(ro).max(-1)+1e-3)).sum()), "of", rr.shape[0]*rr.shape[1])
print("rays oracle", [int(so[k]) for k in ("camera_rays","rays_closest","rays_shadow","rays_mis")], "device", [int(sr[k]) for k in ("camera_rays","rays_closest","rays_shadow","rays_mis")])
print("oracle %.2f s (all cores), device ref-stream %.3f s, ms_total %.1f" % (to, tr, sr["ms_total"]))
