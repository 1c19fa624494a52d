"""Film parity of the PRODUCTION kernels (the ones rt_render launches, not the counting ones) against the oracle on medium-size versions of the six bench workloads:
relative L2 of the RGB films, pixels off by more than 1e-3, ray counts. Writes profiles/<tag>_parity.json. Usage (GPU box): python scripts/parity_report.py r03"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import blob_scene, cornell_box, forest, mis_plates, room_env
from oracle import orc
from util import rel_l2

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
res = {}
for name, d in [("cornell", cornell_box(128, 128, 64)), ("blob", blob_scene(256, 128, 160, 90, 32)), ("mis", mis_plates(160, 90, 32, sphere_level=2)),
                ("room", room_env(160, 90, 32, detail=2, tex_size=256, env_size=256)), ("mis-spheres", mis_plates(160, 90, 32, analytic_spheres=True)),
                ("instances", forest(12, 2, 16, True, (160, 90)))]:
    t = time.time(); fo, so = orc.OracleScene(d).render(mode=1); to = time.time() - t
    fh, sh = host.HostScene(d).render()
    ro, rh = orc.film_to_rgb(fo), host.film_to_rgb(fh)
    bad = np.abs(rh - ro).max(axis=-1) > 1e-3 * (np.abs(ro).max(axis=-1) + 1e-3)
    res[name] = dict(rel_l2=float(rel_l2(rh, ro)), pixels_off_by_1e3=int(bad.sum()), n_pixels=int(bad.size), weights_equal=bool(np.array_equal(fo[..., 3], fh[..., 3])), oracle_s=round(to, 2),
                     rays_oracle=[int(so[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")], rays_gpu=[int(sh[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")],
                     mis_rays_not_cast=int(sh.get("rays_mis_not_cast", 0)))
    print(name, res[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"{tag}_parity.json"), "w"), indent=1)
