#!/usr/bin/env python3
"""Parity numbers (medium size, vs oracle) and full-size GPU timings for the S2/S3/S4 workloads."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import blob_scene, mis_plates, room_env, cornell_box
from util import rel_l2

ap = argparse.ArgumentParser()
ap.add_argument("--parity", action="store_true")
ap.add_argument("--full", default="", help="comma list of blob,mis,room")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/scenes.json")
args = ap.parse_args()
res = {}
if args.parity:
    from oracle import orc
    for name, d in [("blob", blob_scene(256, 128, 160, 90, 32)), ("mis", mis_plates(160, 90, 32, sphere_level=2)),
                    ("room", room_env(160, 90, 32, detail=2, tex_size=256, env_size=256))]:
        t = time.time(); fo, so = orc.OracleScene(d).render(mode=1); to = time.time() - t
        t = time.time(); fh, sh = host.HostScene(d).render(count_traversal=True); th = time.time() - t
        ro, rh = orc.film_to_rgb(fo), host.film_to_rgb(fh)
        bad = np.abs(rh - ro).max(axis=-1) > 1e-3 * (np.abs(ro).max(axis=-1) + 1e-3)
        res["parity_" + name] = dict(rel_l2=rel_l2(rh, ro), bad_pixels=int(bad.sum()), n_pixels=int(bad.size), w_equal=bool(np.array_equal(fo[..., 3], fh[..., 3])),
                                     oracle_s=to, gpu_s=th, rays_o=[int(so[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")],
                                     rays_g=[int(sh[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")])
        print(name, res["parity_" + name], flush=True)
full = {"blob": lambda: blob_scene(spp=args.spp or 256), "mis": lambda: mis_plates(spp=args.spp or 512), "room": lambda: room_env(spp=args.spp or 1024),
        "cornell": lambda: cornell_box(1024, 1024, args.spp or 1024)}
for name in [x for x in args.full.split(",") if x]:
    t = time.time(); d = full[name](); tg = time.time() - t
    t = time.time(); h = host.HostScene(d); tb = time.time() - t
    t = time.time(); h.upload(0); tu = time.time() - t
    h.render(time_kernels=False) if name == "cornell" else None
    t = time.time(); film, st = h.render(time_kernels=True); tr = time.time() - t
    rgb = host.film_to_rgb(film)
    k = {a: round(float(b), 2) for a, b in st.items() if a.startswith("ms_")}
    res["full_" + name] = dict(tris=d.n_tris, lights=len(d.lights), gen_s=tg, build_s=tb, upload_s=tu, render_s=tr, msamples_s=st["camera_rays"] / tr / 1e6,
                               rays=[int(st[k2]) for k2 in ("rays_closest", "rays_shadow", "rays_mis")], finite=bool(np.isfinite(rgb).all()), mean=rgb.mean(axis=(0, 1)).tolist(), kernels=k)
    print(name, res["full_" + name], flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(res, open(args.out, "w"), indent=1)
