"""Full-size S2 / S3 / S4 through the file path: SceneDesc -> .pbrt (+ PLY, PFM) -> C++ parser (device ingest) -> render, against the same
SceneDesc handed over call by call. Prints parse / build times and the film difference (expected: identical)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import host
from rustracer_amd.pbrt_export import write_pbrt
from rustracer_amd.scenes import blob_scene, room_env, mis_plates
os.makedirs("/tmp/soak", exist_ok=True)
for name, d in (("blob", blob_scene(spp=4)), ("mis", mis_plates(1280, 720, 4)), ("room", room_env(spp=2))):
    d.name = name
    t0 = time.time(); write_pbrt(d, f"/tmp/soak/{name}.pbrt", ply_over=20000); t1 = time.time()
    p = host.PbrtScene(f"/tmp/soak/{name}.pbrt", device_ingest=True); t2 = time.time()
    h = host.HostScene(d); t3 = time.time()
    fp, _ = p.render(); fh, _ = h.render()
    same_w = np.array_equal(fp[..., 3], fh[..., 3])
    err = float(np.linalg.norm(fp[..., :3].astype(np.float64) - fh[..., :3]) / np.linalg.norm(fh[..., :3].astype(np.float64)))
    print(f"{name}: export {t1-t0:.1f}s, parse+build {t2-t1:.2f}s (host path {t3-t2:.2f}s), {len(p.table('indices'))} tris, weights equal {same_w}, rel L2 {err:.2e}, warnings {p.n_warnings}", flush=True)
