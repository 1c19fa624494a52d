"""Measurement build only (make -C rustracer_amd/csrc ABLATE=1): per-kernel-stage times and per-front-end vertex counts of a scene with parts of the
shade kernel switched off through RTX_DBG (1 textures, 2 env CDF search, 4 env map lookup, 8 BSDF-sampled MIS half, 16 differentials, 32 per-triangle uv gather, 64 half triangle record,
128 light pick without the row search, 256 every lane the first light's record, 512 light pdf without re-intersecting the emitter). Images are wrong.
Usage (GPU box, after `make -C rustracer_amd/csrc ABLATE=1`): python scripts/exp_ablate.py <scene> <spp> [dbg ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scene, spp = (sys.argv[1] if len(sys.argv) > 1 else "room"), int(sys.argv[2]) if len(sys.argv) > 2 else 256
if len(sys.argv) > 3 and sys.argv[3] == "child":
    from rustracer_amd import host
    from rustracer_amd.scenes import blob_scene, cornell_box, mis_plates, room_env
    d = {"room": room_env, "mis": mis_plates, "blob": blob_scene, "mis-spheres": (lambda spp: mis_plates(spp=spp, analytic_spheres=True))}[scene](spp=spp) if scene != "cornell" else cornell_box(1024, 1024, spp)
    h = host.HostScene(d); h.upload(); h.render(time_kernels=True)
    _, st = h.render(time_kernels=True)
    v = [st[k] for k in ("vertices_lambert_const", "vertices_lambert", "vertices_two_lobe", "vertices_generic")]
    sec = st["shade_section_cycles"]
    if any(sec):
        names = ["state+si", "emit+diff", "material", "lightpick", "light half", "bsdf half", "cont+store", "tail"]
        for fe, fname in enumerate(("lambert_const", "lambert", "two_lobe", "generic")):
            row = sec[8 * fe:8 * fe + 8]
            if sum(row):
                print(f"   {fname:14s} " + "  ".join(f"{n} {100.0 * c / sum(row):4.1f}%" for n, c in zip(names, row)), flush=True)
    print(f"dbg={os.environ.get('RTX_DBG', '0'):>3s} total {st['ms_total']:8.1f} shade {st['ms_shade']:8.1f} (L {st['ms_shade_lambert']:.0f} 2 {st['ms_shade_two_lobe']:.0f} G {st['ms_shade_generic']:.0f} bin {st['ms_shade_bin']:.0f} miss {st['ms_shade_miss']:.0f}) closest {st['ms_trace_closest']:7.1f} any {st['ms_trace_any']:6.1f} "
          f"mis {st['ms_trace_mis']:6.1f} resolve {st['ms_resolve']:6.1f} | vertices {v} rays c/s/m {st['rays_closest']} {st['rays_shadow']} {st['rays_mis']}", flush=True)
else:
    for dbg in (sys.argv[3:] or ["0", "1", "2", "4", "6", "8", "16", "31"]):
        subprocess.run([sys.executable, __file__, scene, str(spp), "child"], env=dict(os.environ, RTX_DBG=dbg))
