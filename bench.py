#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing backend (BASELINE.json configs[1]).

A "step" is one full frame of the hot path: the synthetic Cornell box (S1) at 1024x1024, PathIntegrator maxdepth=5,
1024 spp, scene already resident in HBM. With N > 1 (one process per GPU under torch.distributed) the film is sharded by interleaved
16-row tile rows, no collective on the data path, and gathered at the end of the frame: every rank sends the rows it touched to rank 0
over RCCL point-to-point sends, inside the timed region.

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel stage of the workload: algorithmic bytes (a ray cast =
48 B + 32 B per BVH node visited + 36 B per triangle tested, counted by an untimed counting frame of the same workload; a shaded vertex =
128 B + 32 B per ray it emits + 48 B of sampler tables per path) / the stage's HIP-event time measured in the timed frames.
`cpu_baseline` is the C++ oracle in its reference-faithful tile-sequential sampler mode on all host cores, built on this box with
-march=native, over a bounded sample of the same workload (reported, not the target).

At N = 1 the line also carries, under "other_configs", the same measurement for BASELINE configs[2..4] (S2 blob-1M, S3 mis-plates, S4 room-env;
2 timed frames each whatever --steps says) and, under "config_c1", configs[0] exactly (cornell 400x400, 64 spp) on the GPU and on the CPU port.
`--headline-only` skips those; `--scene X` makes X the headline instead.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE_SPP = {"cornell": 1024, "blob": 256, "mis": 512, "room": 1024}


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (what Rust's
    num_cpus::get(), the reference's default thread count (rc/api.rs:997-1001), reports as well)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


_NATIVE = {}


def native_oracle():
    """The CPU port compiled ON THIS BOX for its own cores (-O3 -march=native; -ffp-contract=off stays, Rust never contracts).
    The in-tree liborc.so is built without -march=native because it travels between machines; this copy does not."""
    if "orc" in _NATIVE:
        return _NATIVE["orc"], _NATIVE["flags"]
    from oracle import orc  # noqa: the checker, timed as the CPU baseline only
    flags = "-std=c++17 -O3 -march=native -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno -pthread"
    out = os.path.join(ROOT, "oracle", "_build", "liborc_native.so")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        src = [os.path.join(ROOT, "oracle", f) for f in ("orc_scene.cpp", "orc_api.cpp")]
        subprocess.check_call(["g++"] + flags.split() + ["-shared", "-o", out] + src, stderr=subprocess.DEVNULL)
        orc._LIB_PATH, orc._lib = out, None  # load the native build for the timing below
        orc.build = lambda force=False: out
    except Exception as e:  # no compiler on the box: the portable build, and say so
        flags = f"portable build (-O3, no -march=native): native compile failed ({type(e).__name__})"
    _NATIVE["orc"], _NATIVE["flags"] = orc, flags
    return orc, flags


def cpu_baseline(desc, cpu_spp, what="sample"):
    """The oracle (C++ restatement of rustracer's CPU path), reference-faithful tile-sequential sampler, all host cores."""
    import copy
    orc, flags = native_oracle()
    d = copy.copy(desc)
    d.sampler = copy.copy(desc.sampler)
    d.sampler.spp = cpu_spp
    o = orc.OracleScene(d)
    cores = host_cores()
    _, st = o.render(mode=0, n_threads=cores)
    sample = (f"same scene and resolution at {cpu_spp} spp ({st['camera_rays']} camera samples, {round(st['seconds'], 1)} s); Msamples/s is spp-independent"
              if what == "sample" else f"the whole configuration ({st['camera_rays']} camera samples, {round(st['seconds'], 1)} s)")
    return {"value": round(st["camera_rays"] / st["seconds"] / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port", "sample": sample,
            "build": flags, "Mrays_per_s": round((st["rays_closest"] + st["rays_shadow"] + st["rays_mis"]) / st["seconds"] / 1e6, 2)}


def make_desc(scene, spp, res=1024):
    from rustracer_amd.scenes import blob_scene, cornell_box, mis_plates, room_env
    if scene == "cornell":
        d = cornell_box(res, res, spp)
        return d, f"cornell-box (synthetic S1, 32 triangles, 2 area lights) {res}x{res} PathIntegrator maxdepth=5 {spp}spp 02sequence box-filter"
    d = {"blob": blob_scene, "mis": mis_plates, "room": room_env}[scene](spp=spp)
    return d, (f"{d.name} (synthetic, {d.n_tris} triangles, {len(d.lights)} lights) {d.film.xres}x{d.film.yres} "
               f"PathIntegrator maxdepth={d.integrator.max_depth} {spp}spp 02sequence box-filter")


def pmc_traffic(scene, kname):
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{scene}.json")))
        e = j.get(kname) or {}
        return e.get("hbm_bytes_per_launch"), {"file": f"profiles/pmc_{scene}.json", "from": j.get("source"), "collected": j.get("collected"), "method": j.get("method")}
    except Exception:
        return None, None


def run_workload(scene_name, scene, workload, steps, warmup, rank, world, local_rank, dist):
    """Times `steps` frames of one workload; returns the result dict (rank 0) or None."""
    import numpy as np
    import torch
    from rustracer_amd.distributed import merge_film
    scene.upload(local_rank)
    st0 = scene.setup()
    cr, sb = st0["cropped"], st0["sample_bounds"]
    h, w = int(cr[3] - cr[1]), int(cr[2] - cr[0])
    radius_y = float(st0["params"].filter_params[1])
    film = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream
    n_gpus = max(world, 1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timed):
        _, st = scene.render(rank=rank, world_size=world, time_kernels=timed, device_out=film, stream=stream)
        merge_film(film, dst=0, cropped=cr, sample_bounds=sb, filter_radius_y=radius_y)  # end-of-frame gather of the touched rows (no-op at N = 1)
        return st

    for _ in range(warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(steps):
        kstats.append(step(True))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([float(kstats[-1]["camera_rays"])], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        samples_per_step = float(cnt.item())
    else:
        samples_per_step = float(kstats[-1]["camera_rays"])
    # untimed counting frame: per-ray BVH node visits / triangle tests for the algorithmic byte count
    _, cst = scene.render(rank=rank, world_size=world, count_traversal=True, device_out=film, stream=stream)
    torch.cuda.synchronize()
    if rank != 0:
        return None
    ms_step = dt / steps * 1e3
    value = samples_per_step * steps / dt / 1e6
    ms_tc = float(np.mean([k["ms_trace_closest"] + k["ms_trace_mis"] for k in kstats]))
    ms_sh = float(np.mean([k["ms_shade"] for k in kstats]))

    def trace_roofline():
        rays = cst["rays_closest"] + cst["rays_mis"]
        algo = 48 * rays + 32 * (cst["nodes_closest"] + cst["nodes_mis"]) + 36 * (cst["tris_closest"] + cst["tris_mis"])
        return algo, kstats[-1]["launches_trace_closest"], ms_tc, "trace_closest", rays, "ray"

    def shade_roofline():
        verts = cst["rays_closest"]
        emitted = cst["rays_shadow"] + cst["rays_mis"] + (cst["rays_closest"] - cst["camera_rays"])
        return 128 * verts + 32 * emitted + 48 * cst["camera_rays"], kstats[-1]["launches_trace_closest"] // 2, ms_sh, "shade", verts, "vertex"

    def roof(algo_bytes, launches, ms_kernel, kname, unit_n, unit):
        achieved = algo_bytes / (ms_kernel * 1e-3) / 1e9  # GB/s
        traffic, prov = pmc_traffic(scene_name, kname)
        return {"bound": "hbm", "kernel": "k_" + kname, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_provenance": prov,
                "algorithmic_bytes_per_launch": round(algo_bytes / max(launches, 1)), "avg_launch_ms": round(ms_kernel / max(launches, 1), 4),
                "launches_per_step": int(launches), f"bytes_per_{unit}": round(algo_bytes / max(unit_n, 1), 1)}
    roofline = roof(*(trace_roofline() if ms_tc >= ms_sh else shade_roofline()))
    roofline_other = roof(*(shade_roofline() if ms_tc >= ms_sh else trace_roofline()))  # the runner-up of the two heavy stages
    kernels_ms = {k[3:]: round(float(np.mean([s[k] for s in kstats])), 2) for k in kstats[-1] if k.startswith("ms_")}
    verts = {k[9:]: int(kstats[-1][k]) for k in kstats[-1] if k.startswith("vertices_") and kstats[-1][k]}
    return {
        "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "sharding": "interleaved 16-row tile rows, end-of-frame gather of the touched rows to rank 0 (RCCL send/recv)" if n_gpus > 1 else "single GPU",
                   "sampler_mode": "pixel-keyed"},
        "s_per_frame": round(ms_step / 1e3, 4),
        "Mrays_per_s": round((kstats[-1]["rays_closest"] + kstats[-1]["rays_shadow"] + kstats[-1]["rays_mis"]) * (n_gpus if n_gpus > 1 else 1) / (ms_step * 1e-3) / 1e6, 1),
        "kernel_ms_per_step": kernels_ms, "vertices_by_shade_front_end": verts,
        "roofline": roofline, "roofline_second_kernel": roofline_other,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "blob", "mis", "room"],
                    help="the headline workload: cornell = BASELINE configs[1]; blob / mis / room = configs[2..4] (SURVEY.md §8d S2-S4)")
    ap.add_argument("--res", type=int, default=1024, help="cornell only (the other scenes use their BASELINE resolution)")
    ap.add_argument("--spp", type=int, default=0, help="0 = the BASELINE spp of the scene (cornell 1024, blob 256, mis 512, room 1024)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip configs[0] and configs[2..4] (they run only at N = 1 anyway)")
    ap.add_argument("--pbrt", default=None, help="render this pbrt-v3 scene file instead of a generated scene (read by the C++ host's parser; no CPU baseline)")
    args = ap.parse_args()

    import torch
    from rustracer_amd import host

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not host.device_available() or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    desc = None
    if args.pbrt:
        scene = host.PbrtScene(args.pbrt)
        if args.spp:
            scene.params.spp = args.spp
        args.spp, args.scene = scene.params.spp, "pbrt"
        workload = (f"{os.path.basename(args.pbrt)} ({len(scene.table('indices'))} triangles, {scene.n_lights()} lights) {scene.params.xres}x{scene.params.yres} "
                    f"PathIntegrator maxdepth={scene.params.max_depth} {args.spp}spp 02sequence")
    else:
        args.spp = args.spp or SCENE_SPP[args.scene]
        desc, workload = make_desc(args.scene, args.spp, args.res)
        scene = host.HostScene(desc)
    out = run_workload(args.scene, scene, workload, args.steps, args.warmup, rank, world, local_rank, dist)
    del scene
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline and desc is not None:
            cpu_spp = args.cpu_spp if args.scene != "room" else max(8, args.cpu_spp // 2)
            out["cpu_baseline"] = cpu_baseline(desc, cpu_spp)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if not args.headline_only and not args.pbrt:
            # BASELINE configs[2..4] next to the headline, measured the same way by the same run (2 timed frames each)
            others = {}
            for name in ("cornell", "blob", "mis", "room"):
                if name == args.scene:
                    continue
                torch.cuda.empty_cache()
                d, wl = make_desc(name, SCENE_SPP[name])
                r = run_workload(name, host.HostScene(d), wl, 2, 1, 0, 1, local_rank, None)
                if not args.no_cpu_baseline:
                    r["cpu_baseline"] = cpu_baseline(d, args.cpu_spp if name != "room" else max(8, args.cpu_spp // 2))
                    r["speedup_vs_cpu_baseline"] = round(r["value"] / r["cpu_baseline"]["value"], 1)
                others[name] = {k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "Mrays_per_s", "kernel_ms_per_step", "vertices_by_shade_front_end",
                                                  "roofline", "roofline_second_kernel", "cpu_baseline", "speedup_vs_cpu_baseline") if k in r}
            out["other_configs"] = others
            # BASELINE configs[0] exactly: cornell 400x400, 64 spp - the reference's own CPU-runnable case, on the GPU and on the CPU port
            d, wl = make_desc("cornell", 64, 400)
            r = run_workload("cornell", host.HostScene(d), wl, 5, 1, 0, 1, local_rank, None)
            c1 = {"workload": wl, "gpu": {"value": r["value"], "unit": "Msamples/s", "ms_per_step": r["ms_per_step"]}}
            if not args.no_cpu_baseline:
                c1["cpu_port"] = cpu_baseline(d, 64, what="whole")
            out["config_c1"] = c1
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
