#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing backend (BASELINE.json configs[1]).

A "step" is one full frame of the hot path: the synthetic Cornell box (S1) at
1024x1024, PathIntegrator maxdepth=5, 1024 spp, scene already resident in HBM. With N > 1 the film is
sharded by interleaved 16-row tile rows (one process per GPU, no collective on the data path) and
merged by one end-of-frame sum-reduce to rank 0 over RCCL, which is inside the timed region.

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel, trace_closest: algorithmic
bytes (48 B per cast + 32 B per BVH node visited + 36 B per triangle tested, counted by
an untimed counting frame of the same workload) / the kernel's HIP-event time measured in the timed
frames. `cpu_baseline` is the C++ oracle in its reference-faithful tile-sequential sampler mode on all
host cores, over a bounded sample of the same workload (reported, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "blob", "mis", "room"],
                    help="cornell = BASELINE configs[1] (the headline); blob / mis / room = configs[2..4] (SURVEY.md §8d S2-S4)")
    ap.add_argument("--res", type=int, default=1024, help="cornell only (the other scenes use their BASELINE resolution)")
    ap.add_argument("--spp", type=int, default=0, help="0 = the BASELINE spp of the scene (cornell 1024, blob 256, mis 512, room 1024)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pbrt", default=None, help="render this pbrt-v3 scene file instead of a generated scene (read by the C++ host's parser; no CPU baseline)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from rustracer_amd import host
    from rustracer_amd.distributed import merge_film
    from rustracer_amd.scenes import blob_scene, cornell_box, mis_plates, room_env

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = max(args.gpus, 1)
    if world != n_gpus and world > 1:
        n_gpus = world
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
    elif args.scene == "cornell":
        args.spp = args.spp or 1024
        desc = cornell_box(args.res, args.res, args.spp)
        workload = f"cornell-box (synthetic S1, 32 triangles, 2 area lights) {args.res}x{args.res} PathIntegrator maxdepth=5 {args.spp}spp 02sequence box-filter"
    else:
        gen, spp0 = {"blob": (blob_scene, 256), "mis": (mis_plates, 512), "room": (room_env, 1024)}[args.scene]
        args.spp = args.spp or spp0
        desc = gen(spp=args.spp)
        workload = (f"{desc.name} (synthetic, {desc.n_tris} triangles, {len(desc.lights)} lights) {desc.film.xres}x{desc.film.yres} "
                    f"PathIntegrator maxdepth={desc.integrator.max_depth} {args.spp}spp 02sequence box-filter")
    if desc is not None:
        scene = host.HostScene(desc)
    scene.upload(local_rank)
    st0 = scene.setup()
    cr = st0["cropped"]
    h, w = int(cr[3] - cr[1]), int(cr[2] - cr[0])
    film = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timed):
        _, st = scene.render(rank=rank, world_size=world, time_kernels=timed, device_out=film, stream=stream)
        merge_film(film, dst=0)  # end-of-frame film merge (no-op at N = 1)
        return st

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(args.steps):
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

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = samples_per_step * args.steps / dt / 1e6
        # --- roofline of the dominant kernel. Algorithmic bytes follow SURVEY.md §8(d): a ray cast = 32 B ray record read + 16 B hit
        # record written + 32 B per BVH node visited + 36 B per triangle tested (the reference algorithm's visit counts, from the
        # counting frame); a shaded vertex = 64 B path state read + 64 B written + 32 B per ray it emits, + 48 B of sampler tables per path.
        ms_tc = float(np.mean([k["ms_trace_closest"] + k["ms_trace_mis"] for k in kstats]))
        ms_sh = float(np.mean([k["ms_shade"] for k in kstats]))
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{args.scene}.json")))
        except Exception:
            pmc = {}
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
            return {"bound": "hbm", "kernel": "k_" + kname, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": (pmc.get(kname) or {}).get("hbm_bytes_per_launch"),
                    "algorithmic_bytes_per_launch": round(algo_bytes / launches), "avg_launch_ms": round(ms_kernel / launches, 4),
                    "launches_per_step": int(launches), f"bytes_per_{unit}": round(algo_bytes / unit_n, 1)}
        roofline = roof(*(trace_roofline() if ms_tc >= ms_sh else shade_roofline()))
        roofline_other = roof(*(shade_roofline() if ms_tc >= ms_sh else trace_roofline()))  # the runner-up of the two heavy kernels
        kernels_ms = {k[3:]: round(float(np.mean([s[k] for s in kstats])), 2) for k in kstats[-1] if k.startswith("ms_")}
        out = {
            "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "sharding": "interleaved 16-row tile rows, end-of-frame sum-reduce to rank 0" if n_gpus > 1 else "single GPU",
                       "sampler_mode": "pixel-keyed"},
            "s_per_frame": round(ms_step / 1e3, 4),
            "Mrays_per_s": round((kstats[-1]["rays_closest"] + kstats[-1]["rays_shadow"] + kstats[-1]["rays_mis"]) * (n_gpus if n_gpus > 1 else 1) / (ms_step * 1e-3) / 1e6, 1),
            "kernel_ms_per_step": kernels_ms,
            "roofline": roofline,
            "roofline_second_kernel": roofline_other,
        }
        if n_gpus == 1 and not args.no_cpu_baseline and desc is not None:
            out["cpu_baseline"] = cpu_baseline(desc, args)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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


def cpu_baseline(desc, args):
    """The oracle (C++ restatement of rustracer's CPU path), reference-faithful tile-sequential sampler, all host cores."""
    import copy
    from oracle import orc  # noqa: the checker, timed as the CPU baseline only
    d = copy.copy(desc)
    d.sampler = copy.copy(desc.sampler)
    d.sampler.spp = args.cpu_spp
    o = orc.OracleScene(d)
    cores = host_cores()
    _, st = o.render(mode=0, n_threads=cores)
    return {"value": round(st["camera_rays"] / st["seconds"] / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"same scene and resolution at {args.cpu_spp} spp ({st['camera_rays']} camera samples, {round(st['seconds'], 1)} s); "
                      "Msamples/s is spp-independent",
            "Mrays_per_s": round((st["rays_closest"] + st["rays_shadow"] + st["rays_mis"]) / st["seconds"] / 1e6, 2)}


if __name__ == "__main__":
    main()
