#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing backend (BASELINE.json configs[1]).

A "step" is one full frame of the hot path: the synthetic Cornell box (S1) at 1024x1024, PathIntegrator maxdepth=5,
1024 spp, scene already resident in HBM. With N > 1 (one process per GPU under torch.distributed) the film is sharded by interleaved
4-row bands (RT_SHARD_ROWS), no collective on the data path, and gathered at the end of the frame: every rank sends the rows it touched to rank 0
over RCCL point-to-point sends, inside the timed region.

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel stage of the workload: algorithmic bytes (a ray cast =
48 B + 32 B per BVH node visited + 36 B per triangle tested, counted by an untimed counting frame of the same workload; a shaded vertex =
128 B + 32 B per ray it emits + 48 B of sampler tables per path) / the stage's HIP-event time measured in the timed frames.
`cpu_baseline` is the C++ oracle in its reference-faithful tile-sequential sampler mode on all host cores, built on this box with
-march=native, over a bounded sample of the same workload (reported, not the target).

`--gpus N` with N > 1: under torchrun (WORLD_SIZE = N, what the driver launches) every rank renders its rows with rt_render and rank 0 gathers them over RCCL;
started as a plain `python bench.py --gpus N` the ONE process drives N devices itself through rt_multi_render (one host thread per device, tile-row chunks from
a shared queue, touched rows peer-copied to device 0) - renderer::render's worker pool with GPUs as the workers (rc/renderer.rs:22-29, 47-71). Either way the
run fails when fewer than N devices are visible; `n_gpus` in the line is the number of devices that actually traced camera rays.

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

SCENE_SPP = {"cornell": 1024, "blob": 256, "mis": 512, "room": 1024, "mis-spheres": 512, "instances-10k": 64}
EXTRA_SCENES = ("mis-spheres", "instances-10k")  # general-primitive workloads measured under other_configs (no CPU baseline: they are not BASELINE configs)


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
    if scene == "mis-spheres":  # S3 with its four emitters as Shape "sphere" - what veach-mis.pbrt is
        d = mis_plates(spp=spp, analytic_spheres=True)
        d.name = "mis-plates, analytic sphere lights"
    elif scene == "instances-10k":  # 10 000 placements of a 1280-triangle object kept two-level (TransformedPrimitive, rc/primitive.rs:79-118)
        from rustracer_amd.scenes import forest
        d = forest(100, 3, spp)
    else:
        d = {"blob": blob_scene, "mis": mis_plates, "room": room_env}[scene](spp=spp)
    extra = ""
    if getattr(d, "spheres", None):
        extra += f" + {len(d.spheres)} analytic quadrics"
    if getattr(d, "instances", None):
        extra += f" + {len(d.instances)} instances of {len(d.objects)} object(s) ({sum(int(o.idx.shape[0]) for o in d.objects)} triangles in their definitions)"
    return d, (f"{d.name} (synthetic, {d.n_tris} triangles{extra}, {len(d.lights)} lights) {d.film.xres}x{d.film.yres} "
               f"PathIntegrator maxdepth={d.integrator.max_depth} {spp}spp 02sequence box-filter")


def source_sha():
    """sha256 over the sources librtx_hip.so is built from: ties a PMC traffic file to the kernels it was collected from (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rustracer_amd", "csrc")
    host_only = ("rtx_pbrt.inl", "rtx_images.inl", "rtx_spectrum_tables.inl")  # parts of librtx_host.so: no kernel is built from them
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".inl")) and f not in host_only:
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(scene, kname):
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{scene}.json")))
        e = j.get(kname) or {}
        prov = {"file": f"profiles/pmc_{scene}.json", "from": j.get("source"), "collected": j.get("collected"), "method": j.get("method"),
                "kernel_source_sha": j.get("kernel_source_sha")}
        # the traffic file is a measurement of the tree it names, not of this run: say when the kernels have changed since
        prov["stale"] = (j.get("kernel_source_sha") != source_sha())
        return e, prov
    except Exception:
        return {}, None


def occupancy_of(kernel_substr):
    """Registers and waves per SIMD of the built kernels whose name contains the string (read from the .so that is about to run)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from kernel_budget import kernel_resources
        out = {}
        for k, v in kernel_resources().items():
            if kernel_substr in k:
                alloc = -(-max(v["vgpr"] + v["agpr"], 1) // 8) * 8
                out[k.replace("rtx::", "")] = {"vgpr": v["vgpr"], "waves_per_simd": min(8, 512 // alloc), "scratch": v["scratch"], "lds": v["lds"]}
        return out
    except Exception:
        return None


class ClockSampler:
    """sclk / mclk / socket power of the GPU read from sysfs right before and right after the timed frames (one binary's k_shade<1> took 370 - 421 ms
    "depending on the box and the hour" with no clock reading beside it). NOT while they run: reading pp_dpm_sclk / power1_average makes the driver query the
    SMU, and a thread doing that every 50 ms cost short frames 15 - 25 ms each on some runs (S2 177 / 202 ms, S3 328 / 353, instances 85 / 101 - found in round 4
    when the stage times of a run no longer added up to its wall time). Best effort: a box without the files yields None."""

    def __init__(self, card=None):
        import glob
        self.rows = []
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.dev = os.path.dirname(cards[card or 0]) if cards else None
        try:  # the card whose PCI address is the current HIP device's (a box may expose more cards in sysfs than it lets the process use)
            import torch
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for c in cards:
                if want in os.path.realpath(os.path.dirname(c)):
                    self.dev = os.path.dirname(c)
            self.pci = want
        except Exception:
            self.pci = None
        pw = glob.glob(os.path.join(self.dev, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(self.dev, "hwmon/hwmon*/power1_input")) if self.dev else []
        self.power = pw[0] if pw else None
        # the socket power limit of this GPU (boxes of one pool differ: k_shade<1>, the one stage that loads both the VALUs and HBM, took 271 ms per S1 frame on
        # GPUs that drew 950 - 1000 W in it and 328 ms on one that stayed at 850 W, every other stage equal)
        self.power_cap = None
        try:
            cap = glob.glob(os.path.join(self.dev, "hwmon/hwmon*/power1_cap")) if self.dev else []
            self.power_cap = int(open(cap[0]).read()) / 1e6 if cap else None
        except Exception:
            pass

    @staticmethod
    def _active(path):
        try:
            for ln in open(path):
                if "*" in ln:
                    return int(float(ln.split(":")[1].strip().split("M")[0]))
        except Exception:
            pass
        return None

    def sample(self):
        if not self.dev:
            return
        p = None
        try:
            p = int(open(self.power).read()) / 1e6 if self.power else None
        except Exception:
            pass
        self.rows.append((self._active(os.path.join(self.dev, "pp_dpm_sclk")), self._active(os.path.join(self.dev, "pp_dpm_mclk")), p))

    def __enter__(self):
        self.sample()
        return self

    def __exit__(self, *a):
        self.sample()

    def summary(self):
        out = {}
        for k, name in enumerate(("sclk_MHz", "mclk_MHz", "power_W")):
            v = [r[k] for r in self.rows if r[k] is not None]
            if v:
                out[name] = {"before": v[0], "after": v[-1]}
        return dict(out, pci=self.pci, power_cap_W=self.power_cap) if out else None


class Runner:
    """One of the three ways a frame is rendered: 'single' (rt_render on one device), 'dist' (one process per GPU, rows gathered over RCCL),
    'multi' (this process drives several devices through rt_multi_render)."""

    def __init__(self, mode, scene, rank=0, world=1, local_rank=0, dist=None, devices=None, chunks=1):
        import torch
        self.mode, self.scene, self.rank, self.world, self.local_rank, self.dist, self.devices, self.chunks = mode, scene, rank, world, local_rank, dist, devices, chunks
        self.torch = torch
        dev0 = devices[0] if mode == "multi" else local_rank
        if mode != "multi":
            scene.upload(local_rank)
        st0 = scene.setup()
        self.cr, self.sb = st0["cropped"], st0["sample_bounds"]
        h, w = int(self.cr[3] - self.cr[1]), int(self.cr[2] - self.cr[0])
        self.radius_y = float(st0["params"].filter_params[1])
        self.film = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{dev0}")
        with torch.cuda.device(dev0):
            self.stream = torch.cuda.current_stream().cuda_stream
        self.per_device = None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.mode == "multi":
            for d in sorted(set(self.devices)):
                self.torch.cuda.synchronize(d)
        else:
            self.torch.cuda.synchronize()

    def step(self, timed):
        if self.mode == "multi":
            _, st, per = self.scene.render_multi(self.devices, chunks_per_device=self.chunks, time_kernels=timed, device_out=self.film)
            st["ms_total_max_device"] = max(p["ms_total"] for p in per)
            self.per_device = per
            return st
        from rustracer_amd.distributed import merge_film
        _, st = self.scene.render(rank=self.rank, world_size=self.world, time_kernels=timed, device_out=self.film, stream=self.stream)
        t0 = time.perf_counter()
        merge_film(self.film, dst=0, cropped=self.cr, sample_bounds=self.sb, filter_radius_y=self.radius_y)  # end-of-frame gather of the touched rows (no-op at N = 1)
        if self.dist is not None:  # this rank's share of the gather: its send (or, on rank 0, the receives and the additions) done
            self.torch.cuda.synchronize()
            st["ms_gather"] = (time.perf_counter() - t0) * 1e3
        return st

    def count(self):
        """untimed counting frame: per-ray BVH node visits / triangle tests of the walks the timed frames run"""
        if self.mode == "multi":
            _, st, _ = self.scene.render_multi(self.devices, chunks_per_device=self.chunks, count_traversal=True, count_as_rendered=True, device_out=self.film)
        else:
            _, st = self.scene.render(rank=self.rank, world_size=self.world, count_traversal=True, count_as_rendered=True, device_out=self.film, stream=self.stream)
        return st


def run_workload(scene_name, runner, workload, steps, warmup):
    """Times `steps` frames of one workload; returns the result dict (rank 0) or None."""
    import numpy as np
    import torch
    rank, world, dist = runner.rank, runner.world, runner.dist
    # integrator.preprocess runs inside renderer::render (rc/renderer.rs:30); rt_render builds the light-distribution tables in the FIRST frame of a scene and keeps
    # them, so the timed frames (after a warm-up) do not contain it: the first frame's build time is reported beside them (config.light_distribution)
    lightdist_first_ms = None
    for k in range(warmup):
        st = runner.step(k == 0)
        if k == 0:
            lightdist_first_ms = st.get("ms_lightdist")
    kstats = []
    clocks = ClockSampler()
    clocks.sample()  # (before the opening barrier: the sysfs reads are outside the timed region)
    runner.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        kstats.append(runner.step(True))
    runner.barrier()
    dt = time.perf_counter() - t0
    clocks.sample()
    cst = runner.count()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{runner.local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # whole-frame counters: every rank's share summed; stage times: the slowest rank's
        keys_sum = [k for k, v in cst.items() if isinstance(v, int)]
        v = torch.tensor([float(kstats[-1]["camera_rays"])] + [float(cst[k]) for k in keys_sum], dtype=torch.float64, device=f"cuda:{runner.local_rank}")
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        samples_per_step = float(v[0].item())
        for k, x in zip(keys_sum, v[1:].tolist()):
            cst[k] = int(x)
        mine = torch.tensor([float(kstats[-1]["camera_rays"] > 0)], dtype=torch.float64, device=f"cuda:{runner.local_rank}")
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        n_active = int(mine.item())
        ms_keys = [k for k in kstats[-1] if k.startswith("ms_")]
        m = torch.tensor([[ks[k] for k in ms_keys] for ks in kstats], dtype=torch.float64, device=f"cuda:{runner.local_rank}").mean(0)
        gathered = [torch.zeros_like(m) for _ in range(world)]
        dist.all_gather(gathered, m)
        per_rank_ms = [float(g[ms_keys.index("ms_total")].item()) for g in gathered]
        mmax = torch.stack(gathered).max(0).values.tolist()
        ms_mean = dict(zip(ms_keys, mmax))
        n_gpus_asked = world
    else:
        samples_per_step = float(kstats[-1]["camera_rays"])
        ms_keys = [k for k in kstats[-1] if k.startswith("ms_")]
        ms_mean = {k: float(np.mean([ks[k] for ks in kstats])) for k in ms_keys}
        if runner.mode == "multi":
            nd = len(runner.devices)
            n_active = sum(1 for p in runner.per_device if p["camera_rays"] > 0)
            per_rank_ms = [round(p["ms_total"], 2) for p in runner.per_device]
            # stage timers of rt_multi_render's total are device-milliseconds summed over the devices: per-device mean for the roofline
            for k in ms_keys:
                if k not in ("ms_total", "ms_gather", "ms_total_max_device"):
                    ms_mean[k] /= nd
            n_gpus_asked = nd
        else:
            n_active, per_rank_ms, n_gpus_asked = 1, [round(ms_mean["ms_total"], 2)], 1
    if rank != 0:
        return None
    ms_step = dt / steps * 1e3
    value = samples_per_step * steps / dt / 1e6
    n_gpus = n_active  # devices that traced camera rays in the last timed frame
    ldiv = n_gpus_asked if runner.mode == "multi" else 1
    launches_path = kstats[-1]["launches_trace_path"] // ldiv   # per-stage launch counts as rt_render issued them (rt_stats, round 5; round 4 derived them as launches_trace_closest // 2)
    launches_shade = kstats[-1]["launches_shade"] // ldiv
    if lightdist_first_ms is None:  # no warm-up: the first timed frame built the tables, its time is inside the timed region
        lightdist_first_ms = kstats[0].get("ms_lightdist")

    # SURVEY §8d bytes of the three ray classes, each divided by the time of ITS launches. The counting frame walks what the timed frames walk
    # (RT_FLAG_COUNT_AS_RENDERED): path rays and MIS rays toward area lights closest-hit, shadow rays and MIS rays toward the environment any-hit.
    def ray_class(rays, nodes, tris, ms, per_ray):
        b = per_ray * rays + 32 * nodes + 36 * tris
        # `reference_walk_rate`: the bytes the REFERENCE's node-at-a-time walk would move for these rays / the time of the launches that traced them, over 8 TB/s.
        # A rate in the reference's currency, not a share of this GPU's bandwidth: a four-wide kernel fetches fewer bytes than that for the same
        # walk and a tree top held in LDS none, so it can exceed 1. The share of the bandwidth is `hbm_frac` of the kernel group below (PMC bytes / time / 8 TB/s).
        return {"rays": int(rays), "algorithmic_bytes": int(b), "ms": round(ms, 3), "reference_walk_GB_per_s": round(b / max(ms, 1e-9) / 1e6, 1),
                "reference_walk_rate": round(b / max(ms, 1e-9) / 1e6 / 8000.0, 4),
                "nodes_per_ray": round(nodes / max(rays, 1), 2), "tris_per_ray": round(tris / max(rays, 1), 2)}
    ms_mis_any = ms_mean.get("ms_trace_mis_any", 0.0)
    not_cast = cst.get("rays_mis_not_cast", 0)  # of rays_mis: rays toward a sphere light that cannot reach it (rt_stats::rays_mis_not_cast): in no launch, no bytes
    tail_nc = cst.get("rays_tail_not_cast", 0)  # of rays_closest: path rays at the depth limit after a non-specular bounce (nothing reads their hit): in no launch, no vertex, no bytes
    nd_div = n_gpus_asked if runner.mode != "single" else 1  # whole-frame bytes against per-device (mean / slowest-rank) stage time: bytes per device
    classes = {
        "path_closest": ray_class((cst["rays_closest"] - tail_nc) / nd_div, cst["nodes_closest"] / nd_div, cst["tris_closest"] / nd_div, ms_mean["ms_trace_closest"], 48),
        "shadow_any": ray_class(cst["rays_shadow"] / nd_div, cst["nodes_shadow"] / nd_div, cst["tris_shadow"] / nd_div, ms_mean["ms_trace_any"], 36),
        "mis_closest": ray_class((cst["rays_mis"] - cst["rays_mis_any"] - not_cast) / nd_div, (cst["nodes_mis"] - cst["nodes_mis_any"]) / nd_div, (cst["tris_mis"] - cst["tris_mis_any"]) / nd_div,
                                 ms_mean["ms_trace_mis"] - ms_mis_any, 48),
        "mis_any": ray_class(cst["rays_mis_any"] / nd_div, cst["nodes_mis_any"] / nd_div, cst["tris_mis_any"] / nd_div, ms_mis_any, 36),
    }
    # What the traversal kernels really take from HBM: PMC bytes per camera sample of each kernel group (profiles/pmc_<scene>.json, collected at this pass size)
    # x this frame's samples / the group's launches' time / 8 TB/s. PMC counters are per kernel name, so the path and MIS launches of one kernel are one group.
    def hbm_frac(group, ms):
        e, prov = pmc_entry(scene_name, group)
        bps = e.get("hbm_bytes_per_camera_sample")
        if not bps or ms <= 0:
            return None
        n = cst["camera_rays"] / nd_div
        return {"hbm_frac": round(bps * n / (ms * 1e-3) / 8e12, 4), "hbm_frac_raw_reads": round(e.get("hbm_bytes_per_camera_sample_raw", 0.0) * n / (ms * 1e-3) / 8e12, 4),
                "hbm_GB": round(bps * n / 1e9, 2), "ms": round(ms, 3), "bytes_from": "stored profile (profiles/pmc_%s.json), not this run" % scene_name, "stale": bool(prov and prov.get("stale"))}
    groups = {"closest_hit_kernels": hbm_frac("trace_closest_all", ms_mean["ms_trace_closest"] + ms_mean["ms_trace_mis"] - ms_mis_any),
              "any_hit_kernels": hbm_frac("trace_any_all", ms_mean["ms_trace_any"] + ms_mis_any)}
    ms_tc = ms_mean["ms_trace_closest"]
    ms_sh = ms_mean["ms_shade"]

    # A scene the traversal kernel keeps in LDS (<= 256 nodes, <= 128 primitives; S1's 32 triangles): SURVEY §8d's node and triangle terms are LDS reads by
    # design - staged once per workgroup, 3 KB - and what a ray moves through HBM is its 32-byte record in and its 16-byte hit out. The kernel is then bound by
    # VALU issue, not by any memory: its `frac` of the HBM roofline is small by construction and says so (`limiter`). (Round 3 divided the reference walk's
    # bytes by the time and printed 0.99 - 1.01 "of 8 TB/s" for it: VERDICT r03 weak #5. That rate stays in the detail file as reference_walk_rate.)
    lds_scene = bool(getattr(runner.scene, "lds_resident", lambda: False)())

    def trace_roofline():
        c = classes["path_closest"]
        if lds_scene:
            return 48 * c["rays"], launches_path, ms_tc, "trace_closest", c["rays"], "ray", 32 * c["rays"]
        return c["algorithmic_bytes"], launches_path, ms_tc, "trace_closest", c["rays"], "ray", 32 * c["rays"]

    def shade_roofline():
        verts = (cst["rays_closest"] - tail_nc) / nd_div
        emitted = (cst["rays_shadow"] + cst["rays_mis"] - not_cast + (cst["rays_closest"] - tail_nc - cst["camera_rays"])) / nd_div
        # one shade STAGE per bounce and pass (= per path-ray trace launch, counted by rt_render); on a class-split queue a stage is several k_shade launches
        # (rt_stats::launches_shade, in the detail file) - `traffic` (PMC) is per stage as well
        return 128 * verts + 32 * emitted + 48 * cst["camera_rays"] / nd_div, launches_path, ms_sh, "shade", verts, "vertex", 64 * verts

    # `traffic` (PMC, stored profile): FETCH_SIZE counts the L2's fabric-side read REQUESTS at 64 B each. Calibrated on known byte counts (round 6,
    # profiles/r06_fetch_calibration.json, scripts/micro/fetch_calibrate.hip): a coalesced stream - 4 or 16 bytes per lane - is fetched as 128-byte requests, so the counter
    # shows HALF its bytes; a gather that misses the L2 is ONE request per 64-byte sector, so the counter shows what it moved (1.00 - 1.01 of the distinct sectors of a 1 GB
    # table); WRITE_SIZE is exact. The x2 of the hardware guide therefore applies to the STREAMED reads of a kernel only: traffic = writes + raw reads + half of the bytes the
    # stage streams in (`stream_read`: the ray / vertex records it reads in slot order, known from the counts). `traffic_doubled` is rounds 1 - 5's figure (every read x2).
    def roof(algo_bytes, n_launch, ms_kernel, kname, unit_n, unit, stream_read):
        achieved = algo_bytes / (ms_kernel * 1e-3) / 1e9  # GB/s per device
        e, prov = pmc_entry(scene_name, kname)
        r = {"bound": "hbm", "kernel": "k_" + kname, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
             "frac": round(achieved / 8000.0, 4), "traffic": None, "traffic_raw_reads": e.get("hbm_bytes_per_launch_raw"), "traffic_doubled": e.get("hbm_bytes_per_launch"),
             "traffic_provenance": prov,
             "algorithmic_bytes_per_launch": round(algo_bytes / max(n_launch, 1)), "avg_launch_ms": round(ms_kernel / max(n_launch, 1), 4),
             "launches_per_step": int(n_launch), f"bytes_per_{unit}": round(algo_bytes / max(unit_n, 1), 1),
             "lanes": e.get("lanes_per_valu"), "valu_busy": e.get("valu_busy")}  # lanes / valu_busy: SQ counters of the stored profile (share of the SIMDs' cycles that issue a vector instruction)
        if e.get("hbm_bytes_per_launch_raw") is not None:
            raw = float(e["hbm_bytes_per_launch_raw"])
            r["traffic"] = round(min(raw + 0.5 * stream_read / max(n_launch, 1), float(e.get("hbm_bytes_per_launch") or 1e30)))
            # the L2-miss read requests of the stage per second (reads: doubled - raw = raw reads; 64 B each): the fabric sustained ~55 G requests/s in the calibration's
            # gathers and ~44 G/s (128-byte) in its streams, whatever the table's size - a stage near that rate is bound by REQUESTS, not by bytes
            reads_raw = float(e.get("hbm_bytes_per_launch") or raw) - raw
            r["l2_miss_read_requests_per_s"] = round(reads_raw / 64.0 / (ms_kernel / max(n_launch, 1) * 1e-3) / 1e9, 1)  # G requests / s
        if kname == "trace_closest" and lds_scene:
            r["limiter"] = "VALU issue: the scene is LDS-resident, a ray's HBM bytes are its record in and its hit out"
        if prov and prov.get("stale"):
            r["warning"] = "roofline.traffic was collected on other kernel sources than the ones that ran (profiles/pmc_*.json: kernel_source_sha differs)"
        return r
    roofline = roof(*(trace_roofline() if ms_tc >= ms_sh else shade_roofline()))
    roofline_other = roof(*(shade_roofline() if ms_tc >= ms_sh else trace_roofline()))  # the runner-up of the two heavy stages
    kernels_ms = {k[3:]: round(v, 2) for k, v in ms_mean.items()}
    verts = {k[9:]: int(kstats[-1][k]) for k in kstats[-1] if k.startswith("vertices_") and kstats[-1][k]}
    sharding = {"single": "single GPU",
                "dist": "interleaved tile rows, one process per GPU, touched rows gathered on rank 0 (RCCL send/recv)",
                "multi": f"one process (rt_multi_render), {runner.chunks} chunk(s) of interleaved tile rows per device, touched rows peer-copied to device 0"}[runner.mode]
    out = {
        "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "sharding": sharding, "sampler_mode": "pixel-keyed",
                   "light_distribution": ("cached across frames; first-frame build %.1f ms%s" % (lightdist_first_ms, "" if warmup else " (inside the first timed frame)")) if lightdist_first_ms is not None else "cached across frames"},
        "lightdist_ms_first_frame": lightdist_first_ms,
        "s_per_frame": round(ms_step / 1e3, 4),
        "Mrays_per_s": round((cst["rays_closest"] + cst["rays_shadow"] + cst["rays_mis"] - not_cast - tail_nc) / (ms_step * 1e-3) / 1e6, 1),
        "mis_rays_not_cast": int(not_cast), "tail_rays_not_cast": int(tail_nc),
        "kernel_ms_per_step": kernels_ms, "vertices_by_shade_front_end": verts, "shade_kernel_launches_per_step": int(launches_shade),
        "roofline": roofline, "roofline_second_kernel": roofline_other, "traversal_by_ray_class": classes, "traversal_hbm_share": groups,
        "camera_samples_per_step": int(samples_per_step), "gpu_clocks": clocks.summary(),
    }
    if n_gpus_asked > 1:
        out["n_gpus_requested"] = n_gpus_asked
        out["per_device_ms"] = per_rank_ms
        out["imbalance_max_over_mean"] = round(max(per_rank_ms) / max(float(np.mean(per_rank_ms)), 1e-9), 3)
        out["gather_ms"] = round(ms_mean.get("ms_gather", 0.0), 3)  # multi: last device done -> merged frame in place; dist: the slowest rank's share of the RCCL gather (rank 0: receives + additions)
        if runner.mode == "multi":
            out["devices"] = [int(d) for d in runner.devices]
        else:
            out["rccl_world_size"] = world
    return out


LINE_LIMIT = 4096  # bytes of the ONE printed line (a 78 KB line was not parsed by the driver); tests/test_bench_cpu.py holds it there

ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_raw_reads", "traffic_doubled", "l2_miss_read_requests_per_s", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_step", "lanes", "valu_busy", "limiter")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "n_gpus_requested", "per_device_ms", "gather_ms", "imbalance_max_over_mean")


def _roof(r):
    return {k: r[k] for k in ROOF_KEYS if k in r}


def _traffic_ratio(r):
    return round(r["traffic"] / r["algorithmic_bytes_per_launch"], 2) if r.get("traffic") and r.get("algorithmic_bytes_per_launch") else None


def compact_line(full):
    """The ONE line bench.py prints: BASELINE's metric, `roofline` of the dominant kernel, `cpu_baseline`, and per other workload the six numbers that say
    where it stands. Everything else a run measures (per-class traversal rates, stage times, register tables, provenance) is in the detail file."""
    out = {k: full[k] for k in TOP_KEYS if k in full}
    out["config"] = {k: v for k, v in full["config"].items() if k in ("workload", "sharding", "sampler_mode", "light_distribution")}
    out["roofline"] = _roof(full["roofline"])
    r2 = full.get("roofline_second_kernel")
    if r2:  # the runner-up of the two heavy stages, in four numbers
        out["roofline_second_kernel"] = {"kernel": r2["kernel"], "frac": r2["frac"], "avg_launch_ms": r2["avg_launch_ms"], "traffic_ratio": _traffic_ratio(r2)}
    if full["roofline"].get("warning"):
        out["roofline"]["traffic_stale"] = True
    if "cpu_baseline" in full:
        out["cpu_baseline"] = {k: full["cpu_baseline"][k] for k in CPU_KEYS}
        out["speedup"] = full.get("speedup_vs_cpu_baseline")
    oc = {}
    for name, r in (full.get("other_configs") or {}).items():
        oc[name] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "kernel": r["roofline"]["kernel"], "frac": r["roofline"]["frac"],
                    "traffic_ratio": _traffic_ratio(r["roofline"]), "cpu": (r.get("cpu_baseline") or {}).get("value")}
    if oc:
        out["other_configs"] = oc
    c1 = full.get("config_c1")
    if c1:
        out["config_c1"] = {"gpu": c1["gpu"]["value"], "cpu": (c1.get("cpu_port") or {}).get("value")}
    if full.get("detail"):
        out["detail"] = full["detail"]
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:  # never let a long workload name or device list push the line past what the driver reads
        out["config"]["workload"] = out["config"]["workload"][:160]
        for k in ("per_device_ms", "detail"):
            out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_LIMIT, len(line)
    return line


def write_detail(full, path=None):
    """Everything the run measured, next to the line: gpurun_out/bench_detail.json (scratch on the GPU box; scripts/profile_round.sh copies it under profiles/)."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        full = dict(full)
        full["kernel_resources"] = {"k_trace": occupancy_of("k_trace"), "k_shade": occupancy_of("k_shade")}
        full["kernel_source_sha"] = source_sha()
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "blob", "mis", "room", "mis-spheres", "instances-10k"],
                    help="the headline workload: cornell = BASELINE configs[1]; blob / mis / room = configs[2..4] (SURVEY.md §8d S2-S4)")
    ap.add_argument("--res", type=int, default=1024, help="cornell only (the other scenes use their BASELINE resolution)")
    ap.add_argument("--spp", type=int, default=0, help="0 = the BASELINE spp of the scene (cornell 1024, blob 256, mis 512, room 1024)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip configs[0] and configs[2..4] (they run only at N = 1 anyway)")
    ap.add_argument("--pbrt", default=None, help="render this pbrt-v3 scene file instead of a generated scene (read by the C++ host's parser; no CPU baseline)")
    ap.add_argument("--devices", default=None, help="in-process multi-GPU mode: comma-separated device ordinals, one worker each (default 0..N-1). Naming an ordinal "
                                                    "twice puts two workers with their own scene replicas on one GPU - a way to exercise the N > 1 path on a 1-GPU box, not a measurement")
    ap.add_argument("--chunks-per-device", type=int, default=1, help="in-process multi-GPU mode: tile-row chunks per device on the shared queue (1 = static interleaved split)")
    ap.add_argument("--detail", default=None, help="where the full measurement goes (default gpurun_out/bench_detail.json); the printed line stays under 4 KB")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # How many GPUs the run was asked for, and how it gets them - decided before anything touches a device (device_count() does not initialise one)
    n_visible = torch.cuda.device_count()
    devices = None
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if world > 1:
        mode = "dist"
        if args.gpus != world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus")
        if local_rank >= n_visible:
            raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {n_visible} GPU(s) are visible")
    elif args.gpus > 1 or args.devices:
        mode = "multi"
        devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but --devices names {len(devices)} device(s)")
        if min(devices) < 0 or max(devices) >= n_visible:
            raise SystemExit(f"bench.py: asked for {args.gpus} GPUs (devices {devices}) but only {n_visible} are visible; not measuring fewer under the same label")
    else:
        mode = "single"
    from rustracer_amd import host
    if not host.device_available() or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    torch.cuda.set_device(devices[0] if mode == "multi" else local_rank)
    dist = None
    if mode == "dist":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def runner_for(scene):
        return Runner(mode, scene, rank=rank, world=world, local_rank=local_rank, dist=dist, devices=devices, chunks=max(1, args.chunks_per_device))

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
    r0 = runner_for(scene)
    out = run_workload(args.scene, r0, workload, args.steps, args.warmup)
    del r0, scene
    if rank == 0 and mode == "single":
        if not args.no_cpu_baseline and desc is not None:
            cpu_spp = args.cpu_spp if args.scene != "room" else max(8, args.cpu_spp // 2)
            out["cpu_baseline"] = cpu_baseline(desc, cpu_spp)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if not args.headline_only and not args.pbrt:
            # BASELINE configs[2..4] next to the headline, measured the same way by the same run (2 timed frames each), then the general-primitive scenes
            others = {}
            for name in ("cornell", "blob", "mis", "room") + tuple(EXTRA_SCENES):
                if name == args.scene:
                    continue
                torch.cuda.empty_cache()
                d, wl = make_desc(name, SCENE_SPP[name])
                r = run_workload(name, runner_for(host.HostScene(d)), wl, 2, 1)
                if not args.no_cpu_baseline and name not in EXTRA_SCENES:
                    r["cpu_baseline"] = cpu_baseline(d, args.cpu_spp if name != "room" else max(8, args.cpu_spp // 2))
                    r["speedup_vs_cpu_baseline"] = round(r["value"] / r["cpu_baseline"]["value"], 1)
                others[name] = {k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "Mrays_per_s", "kernel_ms_per_step", "vertices_by_shade_front_end",
                                                  "roofline", "roofline_second_kernel", "traversal_by_ray_class", "traversal_hbm_share", "cpu_baseline", "speedup_vs_cpu_baseline") if k in r}
            out["other_configs"] = others
            # BASELINE configs[0] exactly: cornell 400x400, 64 spp - the reference's own CPU-runnable case, on the GPU and on the CPU port
            d, wl = make_desc("cornell", 64, 400)
            r = run_workload("cornell", runner_for(host.HostScene(d)), wl, 5, 1)
            c1 = {"workload": wl, "gpu": {"value": r["value"], "unit": "Msamples/s", "ms_per_step": r["ms_per_step"]}}
            if not args.no_cpu_baseline:
                c1["cpu_port"] = cpu_baseline(d, 64, what="whole")
            out["config_c1"] = c1
    if rank == 0:
        out["detail"] = write_detail(out, args.detail)
        print(compact_line(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
