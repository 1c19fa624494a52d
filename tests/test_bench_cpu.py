"""bench.py's handling of --gpus: a run that cannot get the GPUs it was asked for stops, it never measures fewer under the same label."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_more_gpus_than_visible_is_an_error():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and f"asked for {n + 2} GPUs" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line


def test_gpus_must_match_the_torchrun_world_size():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_device_list_must_match_gpus():
    r = _run(["--gpus", "3", "--devices", "0,0"])
    assert r.returncode != 0 and "--devices names 2" in r.stderr


def _fake_workload(name, long_strings=False):
    """The dict shape run_workload returns (bench.py), with the widest values a real run produces."""
    roof = {"bound": "hbm", "kernel": "k_shade", "achieved": 1919.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.2399, "traffic": 50812345678,
            "traffic_raw_reads": 31234567890, "traffic_provenance": {"file": "profiles/pmc_x.json", "from": "y" * 60, "collected": "r04", "method": "m" * 150,
                                                                      "kernel_source_sha": "0" * 16, "stale": True},
            "algorithmic_bytes_per_launch": 30207123456, "avg_launch_ms": 15.7234, "launches_per_step": 24, "bytes_per_vertex": 187.3, "lanes": 44.4,
            "warning": "w" * 150}
    cls = {"rays": 10 ** 10, "algorithmic_bytes": 10 ** 13, "ms": 344.123, "reference_walk_GB_per_s": 7000.1, "reference_walk_rate": 0.8751, "nodes_per_ray": 17.31, "tris_per_ray": 2.35}
    return {"metric": "Msamples/s", "value": 1115.12, "unit": "Msamples/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 962.91, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name} " + "x" * (900 if long_strings else 120), "sharding": "s" * 110, "sampler_mode": "pixel-keyed"},
            "s_per_frame": 0.9629, "Mrays_per_s": 9876.5, "mis_rays_not_cast": 0, "kernel_ms_per_step": {f"stage_{i}": 123.45 for i in range(14)},
            "vertices_by_shade_front_end": {"lambert_const": 10 ** 10}, "roofline": roof, "roofline_second_kernel": dict(roof, kernel="k_trace_closest"),
            "traversal_by_ray_class": {k: cls for k in ("path_closest", "shadow_any", "mis_closest", "mis_any")},
            "traversal_hbm_share": {"closest_hit_kernels": None, "any_hit_kernels": None}, "camera_samples_per_step": 2 ** 30,
            "n_gpus_requested": 8, "per_device_ms": [962.91] * 8, "imbalance_max_over_mean": 1.059, "gather_ms": 1.234, "devices": list(range(8))}


def test_the_printed_line_stays_under_4_kb_and_carries_roofline_and_cpu_baseline():
    """a 78 KB line left BENCH_r03.json unparsed. Whatever a run measures, the ONE printed line is bounded; the rest goes to the detail file."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    cpu = {"value": 9.712, "unit": "Msamples/s", "cores": 16, "kind": "port", "sample": "same scene and resolution at 32 spp (33554432 camera samples, 3.5 s); Msamples/s is spp-independent",
           "build": "f" * 120, "Mrays_per_s": 99.1}
    for long_strings in (False, True):
        full = _fake_workload("cornell", long_strings)
        full["cpu_baseline"], full["speedup_vs_cpu_baseline"] = cpu, 114.8
        full["other_configs"] = {n: dict(_fake_workload(n), cpu_baseline=cpu, speedup_vs_cpu_baseline=85.5) for n in ("blob", "mis", "room", "mis-spheres", "instances-10k")}
        full["config_c1"] = {"workload": "w" * 120, "gpu": {"value": 1002.4, "unit": "Msamples/s", "ms_per_step": 10.2}, "cpu_port": cpu}
        full["detail"] = "gpurun_out/bench_detail.json"
        line = bench.compact_line(full)
        assert len(line) < 4096 and "\n" not in line
        j = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in j, k
        assert j["roofline"]["frac"] == 0.2399 and j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0 and j["roofline"]["traffic"] == 50812345678
        assert j["cpu_baseline"] == {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample")} and j["speedup"] == 114.8
        assert set(j["other_configs"]) == {"blob", "mis", "room", "mis-spheres", "instances-10k"}
        assert j["other_configs"]["room"] == {"value": 1115.12, "ms_per_step": 962.91, "kernel": "k_shade", "frac": 0.2399, "traffic_ratio": 1.68, "cpu": 9.712}
        assert "workload" in j["config"] and "model" not in j["config"]
        assert "occupancy" not in line and "traversal_by_ray_class" not in line


def test_detail_file_holds_what_the_line_leaves_out(tmp_path):
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _fake_workload("cornell")
    rel = bench.write_detail(full, str(tmp_path / "d.json"))
    j = json.load(open(tmp_path / "d.json"))
    assert rel and set(j["traversal_by_ray_class"]) == {"path_closest", "shadow_any", "mis_closest", "mis_any"} and "kernel_resources" in j and len(j["kernel_source_sha"]) == 16
