"""bench.py's three ways of rendering a frame on the GPU box: one device, several workers of one process (rt_multi_render; two workers on the one GPU here),
and the JSON line of `--gpus 2 --devices 0,0`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_in_process_multi_gpu_runner_renders_the_single_gpu_film(gpu_host):
    sys.path.insert(0, ROOT)
    import bench
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(96, 80, 16)
    single = bench.Runner("single", gpu_host.HostScene(d))
    st1 = single.step(True)
    multi = bench.Runner("multi", gpu_host.HostScene(d), devices=[0, 0], chunks=2)
    st2 = multi.step(True)
    a, b = single.film.cpu().numpy(), multi.film.cpu().numpy()
    assert np.array_equal(a[..., 3], b[..., 3]) and np.allclose(a, b, rtol=1e-6, atol=0)
    assert st1["camera_rays"] == st2["camera_rays"] == 96 * 80 * 16
    assert len(multi.per_device) == 2 and sum(p["camera_rays"] for p in multi.per_device) == st1["camera_rays"]
    c1, c2 = single.count(), multi.count()
    for k in ("rays_closest", "nodes_closest", "tris_closest", "rays_shadow", "nodes_shadow", "rays_mis", "nodes_mis"):
        assert c1[k] == c2[k], k


def test_bench_line_of_the_in_process_multi_gpu_mode(tmp_path):
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--res", "128", "--spp", "16", "--steps", "1", "--warmup", "1",
                        "--headline-only", "--no-cpu-baseline", "--detail", detail], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096  # ONE line the driver can parse (78 KB were not)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["n_gpus_requested"] == 2 and len(j["per_device_ms"]) == 2
    assert j["value"] > 0 and j["roofline"]["frac"] > 0 and "gather_ms" in j and j["imbalance_max_over_mean"] >= 1.0
    d = json.load(open(detail))  # the rest of the measurement
    assert d["devices"] == [0, 0] and set(d["traversal_by_ray_class"]) == {"path_closest", "shadow_any", "mis_closest", "mis_any"}
    assert any("k_shade" in k for k in d["kernel_resources"]["k_shade"])


def test_default_bench_line_parses_and_is_short(tmp_path):
    """The command the driver runs, at a size that takes seconds: one line < 4 KB with roofline, cpu_baseline and the other configs' summary."""
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--res", "128", "--spp", "16", "--steps", "1", "--warmup", "0", "--cpu-spp", "1", "--headline-only",
                        "--detail", detail], capture_output=True, text=True, timeout=900, env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1 and j["roofline"]["peak"] == 8000.0
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1 and j["speedup"] > 0


def test_counting_as_rendered_walks_environment_mis_rays_as_occlusion_rays(gpu_host):
    from rustracer_amd.scenes import room_env
    d = room_env(96, 64, 8, detail=1, tex_size=64, env_size=128)
    h = gpu_host.HostScene(d)
    _, ref = h.render(count_traversal=True)                           # the reference's walk: every MIS ray closest-hit
    _, asr = h.render(count_traversal=True, count_as_rendered=True)   # what an uncounted frame walks
    assert ref["rays_mis_any"] == 0 and asr["rays_mis_any"] > 0
    assert asr["rays_mis"] == ref["rays_mis"] and asr["rays_closest"] == ref["rays_closest"]
    assert asr["nodes_mis"] < ref["nodes_mis"]                        # an any-hit walk leaves at the first hit
    assert asr["nodes_shadow"] == ref["nodes_shadow"]
    # ... and does not cast the path rays at the depth limit that nothing reads (rays_tail_not_cast, still part of rays_closest): their node visits are the difference
    assert asr["rays_tail_not_cast"] > 0 and ref["rays_tail_not_cast"] == 0 and asr["nodes_closest"] < ref["nodes_closest"]
    assert (ref["nodes_closest"] - asr["nodes_closest"]) < 4 * ref["nodes_closest"] * asr["rays_tail_not_cast"] / ref["rays_closest"]
