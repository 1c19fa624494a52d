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
