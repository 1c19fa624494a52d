"""The randomised checks of scripts/fuzz_*.py on a few scenes each (the seeds the round's full runs used: profiles/r06_fuzz_*.txt), so that they run with the suite:
hit records, occlusion, light tables and films of random rooms; two-level scenes whose objects hold quadrics and masked meshes; the linear BVH builder; the
reference-stream mode. Each script exits non-zero on a mismatch and prints the scene."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("fuzz_shading.py", ("24", "1")), ("fuzz_objects.py", ("24", "1")), ("fuzz_lds_walks.py", ("24", "21")), ("fuzz_device_bvh.py", ("24", "1")),
                                         ("fuzz_ref_stream.py", ("24", "1"))])
def test_random_scenes(gpu_host, script, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *args], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])


def test_random_two_level_scenes_in_the_material_zoo(gpu_host):
    """scripts/fuzz_objects.py with FUZZ_RICH=1: the instanced objects' triangles and quadrics wear every material / texture class / bump maps of scripts/fuzz_shading.py
    (the interaction of a hit inside a rotated, scaled, mirrored instance carries dpdu / dpdv / the shading frame through the instance's transform into them)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_objects.py"), "24", "2"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=dict(os.environ, FUZZ_RICH="1"))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
