"""Random scenes through the FILE path, no GPU needed: SceneDesc -> rustracer_amd.pbrt_export.write_pbrt -> the C++ loader (rtx_pbrt.inl) against the same SceneDesc handed
over call by call - geometry, per-triangle tables, the BVH node for node, camera / film / sampler / integrator parameters, every material and texture a triangle names, the
light list in order, object definitions and instances (tests/test_pbrt_cpu.py assert_same_scene). The scenes are those of scripts/fuzz_shading.py (rooms over every material,
texture and light class) and scripts/fuzz_objects.py (two-level scenes whose objects hold quadrics and masked meshes).
    python scripts/fuzz_pbrt.py [n_scenes=60] [seed=1]          (FUZZ_DEVICE_INGEST=1 on the GPU box: the loader's device ingest path, meshes over 2000 triangles as PLY)
Scenes the reference's file format cannot say are counted apart (a float checkerboard: api.rs:1201-1216 has none)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))

from rustracer_amd import host  # noqa: E402
from rustracer_amd.pbrt_export import write_pbrt  # noqa: E402
import fuzz_objects  # noqa: E402
import fuzz_shading  # noqa: E402
import test_pbrt_cpu as T  # noqa: E402


DEVICE_INGEST = bool(os.environ.get("FUZZ_DEVICE_INGEST"))   # (GPU box: PLY / PFM decoding, MIP pyramids and the environment's distribution built on the device)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = tempfile.mkdtemp()
    total_bad = 0
    for name, gen in (("rooms", fuzz_shading.make_scene), ("objects", fuzz_objects.make_scene)):
        rng = np.random.default_rng(seed)
        ok, unsayable, bad = 0, 0, {}
        for k in range(n):
            d = gen(rng)[0]
            d.name = f"{name}{k}"
            try:
                path = os.path.join(tmp, f"{d.name}.pbrt")
                write_pbrt(d, path, ply_over=2000 if DEVICE_INGEST else 20000)
                p = host.PbrtScene(path, device_ingest=DEVICE_INGEST)
                assert p.n_warnings == 0, p.first_warning
                T.assert_same_scene(p, host.HostScene(d))
                if d.objects:   # the definitions themselves; the loader numbers an object when it is first instanced, the SceneDesc when it is defined
                    h = host.HostScene(d)
                    ip, ih = p.table("instances"), h.table("instances")
                    assert len(ip) == len(ih) and np.array_equal(ip["o2w"], ih["o2w"])
                    pairs = sorted(set(zip(ip["object"].tolist(), ih["object"].tolist())))
                    assert len({a for a, _ in pairs}) == len(pairs) == len({b for _, b in pairs}), "instances name other objects"
                    for op, oh in pairs:
                        for sub in ("P", "N", "UV", "S", "indices", "tri_flags"):
                            assert np.array_equal(p.table((op, sub)), h.table((oh, sub))), (oh, sub)
                        qa, qb = p.table((op, "quadrics")), h.table((oh, "quadrics"))
                        assert len(qa) == len(qb), (oh, "quadrics")
                        for f in (qa.dtype.names or ()):
                            if f in ("material", "light"):
                                continue   # (ids of another numbering: the file's order of definition)
                            assert np.allclose(qa[f], qb[f], rtol=1e-5, atol=1e-6) if f == "w2o" else np.array_equal(qa[f], qb[f]), (oh, "quadrics", f)
                        assert np.array_equal(p.table((op, "tri_alpha")) >= 0, h.table((oh, "tri_alpha")) >= 0), (oh, "tri_alpha")
                ok += 1
            except ValueError as e:
                if "no float checkerboard" in str(e):
                    unsayable += 1
                else:
                    bad.setdefault(f"ValueError: {str(e)[:160]}", []).append(k)
            except Exception as e:  # noqa: BLE001
                bad.setdefault(f"{type(e).__name__}: {str(e)[:160]}", []).append(k)
        print(f"{name}: {ok} of {n} scenes equal, {unsayable} not expressible in the reference's format, {sum(len(v) for v in bad.values())} different")
        for key, v in bad.items():
            print(f"   {len(v)} x {key}   (scenes {v[:8]})")
        total_bad += sum(len(v) for v in bad.values())
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
