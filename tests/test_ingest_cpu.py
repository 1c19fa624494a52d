"""PLY / PFM readers of the host layer against files written by an independent numpy writer:
the synthetic scenes emitted in the reference's input formats come back bit-identical, and the reader keeps the
reference loader's rules (rc/shapes/plymesh.rs, rc/imageio.rs)."""
import numpy as np
import pytest

from util import bits


@pytest.fixture(scope="module")
def ingest(host):
    from rustracer_amd import ingest as _i
    return _i


@pytest.mark.parametrize("fmt", ["binary_little_endian", "binary_big_endian", "ascii"])
def test_ply_round_trip_of_the_blob_mesh(ingest, host, orc, tmp_path, fmt):
    from rustracer_amd.scenes.procedural import displaced_sphere
    P, idx, N, UV = displaced_sphere(48, 24, (0, 1.1, 0), 1.0, 0.18, 1234)
    path = tmp_path / "blob.ply"
    ingest.write_ply(path, P, idx, N, UV, fmt=fmt, uv_names=("s", "t") if fmt == "ascii" else ("u", "v"))
    m = ingest.read_ply(path)
    assert np.array_equal(m["idx"], idx) and m["dropped_faces"] == 0
    for k, ref in (("P", P), ("N", N), ("UV", UV)):
        assert np.array_equal(bits(m[k]), bits(ref)), k  # repr(float32 as float) round-trips exactly in ascii too
    # a scene built from the file has the same BVH as the one built from memory
    from rustracer_amd.scene_desc import SceneDesc
    bv = []
    for (p_, i_, n_, uv_) in ((P, idx, N, UV), (m["P"], m["idx"], m["N"], m["UV"])):
        s = SceneDesc(); s.add_mesh(p_, i_, s.matte((0.5, 0.5, 0.5)), N=n_, UV=uv_)
        bv.append(host.HostScene(s).bvh())
    assert all(np.array_equal(bv[0][k], bv[1][k]) for k in bv[0])


def test_ply_quads_polygons_and_ignored_properties(ingest, tmp_path):
    P = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 2, 2]], np.float32)
    path = tmp_path / "mixed.ply"
    ingest.write_ply(path, P, [[0, 1, 2, 3], [0, 1, 4], [0, 1, 2, 3, 4], [3, 2]], index_type="uint")
    m = ingest.read_ply(path)
    # quad (a b c d) -> (a b c) (d a c) (plymesh.rs:113-118); pentagon and 2-gon dropped (:104-107)
    assert m["idx"].tolist() == [[0, 1, 2], [3, 0, 2], [0, 1, 4]] and m["dropped_faces"] == 2
    assert m["N"] is None and m["UV"] is None
    # a double-typed coordinate is not a Property::Float: ignored, the coordinate stays 0 (plymesh.rs:196-199); extra
    # properties and short index lists are parsed and skipped
    with open(tmp_path / "odd.ply", "wb") as f:
        f.write(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty double y\nproperty float z\nproperty uchar red\n"
                b"element face 2\nproperty list uchar short vertex_indices\nproperty list uchar int vertex_indices\nend_header\n"
                b"1 5 2 255\n3 6 4 0\n5 7 6 9\n3 0 1 2 3 0 1 2\n3 0 1 2 3 2 1 0\n")
    o = ingest.read_ply(tmp_path / "odd.ply")
    assert o["P"].tolist() == [[1, 0, 2], [3, 0, 4], [5, 0, 6]]
    assert o["idx"].tolist() == [[0, 1, 2], [2, 1, 0]]


def test_ply_errors(ingest, host, tmp_path):
    with open(tmp_path / "noz.ply", "wb") as f:
        f.write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0 0\n3 0 0 0\n")
    with pytest.raises(host.BackendError, match="coordinate"):
        ingest.read_ply(tmp_path / "noz.ply")
    with open(tmp_path / "extra.ply", "wb") as f:
        f.write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\n"
                b"element edge 1\nproperty int a\nend_header\n0 0 0\n3 0 0 0\n1\n")
    with pytest.raises(host.BackendError, match="unexpected PLY element"):
        ingest.read_ply(tmp_path / "extra.ply")
    with pytest.raises(host.BackendError):
        ingest.read_ply(tmp_path / "missing.ply")
    with open(tmp_path / "trunc.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n\0\0\0\0")
    with pytest.raises(host.BackendError, match="truncated"):
        ingest.read_ply(tmp_path / "trunc.ply")


@pytest.mark.parametrize("little", [True, False])
def test_pfm_round_trip(ingest, tmp_path, little):
    from rustracer_amd.scenes.procedural import sky_image
    img = sky_image(64, 32)
    ingest.write_pfm(tmp_path / "sky.pfm", img, little_endian=little)
    back = ingest.read_pfm(tmp_path / "sky.pfm")
    assert back.shape == (32, 64, 3) and np.array_equal(bits(back), bits(img))  # row 0 = top after the reader's flip
    # grey + scale: samples are multiplied by |scale| (imageio.rs:228-230) and replicated to rgb
    g = np.arange(12, dtype=np.float32).reshape(3, 4)
    ingest.write_pfm(tmp_path / "g.pfm", g, little_endian=little, scale=2.0)
    b = ingest.read_pfm(tmp_path / "g.pfm")
    assert np.array_equal(b[..., 0], g * 2) and np.array_equal(b[..., 1], b[..., 0]) and np.array_equal(b[..., 2], b[..., 0])


def test_pfm_errors(ingest, host, tmp_path):
    (tmp_path / "bad.pfm").write_bytes(b"P6\n1 1\n-1\n\0\0\0\0")
    with pytest.raises(host.BackendError):
        ingest.read_pfm(tmp_path / "bad.pfm")
    (tmp_path / "short.pfm").write_bytes(b"PF\n2 2\n-1.0\n\0\0\0\0")
    with pytest.raises(host.BackendError, match="truncated"):
        ingest.read_pfm(tmp_path / "short.pfm")


def test_scene_from_ply_and_pfm_equals_the_in_memory_scene(ingest, host, orc, tmp_path):
    from rustracer_amd.scenes import blob_scene
    from rustracer_amd.scenes.procedural import checker_fbm_image
    a, b = blob_scene(32, 16, 16, 16, 1), blob_scene(32, 16, 16, 16, 1, via_ply=str(tmp_path / "b.ply"))
    ba, bb = host.HostScene(a).bvh(), host.HostScene(b).bvh()
    assert all(np.array_equal(ba[k], bb[k]) for k in ba)
    img = checker_fbm_image(16, 3)
    ingest.write_pfm(tmp_path / "t.pfm", img)
    from rustracer_amd.scene_desc import SceneDesc
    s1, s2 = SceneDesc(), SceneDesc()
    s1.add_mip(img); s2.add_pfm_mip(str(tmp_path / "t.pfm"))
    assert np.array_equal(s1.mipmaps[0].data, s2.mipmaps[0].data)
