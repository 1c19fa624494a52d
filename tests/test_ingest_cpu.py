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


# ------------------------------------------------------------------------------------------------ PNG / TGA / HDR (rc/imageio.rs:16-132)
def _u16_to_u8(c):   # image 0.24 FromPrimitive<u16> for u8
    return ((c.astype(np.uint32) + 128) // 257).astype(np.uint8)


@pytest.mark.parametrize("color_type, depth", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_depth_filter_and_interlace(ingest, tmp_path, color_type, depth, interlace):
    rng = np.random.default_rng(100 * color_type + depth + (7 if interlace else 0))
    h, w = 13, 19   # odd sizes: partial bytes at low bit depths, ragged Adam7 passes
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    palette = rng.integers(0, 256, (1 << depth, 3)).astype(np.uint8) if color_type == 3 else None
    a = rng.integers(0, 1 << depth, (h, w, ch))
    a[3:9, 2:15] = a[3, 2]   # flat region: long matches for the DEFLATE back-references
    path = str(tmp_path / "t.png")
    ingest.write_png(path, a, color_type, depth, palette=palette, interlace=interlace, idat_split=97)
    got = ingest.read_image(path)
    if color_type == 3:
        want8 = palette[a[..., 0]]
    else:
        s = _u16_to_u8(a) if depth == 16 else (a * (255 // ((1 << depth) - 1))).astype(np.uint8) if depth < 8 else a.astype(np.uint8)
        want8 = s[..., :3] if ch >= 3 else np.repeat(s[..., :1], 3, -1)   # to_rgb8: grey replicated, alpha dropped
    assert got.shape == (h, w, 3) and np.array_equal(got, want8.astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_stored_fixed_and_dynamic_deflate_blocks(ingest, tmp_path, level):
    rng = np.random.default_rng(level)
    a = rng.integers(0, 256, (64, 80, 3))
    a[:, 40:] = a[:, :40]                      # level 0: stored blocks; 1: mostly fixed codes; 9: dynamic codes with long matches
    if level == 1:
        a = a[:2, :3]                          # tiny image: zlib emits a fixed-Huffman block
    path = str(tmp_path / "t.png")
    ingest.write_png(path, a, 2, 8, filters=(0,), level=level)
    assert np.array_equal(ingest.read_image(path), a.astype(np.float32) / np.float32(255.0))


def test_png_ancillary_chunks_are_skipped_and_damage_is_reported(ingest, host, tmp_path):
    a = np.arange(5 * 4 * 3).reshape(5, 4, 3) % 256
    path = str(tmp_path / "t.png")
    ingest.write_png(path, a, 2, 8, extra_chunks=((b"gAMA", (45455).to_bytes(4, "big")), (b"tEXt", b"Comment\0hello")))
    assert np.array_equal(ingest.read_image(path), a.astype(np.float32) / np.float32(255.0))   # gamma chunks do not touch the samples
    raw = bytearray(open(path, "rb").read())
    bad = bytearray(raw); bad[-20] ^= 0x55
    open(path, "wb").write(bad)
    with pytest.raises(host.BackendError, match="CRC|Adler|deflate"):
        ingest.read_image(path)
    open(path, "wb").write(raw[:40])
    with pytest.raises(host.BackendError, match="truncated|missing"):
        ingest.read_image(path)
    ingest.write_png(path, a, 2, 8, extra_chunks=((b"ABCD", b"critical"),))
    with pytest.raises(host.BackendError, match="critical"):
        ingest.read_image(path)


@pytest.mark.parametrize("kind", ["rgb", "rgba", "grey", "mapped"])
@pytest.mark.parametrize("rle", [False, True])
@pytest.mark.parametrize("top_origin", [False, True])
def test_tga_types_origins_and_run_lengths(ingest, tmp_path, kind, rle, top_origin):
    rng = np.random.default_rng(3)
    h, w = 9, 14
    palette = rng.integers(0, 256, (32, 3)).astype(np.uint8) if kind == "mapped" else None
    if kind in ("rgb", "rgba"):
        a = rng.integers(0, 256, (h, w, 3 if kind == "rgb" else 4)).astype(np.uint8)
        a[2:5] = a[2, 0]          # runs that cross scanlines
        want = a[..., :3]
    else:
        a = rng.integers(0, 32, (h, w)).astype(np.uint8)
        a[2:5] = 7
        want = palette[a] if kind == "mapped" else np.repeat(a[..., None], 3, -1)
    path = str(tmp_path / "t.tga")
    ingest.write_tga(path, a, kind, rle=rle, top_origin=top_origin, palette=palette, id_field=b"id!")
    assert np.array_equal(ingest.read_image(path), want.astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr(ingest, tmp_path, rle):
    rng = np.random.default_rng(11)
    h, w = 6, 40
    a = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    a[..., 3] = rng.integers(120, 140, (h, w))
    a[2, 5:30] = a[2, 5]          # runs
    a[4, 3] = (9, 9, 9, 0)        # exponent 0 -> black
    path = str(tmp_path / "t.hdr")
    ingest.write_hdr(path, a, rle=rle)
    e = np.exp2(a[..., 3:].astype(np.float32) - np.float32(136.0))
    want = np.where(a[..., 3:] == 0, np.float32(0), e * a[..., :3].astype(np.float32)).astype(np.float32)   # Rgbe8Pixel::to_hdr
    assert np.array_equal(ingest.read_image(path), want)


def test_read_image_extension_rules(ingest, host, tmp_path):
    # imageio.rs:19-32: no extension / unknown extension are errors; exr is refused by this host layer, loudly
    for name, msg in (("noext", "doesn.t have an extension"), ("x.jpg", "Unsupported file format"), ("missing.exr", "cannot open"), ("missing.png", "cannot open")):
        with pytest.raises(host.BackendError, match=msg):
            ingest.read_image(str(tmp_path / name))
    a = np.full((2, 2, 3), 200, np.uint8)
    ingest.write_png(str(tmp_path / "upper.PNG"), a, 2, 8)
    assert ingest.read_image(str(tmp_path / "upper.PNG"))[0, 0, 0] == np.float32(200) / np.float32(255)
    ingest.write_pfm(str(tmp_path / "f.pfm"), a.astype(np.float32))
    assert np.array_equal(ingest.read_image(str(tmp_path / "f.pfm")), a.astype(np.float32))


@pytest.mark.parametrize("compression", ["none", "rle", "zips", "zip", "piz", "pxr24"])
@pytest.mark.parametrize("pixel_type", ["half", "float"])
def test_openexr_scanline_files(ingest, tmp_path, compression, pixel_type):
    rng = np.random.default_rng(21)
    h, w = 37, 29   # not a multiple of the 16-line ZIP / 32-line PIZ block
    a = (rng.random((h, w, 3), dtype=np.float32) * 4.0).astype(np.float32)
    a[5:20, 3:25] = a[5, 3]                          # flat region: long runs after the delta predictor / zero wavelet coefficients
    a[0, 0] = (0.0, 6.1e-5, 3.0e-6)                  # smallest normal half and a subnormal half
    a[1, 1] = (65504.0, np.inf, -2.5)
    if compression == "pxr24":
        a = (a.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)   # PXR24 keeps 24 bits of a float: the test image is representable in them
    path = str(tmp_path / "t.exr")
    ingest.write_exr(path, a, compression, pixel_type, alpha=True, origin=(-3, 7), extra_channels=("Z",))
    want = a.astype(np.float16).astype(np.float32) if pixel_type == "half" else a
    assert np.array_equal(ingest.read_image(path), want)


def test_openexr_piz_details(ingest, tmp_path):
    """PIZ on images that take its separate paths: few distinct values (14-bit wavelet), all 65536 half patterns' worth of values (16-bit wavelet),
    run-length symbols on and off, uint samples, an all-zero block, one pixel, one row, one column."""
    rng = np.random.default_rng(5)
    few = rng.integers(0, 7, (40, 33, 3)).astype(np.float32) * 0.25
    for runs in (True, False):
        ingest.write_exr(str(tmp_path / "few.exr"), few, "piz", "half", piz_runs=runs)
        assert np.array_equal(ingest.read_image(str(tmp_path / "few.exr")), few)
    many = rng.integers(0, 0x7c00, (64, 70, 3)).astype(np.uint16).view(np.float16).astype(np.float32)   # > 2^14 distinct half patterns in one block
    ingest.write_exr(str(tmp_path / "many.exr"), many, "piz", "half")
    assert np.array_equal(ingest.read_image(str(tmp_path / "many.exr")), many)
    ints = rng.integers(0, 100000, (35, 18, 3)).astype(np.float32)
    ingest.write_exr(str(tmp_path / "u.exr"), ints, "piz", "uint", extra_channels=("id",))
    assert np.array_equal(ingest.read_image(str(tmp_path / "u.exr")), ints)
    for shape in ((33, 20), (1, 1), (1, 50), (45, 1)):
        z = np.zeros(shape + (3,), np.float32)
        ingest.write_exr(str(tmp_path / "z.exr"), z, "piz", "float")
        assert np.array_equal(ingest.read_image(str(tmp_path / "z.exr")), z)
        r = rng.random(shape + (3,), dtype=np.float32)
        ingest.write_exr(str(tmp_path / "r.exr"), r, "piz", "float")
        assert np.array_equal(ingest.read_image(str(tmp_path / "r.exr")), r)


def _b44_model(a, flat_ok, plinear=False):
    """What a B44 file of the half image `a` (h, w, 3) decodes to, cell by cell through the test writer's packer and the model unpacker (rustracer_amd.ingest):
    32-line blocks, 4 x 4 cells whose missing rows / columns repeat the last one; pLinear channels go through exp(x / 8) and come back through 8 ln(x)."""
    from rustracer_amd.ingest import _b44_pack, b44_unpack
    hb = a.astype(np.float16).view(np.uint16)
    h, w = hb.shape[:2]
    out = np.zeros_like(hb)
    for c in range(3):
        for by in range(0, h, 32):
            nrows = min(32, h - by)
            for y in range(0, nrows, 4):
                ys = [by + min(y + k, nrows - 1) for k in range(4)]
                for x in range(0, w, 4):
                    xs = [min(x + k, w - 1) for k in range(4)]
                    cell = hb[np.ix_(ys, xs, [c])].reshape(-1)
                    if plinear:
                        f = cell.view(np.float16).astype(np.float64)
                        cell = np.where(np.isfinite(f), np.where(f >= 8 * np.log(65504.0), 65504.0, np.exp(np.minimum(f, 100.0) / 8)), 0.0).astype(np.float16).view(np.uint16)
                    dec = b44_unpack(_b44_pack(cell, flat_ok))
                    if plinear:
                        f = dec.view(np.float16).astype(np.float64)
                        with np.errstate(divide="ignore", invalid="ignore"):
                            dec = np.where(np.isfinite(f) & (f >= 0), 8.0 * np.log(f), 0.0).astype(np.float32).astype(np.float16).view(np.uint16)
                    for k, yy in enumerate(ys):
                        for j, xx in enumerate(xs):
                            out[yy, xx, c] = dec[4 * k + j]
    return out.view(np.float16).astype(np.float32)


@pytest.mark.parametrize("compression", ["b44", "b44a"])
def test_openexr_b44_files(ingest, tmp_path, compression):
    """B44 / B44A (lossy, fixed rate: exr 1.4.2 reads them, rc/imageio.rs:134-177). The decoder against the model of the format: the same half bit patterns,
    and those within the format's error of the original - smooth cells, noisy cells (large shifts), negative values, a flat region (3-byte cells in B44A),
    edge cells, a FLOAT channel beside the half ones (stored raw), tiles."""
    rng = np.random.default_rng(44)
    h, w = 45, 38                                                          # two 32-line blocks, clipped cells on both edges
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([0.5 + 0.4 * np.sin(xx / 5.0) * np.cos(yy / 7.0), rng.random((h, w)) * 100.0, -2.0 + 0.01 * xx - 0.02 * yy], -1).astype(np.float32)
    a[8:24, 4:20] = (0.25, 3.0, -1.0)                                      # flat cells
    a[30, 30] = (0.0, 65504.0, 1e-7)
    path = str(tmp_path / "b.exr")
    ingest.write_exr(path, a, compression, "half", alpha=True, origin=(2, -5))
    got = ingest.read_image(path)
    want = _b44_model(a, compression == "b44a")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    err0 = np.abs(got[..., 0] - a[..., 0]); err0[28:32, 28:32] = 0                # (the cell of the planted 0.0 spans 0 .. 0.5: a coarse shift)
    assert err0.max() < 4e-3 and np.array_equal(got[8:24, 4:20], np.broadcast_to(np.float32([0.25, 3.0, -1.0]), (16, 16, 3)))
    size_a = len(open(path, "rb").read())
    if compression == "b44a":
        ingest.write_exr(str(tmp_path / "b0.exr"), a, "b44", "half", alpha=True, origin=(2, -5))
        assert size_a < len(open(str(tmp_path / "b0.exr"), "rb").read())   # the flat cells took 3 bytes
    # pLinear channels
    pos = np.abs(a) + 0.01
    ingest.write_exr(path, pos, compression, "half", plinear=("R", "G", "B"))
    got = ingest.read_image(path)
    assert np.array_equal(got.view(np.uint32), _b44_model(pos, compression == "b44a", plinear=True).view(np.uint32))
    ep = np.abs(got[..., 0] - pos[..., 0]); ep[28:32, 28:32] = 0                  # exp(x / 8) as a half near 1 resolves x to 8 * 2^-10: an absolute error
    assert ep.max() < 0.012
    # float samples are stored raw by B44; tiles hold cells of their own
    ingest.write_exr(path, a, compression, "float")
    assert np.array_equal(ingest.read_image(path), a)
    ingest.write_exr(path, a, compression, "half", tiles=(16, 12))
    t = ingest.read_image(path)
    e0 = np.abs(t[..., 0] - a[..., 0]); e0[24:36, 16:32] = 0                     # (the tile cell that holds the planted 0.0)
    e2 = np.abs(t[..., 2] - a[..., 2]); e2[24:36, 16:32] = 0
    assert e0.max() < 4e-3 and e2.max() < 8e-3


@pytest.mark.parametrize("compression", ["none", "zip", "piz", "rle"])
@pytest.mark.parametrize("level_mode", ["one", "mipmap"])
def test_openexr_tiled_files(ingest, tmp_path, compression, level_mode):
    a = np.random.default_rng(8).random((45, 70, 3), dtype=np.float32)
    path = str(tmp_path / "tiled.exr")
    ingest.write_exr(path, a, compression, "half", tiles=(32, 16), level_mode=level_mode, alpha=True, origin=(5, -2))   # edge tiles are clipped; lower levels hold other values
    assert np.array_equal(ingest.read_image(path), a.astype(np.float16).astype(np.float32))


def test_openexr_multi_part_files(ingest, tmp_path):
    a = np.random.default_rng(9).random((21, 17, 3), dtype=np.float32)
    for tiles in (None, (8, 8)):
        path = str(tmp_path / "parts.exr")
        ingest.write_exr(path, a, "zip", "float", parts_before=(("Z",), ("A", "Y")), tiles=tiles)   # the first part WITH R, G and B is the third one
        assert np.array_equal(ingest.read_image(path), a)


def test_openexr_patched_chunk_count_is_refused(ingest, host, tmp_path):
    """a multi-part file states each part's chunkCount; one that names fewer chunks than the data window needs must fail as a truncated offset
    table, not index past the table (tiled parts) or leave rows black (scan-line parts)."""
    a = np.random.default_rng(10).random((21, 17, 3), dtype=np.float32)
    for tiles in (None, (8, 8)):
        path = str(tmp_path / "patched.exr")
        ingest.write_exr(path, a, "none", "float", parts_before=(("Z",),), tiles=tiles)
        raw = bytearray(open(path, "rb").read())
        key = b"chunkCount\0int\0" + (4).to_bytes(4, "little")
        i = raw.index(key, raw.index(key) + 1) + len(key)          # the second part's (the RGB part's) chunkCount
        n = int.from_bytes(raw[i:i + 4], "little")
        assert n == (21 if tiles is None else 9)
        raw[i:i + 4] = (1).to_bytes(4, "little")
        open(path, "wb").write(raw)
        with pytest.raises(host.BackendError, match="truncated offset table"):
            ingest.read_image(path)


def test_openexr_b44_flat_cell_is_any_shift_of_13_or_more(ingest, tmp_path):
    """a B44 cell whose third byte is >= 13 << 2 is a 3-byte flat cell whatever its low bits; this writer emits 0xfc, other encoders need not."""
    a = np.full((8, 8, 3), 0.25, np.float32)
    path = str(tmp_path / "flat.exr")
    ingest.write_exr(path, a, "b44a", "half")
    want = ingest.read_image(path)
    raw = bytearray(open(path, "rb").read())
    hits = [i for i in range(len(raw) - 2) if raw[i + 2] == 0xfc and raw[i:i + 2] == raw[i + 3:i + 5] and i + 5 < len(raw) and raw[i + 5] == 0xfc]
    assert hits, "no run of flat cells found in the block"
    k = hits[0]
    while k + 2 < len(raw) and raw[k + 2] == 0xfc and raw[k:k + 2] == raw[hits[0]:hits[0] + 2]:
        raw[k + 2] = 0x34 | (k & 3)   # shift 13 with arbitrary low bits
        k += 3
    open(path, "wb").write(raw)
    assert np.array_equal(ingest.read_image(path), want)


def test_openexr_named_layer_and_refusals(ingest, host, tmp_path):
    a = np.random.default_rng(3).random((8, 8, 3), dtype=np.float32)
    path = str(tmp_path / "layer.exr")
    ingest.write_exr(path, a, "zip", "float", layer="diffuse.")
    assert np.array_equal(ingest.read_image(path), a)                              # first layer that has R, G and B
    raw = bytearray(open(path, "rb").read())
    i = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    for code, word in ((8, "DWA"), (9, "DWA")):                                   # the DCT compressions are refused by name
        raw[i] = code
        open(path, "wb").write(raw)
        with pytest.raises(host.BackendError, match=word):
            ingest.read_image(path)
    raw[i] = 3; raw[4:8] = (2 | 0x800).to_bytes(4, "little")                       # deep data flag
    open(path, "wb").write(raw)
    with pytest.raises(host.BackendError, match="deep"):
        ingest.read_image(path)
    raw[4:8] = (2 | 0x200).to_bytes(4, "little")                                   # tiled flag without a tile description
    open(path, "wb").write(raw)
    with pytest.raises(host.BackendError, match="tile"):
        ingest.read_image(path)
    open(path, "wb").write(b"not an exr file at all")
    with pytest.raises(host.BackendError, match="magic"):
        ingest.read_image(path)


def test_damaged_image_and_mesh_files_fail_cleanly(ingest, host, tmp_path):
    """Random damage and truncation of valid PNG / TGA / HDR / EXR / PFM / PLY files: an error or a decoded result, never a crash."""
    import random
    rng = np.random.default_rng(1)
    a8 = rng.integers(0, 256, (9, 11, 3))
    ingest.write_png(str(tmp_path / "a.png"), a8, 2, 8)
    ingest.write_png(str(tmp_path / "b.png"), rng.integers(0, 4, (9, 11, 1)), 3, 2, palette=rng.integers(0, 256, (4, 3)), interlace=True)
    ingest.write_tga(str(tmp_path / "a.tga"), a8.astype(np.uint8), "rgb", rle=True)
    ingest.write_hdr(str(tmp_path / "a.hdr"), rng.integers(0, 256, (6, 40, 4)).astype(np.uint8), rle=True)
    ingest.write_exr(str(tmp_path / "a.exr"), rng.random((20, 9, 3), dtype=np.float32), "zip", "half")
    ingest.write_exr(str(tmp_path / "b.exr"), rng.random((5, 9, 3), dtype=np.float32), "rle", "float")
    ingest.write_exr(str(tmp_path / "c.exr"), rng.integers(0, 9, (34, 12, 3)).astype(np.float32), "piz", "half")
    ingest.write_exr(str(tmp_path / "d.exr"), rng.random((20, 20, 3), dtype=np.float32), "pxr24", "half", tiles=(8, 8), parts_before=(("Z",),))
    ingest.write_pfm(str(tmp_path / "a.pfm"), rng.random((4, 5, 3), dtype=np.float32))
    ingest.write_ply(str(tmp_path / "a.ply"), rng.random((6, 3), dtype=np.float32), [[0, 1, 2], [3, 4, 5, 0]])
    ingest.write_ply(str(tmp_path / "b.ply"), rng.random((6, 3), dtype=np.float32), [[0, 1, 2], [3, 4, 5]], fmt="ascii")
    rnd = random.Random(7)
    for name in sorted(p.name for p in tmp_path.iterdir()):
        data = (tmp_path / name).read_bytes()
        ext = name.split(".")[-1]
        for _ in range(40):
            b = bytearray(data)
            if rnd.random() < 0.25:
                b = b[: rnd.randrange(len(b))]
            for _ in range(rnd.choice([1, 2, 4, 16])):
                if b:
                    b[rnd.randrange(len(b))] = rnd.randrange(256)
            if rnd.random() < 0.2 and len(b) > 8:
                i = rnd.randrange(len(b) - 4)
                b[i:i + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f"])
            p = str(tmp_path / ("fz." + ext))
            open(p, "wb").write(b)
            try:
                ingest.read_ply(p) if ext == "ply" else ingest.read_image(p)
            except host.BackendError:
                pass
