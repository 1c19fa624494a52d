"""Input formats on the caller's side of the path: PLY meshes; PFM, PNG, TGA and Radiance HDR images.

Readers are the C++ host's (`rtxh_ply_read`, `rtxh_pfm_read`, `rtxh_image_read` in include/rtx_host.h), restating
plymesh::create (rc/shapes/plymesh.rs:18-186), read_image_pfm (rc/imageio.rs:179-246) and read_image (rc/imageio.rs:16-132).
The writers here exist to emit the synthetic scenes and test images in those formats; they are plain numpy / zlib and
share no code with the readers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import host


class _Ply(C.Structure):
    _fields_ = [("n_verts", C.c_int32), ("n_tris", C.c_int32), ("P", C.POINTER(C.c_float)), ("N", C.POINTER(C.c_float)),
                ("UV", C.POINTER(C.c_float)), ("idx", C.POINTER(C.c_int32)), ("n_dropped_faces", C.c_int32)]


def read_ply(path: str) -> dict:
    """-> dict(P (nv,3) f32, idx (nt,3) i32, N (nv,3)|None, UV (nv,2)|None, dropped_faces)."""
    L = host.lib()
    L.rtxh_ply_read.argtypes = [C.c_char_p, C.POINTER(_Ply)]
    L.rtxh_ply_free.argtypes = [C.POINTER(_Ply)]
    L.rtxh_ply_free.restype = None
    ply = _Ply()
    rc = L.rtxh_ply_read(str(path).encode(), C.byref(ply))
    if rc != 0:
        raise host.BackendError(f"rtxh_ply_read failed ({rc}): {L.rtxh_last_error().decode(errors='replace')}")
    try:
        nv, nt = ply.n_verts, ply.n_tris
        out = dict(P=np.ctypeslib.as_array(ply.P, (nv, 3)).copy(), idx=np.ctypeslib.as_array(ply.idx, (nt, 3)).copy() if nt else np.zeros((0, 3), np.int32),
                   N=np.ctypeslib.as_array(ply.N, (nv, 3)).copy() if ply.N else None, UV=np.ctypeslib.as_array(ply.UV, (nv, 2)).copy() if ply.UV else None,
                   dropped_faces=int(ply.n_dropped_faces))
    finally:
        L.rtxh_ply_free(C.byref(ply))
    return out


def write_ply(path: str, P, faces, N=None, UV=None, fmt: str = "binary_little_endian", uv_names=("u", "v"), index_type: str = "int") -> None:
    """faces: (n,3) / (n,4) array or a list of index lists (mixed triangles / quads / other polygons)."""
    P = np.asarray(P, np.float32)
    cols, names = [P], ["x", "y", "z"]
    if N is not None:
        cols.append(np.asarray(N, np.float32)); names += ["nx", "ny", "nz"]
    if UV is not None:
        cols.append(np.asarray(UV, np.float32)); names += list(uv_names)
    V = np.concatenate(cols, axis=1)
    flist = [list(map(int, f)) for f in faces]
    hdr = ["ply", f"format {fmt} 1.0", "comment written by rustracer_amd.ingest.write_ply", f"element vertex {len(V)}"]
    hdr += [f"property float {n}" for n in names]
    hdr += [f"element face {len(flist)}", f"property list uchar {index_type} vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if fmt == "ascii":
            for v in V:
                f.write((" ".join(repr(float(x)) for x in v) + "\n").encode())
            for fc in flist:
                f.write((" ".join(map(str, [len(fc)] + fc)) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            f.write(V.astype(e + "f4").tobytes())
            it = np.dtype(e + ("i4" if index_type == "int" else "u4"))
            lens = {len(fc) for fc in flist}
            if len(lens) == 1:  # homogeneous: vectorised
                k = lens.pop()
                rec = np.zeros(len(flist), np.dtype([("n", "u1"), ("i", it, (k,))]))
                rec["n"] = k; rec["i"] = np.asarray(flist)
                f.write(rec.tobytes())
            else:
                for fc in flist:
                    f.write(bytes([len(fc)]) + np.asarray(fc).astype(it).tobytes())


def read_pfm(path: str) -> np.ndarray:
    """-> (h, w, 3) float32, row 0 = top of the image."""
    L = host.lib()
    L.rtxh_pfm_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float))]
    L.rtxh_free.argtypes = [C.c_void_p]
    L.rtxh_free.restype = None
    w, h, p = C.c_int32(), C.c_int32(), C.POINTER(C.c_float)()
    rc = L.rtxh_pfm_read(str(path).encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc != 0:
        raise host.BackendError(f"rtxh_pfm_read failed ({rc}): {L.rtxh_last_error().decode(errors='replace')}")
    try:
        return np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    finally:
        L.rtxh_free(p)


def write_pfm(path: str, img, little_endian: bool = True, scale: float = 1.0) -> None:
    """img (h, w, 3) or (h, w): written bottom row first, as the format prescribes."""
    a = np.asarray(img, np.float32)
    grey = a.ndim == 2
    h, w = a.shape[:2]
    with open(path, "wb") as f:
        f.write(f"{'Pf' if grey else 'PF'}\n{w} {h}\n{-scale if little_endian else scale}\n".encode())
        f.write(a[::-1].astype("<f4" if little_endian else ">f4").tobytes())


def read_image(path: str) -> np.ndarray:
    """read_image (rc/imageio.rs:16-33) -> (h, w, 3) float32, row 0 = top: png / tga (c / 255), hdr, pfm by extension."""
    L = host.lib()
    w, h = C.c_int32(), C.c_int32()
    ptr = C.POINTER(C.c_float)()
    L.rtxh_free.argtypes = [C.c_void_p]
    L.rtxh_free.restype = None
    rc = L.rtxh_image_read(str(path).encode(), C.byref(w), C.byref(h), C.byref(ptr))
    if rc != 0:
        raise host.BackendError(f"rtxh_image_read failed ({rc}): {L.rtxh_last_error().decode(errors='replace')}")
    try:
        return np.ctypeslib.as_array(ptr, (h.value, w.value, 3)).copy()
    finally:
        L.rtxh_free(ptr)


# ---- writers of test images (PNG specification / TGA 2.0 / Radiance RGBE), independent of the C++ decoders ----------------
_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))   # x0 y0 dx dy


def _png_pack_rows(a: np.ndarray, depth: int) -> list:
    """a: (h, w, channels) unsigned samples -> list of packed scanlines (bytes)."""
    h, w, ch = a.shape
    rows = []
    for y in range(h):
        if depth == 16:
            rows.append(a[y].astype(">u2").tobytes())
        elif depth == 8:
            rows.append(a[y].astype(np.uint8).tobytes())
        else:
            per = 8 // depth
            vals = a[y, :, 0].astype(np.uint32)
            pad = (-w) % per
            vals = np.concatenate([vals, np.zeros(pad, np.uint32)]).reshape(-1, per)
            shifts = np.array([(per - 1 - i) * depth for i in range(per)], np.uint32)
            rows.append((vals << shifts).sum(1).astype(np.uint8).tobytes())
    return rows


def _png_filter(rows: list, bpp: int, filters) -> bytes:
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, cur in enumerate(rows):
        ft = filters[y % len(filters)]
        line = bytearray(len(cur))
        for i, x in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (x - pred) & 0xff
        out.append(ft)
        out += line
        prev = cur
    return bytes(out)


def write_png(path: str, samples, color_type: int, depth: int = 8, palette=None, filters=(0, 1, 2, 3, 4), interlace: bool = False,
              level: int = 6, idat_split: int = 0, extra_chunks=()) -> None:
    """samples: (h, w, channels) unsigned integers in the file's own sample range (palette indices for colour type 3)."""
    import struct
    import zlib
    a = np.asarray(samples)
    if a.ndim == 2:
        a = a[..., None]
    h, w, ch = a.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = max(1, ch * depth // 8)
    if interlace:
        raw = b""
        for x0, y0, dx, dy in _ADAM7:
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _png_filter(_png_pack_rows(sub, depth), bpp, filters)
    else:
        raw = _png_filter(_png_pack_rows(a, depth), bpp, filters)
    z = zlib.compress(raw, level)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    for t, d in extra_chunks:
        out += chunk(t, d)
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    parts = [z] if not idat_split else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    for part in parts:
        out += chunk(b"IDAT", part)
    out += chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def write_tga(path: str, img, kind: str = "rgb", rle: bool = False, top_origin: bool = False, palette=None, id_field: bytes = b"") -> None:
    """img: (h, w, 3|4) uint8 for "rgb" / "rgba", (h, w) uint8 for "grey" and "mapped" (palette (n, 3) uint8). Row 0 = top."""
    import struct
    a = np.asarray(img, np.uint8)
    h, w = a.shape[:2]
    base = {"rgb": 2, "rgba": 2, "grey": 3, "mapped": 1}[kind]
    if kind in ("rgb", "rgba"):
        px = a[..., [2, 1, 0] + ([3] if kind == "rgba" else [])]
    else:
        px = a[..., None]
    bytes_pp = px.shape[-1]
    rows = px if top_origin else px[::-1]
    cmap = b"" if palette is None else np.asarray(palette, np.uint8)[:, ::-1].tobytes()
    hdr = struct.pack("<BBBHHBHHHHBB", len(id_field), 1 if kind == "mapped" else 0, base + (8 if rle else 0), 0, 0 if palette is None else len(palette),
                      24 if kind == "mapped" else 0, 0, 0, w, h, 8 * bytes_pp, (0x20 if top_origin else 0) | (8 if kind == "rgba" else 0))
    flat = rows.reshape(-1, bytes_pp)
    if not rle:
        body = flat.tobytes()
    else:
        body = bytearray()
        i, n = 0, len(flat)
        while i < n:
            run = 1
            while i + run < n and run < 128 and (flat[i + run] == flat[i]).all():
                run += 1
            if run > 1:
                body.append(0x80 | (run - 1))
                body += flat[i].tobytes()
                i += run
            else:
                lit = 1
                while i + lit < n and lit < 128 and not (i + lit + 1 < n and (flat[i + lit + 1] == flat[i + lit]).all()):
                    lit += 1
                body.append(lit - 1)
                body += flat[i:i + lit].tobytes()
                i += lit
        body = bytes(body)
    with open(path, "wb") as f:
        f.write(hdr + id_field + cmap + body)


def write_hdr(path: str, rgbe, rle: bool = True) -> None:
    """rgbe: (h, w, 4) uint8 mantissas + shared exponent, written as a Radiance picture (-Y h +X w)."""
    a = np.asarray(rgbe, np.uint8)
    h, w = a.shape[:2]
    out = bytearray(b"#?RADIANCE\n# written by rustracer_amd.ingest.write_hdr\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
    for y in range(h):
        if not rle or w < 8 or w >= 32768:
            out += a[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 0xff])
        for c in range(4):
            comp = a[y, :, c]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and comp[i + run] == comp[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(comp[i])])
                    i += run
                else:
                    lit = 1
                    while i + lit < w and lit < 128 and not (i + lit + 2 < w and comp[i + lit] == comp[i + lit + 1] == comp[i + lit + 2]):
                        lit += 1
                    out.append(lit)
                    out += comp[i:i + lit].tobytes()
                    i += lit
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_exr(path: str, img, compression: str = "zip", pixel_type: str = "half", layer: str = "", alpha: bool = False, origin=(0, 0), extra_channels=()) -> None:
    """Scan-line OpenEXR file (OpenEXR file layout specification) of img (h, w, 3) float: channels <layer>B/G/R (+A, + extra names)
    in alphabetical order, half or float samples, compression none | rle | zips | zip. Test writer: numpy + zlib only."""
    import struct
    import zlib
    a = np.asarray(img, np.float32)
    h, w = a.shape[:2]
    comp = {"none": 0, "rle": 1, "zips": 2, "zip": 3}[compression]
    ptype, dt = (1, "<f2") if pixel_type == "half" else (2, "<f4")
    planes = {layer + "R": a[..., 0], layer + "G": a[..., 1], layer + "B": a[..., 2]}
    if alpha:
        planes[layer + "A"] = np.ones((h, w), np.float32)
    for k, name in enumerate(extra_channels):
        planes[name] = np.full((h, w), 0.25 * (k + 1), np.float32)
    names = sorted(planes)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBxxxii", ptype, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = origin
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = (struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp])) +
           attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
           attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) +
           attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines = 16 if comp == 3 else 1

    def pack(block: bytes) -> bytes:
        if comp == 0:
            return block
        b = np.frombuffer(block, np.uint8)
        t = np.concatenate([b[0::2], b[1::2]]).astype(np.int16)          # even bytes, then odd bytes
        d = t.copy()
        d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256                        # byte deltas, biased by 128
        pre = d.astype(np.uint8).tobytes()
        if comp in (2, 3):
            out = zlib.compress(pre, 6)
        else:
            out = bytearray()
            i, n = 0, len(pre)
            while i < n:
                run = 1
                while i + run < n and run < 128 and pre[i + run] == pre[i]:
                    run += 1
                if run >= 3:
                    out += bytes([run - 1, pre[i]])
                    i += run
                else:
                    lit = 1
                    while i + lit < n and lit < 127 and not (i + lit + 2 < n and pre[i + lit] == pre[i + lit + 1] == pre[i + lit + 2]):
                        lit += 1
                    out += bytes([(256 - lit) & 0xff]) + pre[i:i + lit]
                    i += lit
            out = bytes(out)
        return out if len(out) < len(block) else block                    # a block that does not shrink is stored raw
    blocks = []
    for yb in range(0, h, lines):
        rows = b"".join(b"".join(planes[n][y].astype(dt).tobytes() for n in names) for y in range(yb, min(h, yb + lines)))
        data = pack(rows)
        blocks.append(struct.pack("<iI", y0 + yb, len(data)) + data)
    table_at = len(hdr)
    pos = table_at + 8 * len(blocks)
    table = b""
    for blk in blocks:
        table += struct.pack("<Q", pos)
        pos += len(blk)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(blocks))
