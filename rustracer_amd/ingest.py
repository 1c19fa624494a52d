"""Input formats on the caller's side of the path: PLY meshes and PFM images.

Readers are the C++ host's (`rtxh_ply_read`, `rtxh_pfm_read` in include/rtx_host.h), restating
plymesh::create (rc/shapes/plymesh.rs:18-186) and read_image_pfm (rc/imageio.rs:179-246). The writers
here exist to emit the synthetic scenes in those formats; they are plain numpy and share no code with the readers.
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
        raise host.BackendError(f"rtxh_ply_read failed ({rc}): {L.rtxh_last_error().decode()}")
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
        raise host.BackendError(f"rtxh_pfm_read failed ({rc}): {L.rtxh_last_error().decode()}")
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
