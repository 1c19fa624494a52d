"""The drop-in boundary cannot drift: every struct of include/rtx_hip.h and include/rtx_host.h is parsed from the header and compared, field by field
(name, type, array length, order), with
  (a) the `#[repr(C)]` blocks and the `extern "C"` prototypes INTEGRATION.md tells a rustracer maintainer to paste (RtSceneDesc had
      n_unlisted_lights in the wrong place and nothing noticed),
  (b) the ctypes / numpy mirrors in rustracer_amd/ (positional: a wrong order reads the wrong bytes silently),
  (c) the sizes the built libraries report (rt_sizeof / rtxh_sizeof), computed independently from the parsed fields with the C layout rules.
No GPU, no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rustracer_amd import host, ingest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MACROS = {"RT_MAX_MIP_LEVELS": 16, "RT_N_SLOTS": 16}
SCALARS = {"float": ("f32", 4), "double": ("f64", 8), "int32_t": ("i32", 4), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8), "int64_t": ("i64", 8),
           "uint16_t": ("u16", 2), "uint8_t": ("u8", 1), "char": ("c_char", 1), "int": ("i32", 4), "void": ("c_void", 0)}


def _strip_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def parse_c_structs(path):
    """{struct name: [(field, canonical type, array length or 0)]}; canonical type: 'f32', '*f32', '*rt_bvh_node', 'rtxh_render_params' ..."""
    src = _strip_comments(open(path).read())
    enum_n = re.search(r"RT_SLOT_M2,\s*RT_N_SLOTS", src)  # RT_N_SLOTS is an enumerator: count it instead of trusting MACROS
    if enum_n:
        body = re.search(r"enum\s*\{\s*(RT_SLOT_KD.*?RT_N_SLOTS)\s*\}", src, flags=re.S).group(1)
        assert len([x for x in body.split(",") if x.strip()]) - 1 == MACROS["RT_N_SLOTS"]
    m = re.search(r"#define\s+RT_MAX_MIP_LEVELS\s+(\d+)", src)
    if m:
        assert int(m.group(1)) == MACROS["RT_MAX_MIP_LEVELS"]
    out = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*(\w+);", src, flags=re.S):
        assert m.group(1) == m.group(3)
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            base, ptr, rest = mm.group(2), mm.group(3), mm.group(4)
            ctype = SCALARS[base][0] if base in SCALARS else base
            for d in rest.split(","):
                d = d.strip()
                p = ptr
                if d.startswith("*"):
                    p, d = "*", d[1:].strip()
                am = re.match(r"^(\w+)(?:\[(\w+)\])?$", d)
                assert am, (m.group(1), decl)
                n = am.group(2)
                n = 0 if n is None else (int(n) if n.isdigit() else MACROS[n])
                fields.append((am.group(1), p + ctype, n))
        out[m.group(1)] = fields
    return out


def parse_c_prototypes(path):
    """{function: (return type, [argument types])} in the canonical spelling above (argument names dropped)."""
    src = _strip_comments(open(path).read())
    src = re.sub(r"typedef struct \w+\s*\{.*?\}\s*\w+;", "", src, flags=re.S)
    src = re.sub(r"enum\s*\{.*?\}\s*;", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = {}
    for stmt in src.split(";"):
        m = re.search(r"((?:const\s+)?\w+\s*\**)\s*(rtx?h?_\w+)\s*\(([^)]*)\)\s*$", stmt, flags=re.S)
        if m and "typedef" not in stmt:
            out[m.group(2)] = (_c_type(m.group(1)), [] if m.group(3).strip() in ("", "void") else [_c_type(a) for a in m.group(3).split(",")])
    return out


def _c_type(decl):
    decl = " ".join(decl.replace("*", " * ").split())
    decl = re.sub(r"\[\w*\]$", " *", re.sub(r"\s+\w+(\[\w*\])$", r" \1", decl) if re.search(r"\w+\[\w*\]$", decl) else decl)  # `int32_t n_voxels[3]` is a pointer
    toks = [t for t in decl.split() if t != "const"]
    stars = toks.count("*")
    toks = [t for t in toks if t != "*"]
    base = toks[0]
    return "*" * stars + (SCALARS[base][0] if base in SCALARS else base)


def camel(name):  # rt_scene_desc -> RtSceneDesc
    return "".join(w.capitalize() for w in name.split("_"))


def parse_rust(md_path):
    text = open(md_path).read()
    blocks = re.findall(r"```rust\n(.*?)```", text, flags=re.S)
    src = _strip_comments("\n".join(blocks))
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for f in re.split(r",(?![^\[]*\])", m.group(2)):
            f = " ".join(f.split())
            if not f:
                continue
            name, ty = [x.strip() for x in f.split(":", 1)]
            am = re.match(r"^\[(.+);\s*(\d+)\]$", ty)
            n = int(am.group(2)) if am else 0
            ty = am.group(1).strip() if am else ty
            fields.append((name, _rust_type(ty), n))
        structs[m.group(1)] = fields
    fns = {}
    ext = re.search(r'extern "C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    for m in re.finditer(r"fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        fns[m.group(1)] = (_rust_type(m.group(3).strip()) if m.group(3) else "c_void", [_rust_type(a) for a in args])
    return structs, fns


def _rust_type(ty):
    stars = 0
    while True:
        mm = re.match(r"^\*(?:const|mut)\s+(.*)$", ty)
        if not mm:
            break
        stars, ty = stars + 1, mm.group(1).strip()
    return "*" * stars + ty


HIP = parse_c_structs(os.path.join(ROOT, "include", "rtx_hip.h"))
HOSTH = parse_c_structs(os.path.join(ROOT, "include", "rtx_host.h"))
ALL = {**HIP, **HOSTH}


def c_layout(fields):
    """(size, alignment, [offsets]) by the C rules of the x86-64 / amdgcn ABIs (natural alignment, no packing)."""
    off, align, offs = 0, 1, []
    for _, ty, n in fields:
        if ty.startswith("*"):
            sz, al = 8, 8
        elif ty in ALL:
            sz, al, _ = c_layout(ALL[ty])
        else:
            sz = al = {v[0]: v[1] for v in SCALARS.values()}[ty]
        off = (off + al - 1) // al * al
        offs.append(off)
        off += sz * max(n, 1)
        align = max(align, al)
    return (off + align - 1) // align * align, align, offs


def test_the_header_parser_sees_every_struct():
    assert set(HIP) == {"rt_bvh_node", "rt_tri_meta", "rt_sphere", "rt_instance", "rt_texture", "rt_image", "rt_material", "rt_light", "rt_scene_desc", "rt_camera",
                        "rt_film_desc", "rt_sampler_desc", "rt_path_desc", "rt_shard", "rt_stats"}
    assert set(HOSTH) == {"rtxh_render_params", "rtxh_emitter_info", "rtxh_instance_info", "rtxh_light_info", "rtxh_ply", "rtxh_pbrt_result"}
    assert HIP["rt_scene_desc"][-1] == ("n_unlisted_lights", "u32", 0) and HIP["rt_scene_desc"][1] == ("nodes", "*rt_bvh_node", 0)
    assert ("width", "i32", 16) in HIP["rt_image"] and ("slot", "i32", 16) in HIP["rt_material"]


def test_integration_md_rust_structs_equal_the_header_field_for_field():
    structs, _ = parse_rust(os.path.join(ROOT, "INTEGRATION.md"))
    want = {camel(k): [(n, t if t.lstrip("*") not in HIP else "*" * t.count("*") + camel(t.lstrip("*")), a) for n, t, a in v] for k, v in HIP.items()}
    assert set(structs) == set(want), sorted(set(want) ^ set(structs))   # every struct of the header is spelled out, none "as in rtx_hip.h"
    for name in want:
        assert structs[name] == want[name], (name, [x for x in zip(structs[name], want[name]) if x[0] != x[1]][:3])


def test_integration_md_prototypes_equal_the_header():
    _, fns = parse_rust(os.path.join(ROOT, "INTEGRATION.md"))
    protos = parse_c_prototypes(os.path.join(ROOT, "include", "rtx_hip.h"))
    assert {"rt_scene_create", "rt_scene_destroy", "rt_render", "rt_last_error", "rt_multi_create", "rt_multi_render", "rt_multi_destroy", "rt_sizeof"} <= set(fns)

    def canon(t):
        base = t.lstrip("*")
        base = {"rt_scene": "RtScene", "rt_multi": "RtMulti"}.get(base, camel(base) if base in HIP else base)
        return "*" * t.count("*") + base
    for name, (ret, args) in fns.items():
        assert name in protos, name
        cret, cargs = protos[name]
        assert [canon(a) for a in cargs] == args, (name, [canon(a) for a in cargs], args)
        assert canon(cret) == ret or (cret == "c_void" and ret == "c_void"), (name, cret, ret)


_CT = {"f32": C.c_float, "f64": C.c_double, "i32": C.c_int32, "u32": C.c_uint32, "u64": C.c_uint64, "u16": C.c_uint16, "u8": C.c_uint8, "c_char": C.c_char}


def _ctypes_fields(cls):
    out = []
    for name, t in cls._fields_:
        n = 0
        if hasattr(t, "_length_") and not issubclass(t, C.Structure):
            n, t = t._length_, t._type_
        if isinstance(t, type) and issubclass(t, C.Structure):
            ty = {host.RenderParams: "rtxh_render_params"}[t]
        elif t is C.c_void_p:
            ty = "*"
        elif hasattr(t, "_type_") and not isinstance(t._type_, str):
            ty = "*" + {v: k for k, v in _CT.items()}[t._type_]
        else:
            ty = {v: k for k, v in _CT.items()}[t]
        out.append((name, ty, n))
    return out


@pytest.mark.parametrize("cls, cname", [(host.Stats, "rt_stats"), (host.RenderParams, "rtxh_render_params"), (host.PbrtResult, "rtxh_pbrt_result"), (ingest._Ply, "rtxh_ply")])
def test_ctypes_mirrors_equal_the_headers(cls, cname):
    got, want = _ctypes_fields(cls), ALL[cname]
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        wt = "*" if (w[1].startswith("*") and g[1] == "*") else w[1]   # a c_void_p mirrors any pointer
        assert (g[0], g[1], g[2]) == (w[0], wt, w[2]), (g, w)
    assert C.sizeof(cls) == c_layout(want)[0]


def _dtype_fields(dt):
    out = []
    for name in dt.names:
        sub, off = dt.fields[name][:2]
        n = int(np.prod(sub.shape)) if sub.shape else 0
        out.append((name, {"<f4": "f32", "<i4": "i32", "<u4": "u32", "|u1": "u8"}[sub.base.str], n, off))
    return out


@pytest.mark.parametrize("table, cname, renames", [
    ("textures", "rt_texture", {}), ("materials", "rt_material", {}), ("lights", "rtxh_light_info", {}),
    ("instances", "rtxh_instance_info", {}), ("emitters", "rtxh_emitter_info", {})])
def test_numpy_table_mirrors_equal_the_headers(table, cname, renames):
    dt = host._TABLES[table][1]
    want = ALL[cname]
    got = _dtype_fields(dt)
    offs = c_layout(want)[2]
    assert len(got) == len(want)
    for g, w, o in zip(got, want, offs):
        assert (g[0], g[1], g[2], g[3]) == (w[0], w[1], w[2], o), (g, w, o)
    assert dt.itemsize == c_layout(want)[0]


def test_quadric_table_is_rt_sphere_plus_material_and_light():
    dt = host._TABLES["quadrics"][1]
    want = HIP["rt_sphere"] + [("material", "i32", 0), ("light", "i32", 0)]
    got = _dtype_fields(dt)
    assert [(g[0], g[1], g[2]) for g in got] == want and [g[3] for g in got] == c_layout(want)[2]


def test_scene_desc_constants_equal_the_header_enums():
    from rustracer_amd import scene_desc as sd
    src = _strip_comments(open(os.path.join(ROOT, "include", "rtx_hip.h")).read())
    enums = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", src, flags=re.S):
        v = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, val = [x.strip() for x in item.split("=")]
                v = int(val)
            else:
                k, v = item, v + 1
            enums[k] = v
    defines = {k: int(v.rstrip("u")) for k, v in re.findall(r"#define\s+(RT_(?:TRI|PRIM|FLAG)_\w+)\s+(\d+u?)", src)}
    checked = 0
    for k, v in {**enums, **defines}.items():
        py = k[3:]  # RT_MAT_MATTE -> MAT_MATTE
        if hasattr(sd, py):
            assert getattr(sd, py) == v, (k, v, getattr(sd, py))
            checked += 1
        if hasattr(host, k):
            assert getattr(host, k) == v, k
            checked += 1
    assert checked >= 30, checked   # materials, slots, textures, lights, wraps, triangle flags, render flags


def test_built_libraries_report_the_sizes_the_headers_imply():
    L = host.lib()
    for name, fields in ALL.items():
        assert L.rtxh_sizeof(name.encode()) == c_layout(fields)[0], name
    for name in HIP:
        assert host.hip_lib().rt_sizeof(name.encode()) == c_layout(HIP[name])[0], name
    assert L.rtxh_sizeof(b"no_such_struct") == -1


def test_every_declared_entry_point_is_exported():
    for hdr, lib in (("rtx_hip.h", host.hip_lib()), ("rtx_host.h", host.lib())):
        protos = parse_c_prototypes(os.path.join(ROOT, "include", hdr))
        assert len(protos) >= 20
        for name in protos:
            assert hasattr(lib, name), f"{hdr}: {name} is declared and not exported"
