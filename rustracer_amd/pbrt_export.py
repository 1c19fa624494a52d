"""Write a SceneDesc as a pbrt-v3 scene file (+ PFM images) that rustracer - and this repository's C++ host
(`rtxh_pbrt_load`, include/rtx_host.h) - reads back to the same scene.

A caller-side tool: the scenes of `rustracer_amd.scenes` become files the reference's
`rustracer-cli scene.pbrt` can render, and the parser tests use it for a full round trip
(SceneDesc -> .pbrt -> C++ parser -> same tables, same BVH, same render parameters).

Numbers are written with 9 significant digits, which identifies every float32 exactly.
"""
from __future__ import annotations

import os

import numpy as np

from . import scene_desc as sd
from .ingest import write_pfm

_FILTER_NAMES = {sd.FILTER_BOX: "box", sd.FILTER_TRIANGLE: "triangle", sd.FILTER_GAUSSIAN: "gaussian", sd.FILTER_MITCHELL: "mitchell"}
_WRAP_NAMES = {sd.WRAP_REPEAT: "repeat", sd.WRAP_BLACK: "black", sd.WRAP_CLAMP: "clamp"}
_MAT_NAMES = {sd.MAT_MATTE: "matte", sd.MAT_PLASTIC: "plastic", sd.MAT_METAL: "metal", sd.MAT_MIRROR: "mirror", sd.MAT_GLASS: "glass", sd.MAT_UBER: "uber",
              sd.MAT_SUBSTRATE: "substrate", sd.MAT_MIX: "mix", sd.MAT_TRANSLUCENT: "translucent", sd.MAT_DISNEY: "disney"}
# material kind -> {slot: (pbrt parameter, "spectrum" | "float")}   (the create() of each rc/material/*.rs)
_S, _F = "spectrum", "float"
_SLOTS = {
    sd.MAT_MATTE: {"kd": ("Kd", _S), "sigma": ("sigma", _F)},
    sd.MAT_PLASTIC: {"kd": ("Kd", _S), "ks": ("Ks", _S), "roughness": ("roughness", _F)},
    sd.MAT_METAL: {"eta": ("eta", _S), "k": ("k", _S), "roughness": ("roughness", _F), "urough": ("uroughness", _F), "vrough": ("vroughness", _F)},
    sd.MAT_MIRROR: {"kr": ("Kr", _S)},
    sd.MAT_GLASS: {"kr": ("Kr", _S), "kt": ("Kt", _S), "eta": ("index", _F), "urough": ("uroughness", _F), "vrough": ("vroughness", _F)},
    sd.MAT_UBER: {"kd": ("Kd", _S), "ks": ("Ks", _S), "kr": ("Kr", _S), "kt": ("Kt", _S), "roughness": ("roughness", _F), "urough": ("uroughness", _F),
                  "vrough": ("vroughness", _F), "eta": ("index", _F), "opacity": ("opacity", _S)},
    sd.MAT_SUBSTRATE: {"kd": ("Kd", _S), "ks": ("Ks", _S), "urough": ("uroughness", _F), "vrough": ("vroughness", _F)},
    sd.MAT_TRANSLUCENT: {"kd": ("Kd", _S), "ks": ("Ks", _S), "reflect": ("reflect", _S), "transmit": ("transmit", _S), "roughness": ("roughness", _F)},
    sd.MAT_DISNEY: {"kd": ("color", _S), "ks": ("metallic", _F), "eta": ("eta", _F), "roughness": ("roughness", _F), "kr": ("speculartint", _F),
                    "urough": ("anisotropic", _F), "kt": ("sheen", _F), "sigma": ("sheentint", _F), "vrough": ("clearcoat", _F), "k": ("clearcoatgloss", _F),
                    "opacity": ("spectrans", _F), "reflect": ("scatterdistance", _S), "transmit": ("flatness", _F), "amount": ("difftrans", _F)},
    sd.MAT_MIX: {"amount": ("amount", _S)},
}


def _n(x) -> str:
    x = float(np.float32(x))
    if x == float("inf"):
        raise ValueError("infinite values have no pbrt spelling")
    return "%.9g" % x


def _nums(a) -> str:
    return " ".join(_n(x) for x in np.asarray(a, np.float32).reshape(-1))


class _Writer:
    def __init__(self, desc: sd.SceneDesc, path: str, ply_over: int = 1 << 62):
        self.d = desc
        self.ply_over = ply_over
        self.dir = os.path.dirname(os.path.abspath(path))
        self.stem = os.path.splitext(os.path.basename(path))[0]
        self.out = []
        self.tex_names = {}   # (texture id, type) -> name
        self.mat_names = {}
        self.image_files = {}

    # ---- textures -----------------------------------------------------------------------------------------------
    def image_file(self, mip: int, flip: bool, grey: bool) -> str:
        key = (mip, flip, grey)
        if key not in self.image_files:
            name = f"{self.stem}_img{mip}{'f' if flip else ''}{'g' if grey else ''}.pfm"
            data = self.d.mipmaps[mip].data
            if flip:
                data = data[::-1]  # ImageTexture::new flips what it reads (imagemap.rs:52-60)
            write_pfm(os.path.join(self.dir, name), data[..., 0] if grey else data)
            self.image_files[key] = name
        return self.image_files[key]

    def param(self, pname: str, tex: int, typ: str) -> str:
        """`"type name" value` for texture id `tex` used as a parameter of type typ."""
        t = self.d.textures[tex]
        if t.kind == sd.TEX_CONST:
            return f'"float {pname}" [{_n(t.value[0])}]' if typ == _F else f'"rgb {pname}" [{_nums(t.value)}]'
        return f'"texture {pname}" "{self.texture(tex, typ)}"'

    def texture(self, tex: int, typ: str) -> str:
        key = (tex, typ)
        if key in self.tex_names:
            return self.tex_names[key]
        t = self.d.textures[tex]
        name = f"t{tex}{'f' if typ == _F else 's'}"
        mapping = f'"float uscale" [{_n(t.mapping[0])}] "float vscale" [{_n(t.mapping[1])}] "float udelta" [{_n(t.mapping[2])}] "float vdelta" [{_n(t.mapping[3])}]'
        if t.kind == sd.TEX_CONST:
            body = f'"constant" ' + (f'"float value" [{_n(t.value[0])}]' if typ == _F else f'"rgb value" [{_nums(t.value)}]')
        elif t.kind == sd.TEX_SCALE:
            body = f'"scale" {self.param("tex1", t.tex1, typ)} {self.param("tex2", t.tex2, typ)}'
        elif t.kind == sd.TEX_MIX:
            body = f'"mix" {self.param("tex1", t.tex1, typ)} {self.param("tex2", t.tex2, typ)} {self.param("amount", t.amount, _F)}'
        elif t.kind == sd.TEX_IMAGE:
            m = self.d.mipmaps[t.mip]
            body = (f'"imagemap" "string filename" "{self.image_file(t.mip, True, typ == _F)}" "bool gamma" "false" "bool trilinear" "{"true" if m.trilinear else "false"}" '
                    f'"float maxanisotropy" [{_n(m.max_aniso)}] "string wrap" "{_WRAP_NAMES[m.wrap]}" {mapping}')
        elif t.kind == sd.TEX_CHECKER:
            if typ == _F:
                raise ValueError("the reference has no float checkerboard (api.rs:1201-1216)")
            body = f'"checkerboard" {self.param("tex1", t.tex1, _S)} {self.param("tex2", t.tex2, _S)} "string aamode" "{"none" if t.amount == 0 else "closedform"}" {mapping}'
        elif t.kind == sd.TEX_UV:
            if typ == _F:
                raise ValueError("the reference has no float uv texture")
            body = f'"uv" {mapping}'
        elif t.kind == sd.TEX_FBM:
            body = f'"fbm" "float omega" [{_n(t.value[0])}] "integer octaves" [{int(t.amount)}]'
        else:
            raise ValueError(f"texture kind {t.kind}")
        self.out.append(f'Texture "{name}" "{"float" if typ == _F else "spectrum"}" {body}')
        self.tex_names[key] = name
        return name

    # ---- materials ----------------------------------------------------------------------------------------------
    def material(self, mid: int) -> str:
        if mid in self.mat_names:
            return self.mat_names[mid]
        m = self.d.materials[mid]
        parts = [f'"string type" "{_MAT_NAMES[m.kind]}"']
        for slot, (pname, typ) in _SLOTS[m.kind].items():
            if slot in m.params and m.params[slot] >= 0:
                parts.append(self.param(pname, m.params[slot], typ))
        if m.kind == sd.MAT_MIX:
            parts.append(f'"string namedmaterial1" "{self.material(m.params["m1"])}" "string namedmaterial2" "{self.material(m.params["m2"])}"')
        else:
            if m.kind == sd.MAT_DISNEY:
                parts.append(f'"bool thin" "{"true" if m.params.get("m1", 0) else "false"}"')
            else:
                parts.append(f'"bool remaproughness" "{"true" if m.remap_roughness else "false"}"')
            if m.bump >= 0:
                parts.append(self.param("bumpmap", m.bump, _F))
        name = f"m{mid}"
        self.out.append(f'MakeNamedMaterial "{name}" ' + " ".join(parts))
        self.mat_names[mid] = name
        return name

    # ---- lights ---------------------------------------------------------------------------------------------------
    @staticmethod
    def quadric_shape(sp) -> str:
        kind = getattr(sp, "kind", 0)
        if kind == 0:
            return f'Shape "sphere" "float radius" [{_n(sp.radius)}] "float zmin" [{_n(sp.z_min)}] "float zmax" [{_n(sp.z_max)}] "float phimax" [{_n(sp.phi_max)}]'
        if kind == 1:
            return f'Shape "disk" "float radius" [{_n(sp.radius)}] "float height" [{_n(sp.z_min)}] "float innerradius" [{_n(sp.z_max)}] "float phimax" [{_n(sp.phi_max)}]'
        # the reference's Cylinder::create reads z_min / z_max / phi_max (cylinder.rs:31-34)
        return f'Shape "cylinder" "float radius" [{_n(sp.radius)}] "float z_min" [{_n(sp.z_min)}] "float z_max" [{_n(sp.z_max)}] "float phi_max" [{_n(sp.phi_max)}]'

    def light(self, l: sd.Light):
        if l.kind == sd.LIGHT_POINT:
            self.out.append(f'LightSource "point" "rgb I" [{_nums(l.rgb)}] "point from" [{_nums(l.vec)}]')
        elif l.kind == sd.LIGHT_DISTANT:
            self.out.append(f'LightSource "distant" "rgb L" [{_nums(l.rgb)}] "point from" [{_nums(l.vec)}] "point to" [0 0 0]')
        elif l.kind == sd.LIGHT_INFINITE:
            l2w = np.eye(4, dtype=np.float32) if l.l2w is None else np.asarray(l.l2w, np.float32)
            self.out.append("AttributeBegin")
            self.out.append(f"  Transform [{_nums(l2w.T)}]")  # column-major in the file (api.rs:596-600)
            self.out.append(f'  LightSource "infinite" "string mapname" "{self.image_file(l.mip, False, False)}"')
            self.out.append("AttributeEnd")
        else:
            raise ValueError("area lights are written with their shape")

    def write(self):
        d, o = self.d, self.out
        c, f, s, it = d.camera, d.film, d.sampler, d.integrator
        o.append(f"LookAt {_nums(c.pos)}  {_nums(c.look)}  {_nums(c.up)}")
        cam = f'Camera "perspective" "float fov" [{_n(c.fov)}] "float lensradius" [{_n(c.lens_radius)}] "float focaldistance" [{_n(c.focal_distance)}]'
        if c.frame_aspect is not None:
            cam += f' "float frameaspectratio" [{_n(c.frame_aspect)}]'
        if c.screen_window is not None:
            cam += f' "float screenwindow" [{_nums(c.screen_window)}]'
        o.append(cam)
        film = f'Film "image" "integer xresolution" [{f.xres}] "integer yresolution" [{f.yres}] "float cropwindow" [{_nums(f.crop)}] "float scale" [{_n(f.scale)}] "string filename" "{d.name}.png"'
        if np.isfinite(f.max_sample_luminance):
            film += f' "float maxsampleluminance" [{_n(f.max_sample_luminance)}]'
        o.append(film)
        fp = f.filter_params
        extra = {sd.FILTER_BOX: "", sd.FILTER_TRIANGLE: "", sd.FILTER_GAUSSIAN: f' "float alpha" [{_n(fp[2])}]', sd.FILTER_MITCHELL: f' "float B" [{_n(fp[2])}] "float C" [{_n(fp[3])}]'}[f.filter_kind]
        o.append(f'PixelFilter "{_FILTER_NAMES[f.filter_kind]}" "float xwidth" [{_n(fp[0])}] "float ywidth" [{_n(fp[1])}]{extra}')
        o.append(f'Sampler "02sequence" "integer pixelsamples" [{s.spp}] "integer dimensions" [{s.dims}]')
        integ = f'Integrator "path" "integer maxdepth" [{it.max_depth}] "float rrthreshold" [{_n(it.rr_threshold)}] "string lightsamplestrategy" "{it.light_strategy}"'
        if it.pixel_bounds is not None:
            integ += f' "integer pixelbounds" [{" ".join(str(int(x)) for x in it.pixel_bounds)}]'
        o.append(integ)
        o.append(f'Accelerator "bvh" "integer maxnodeprims" [{d.max_prims_per_node}]')
        o.append("WorldBegin")
        other = [(i, l) for i, l in enumerate(d.lights) if l.kind != sd.LIGHT_DIFFUSE_AREA]
        nv0 = 0
        for m in range(len(d._idx)):
            lights = d._light[m]
            first = int(lights[0]) if lights.size and lights[0] >= 0 else None
            while other and first is not None and other[0][0] < first:
                self.light(other.pop(0)[1])
            mats = np.unique(d._mat[m])
            assert mats.size == 1
            flags = int(d._flags[m][0])
            name = self.material(int(mats[0]))
            masks = ""  # "alpha" / "shadowalpha" of the mesh (mesh.rs:134-156): a float texture, or the float 0
            if getattr(d, "_alpha", None):
                for pname, tid in zip(("alpha", "shadowalpha"), (int(x) for x in d._alpha[m][0])):
                    if tid >= 0:
                        masks += " " + self.param(pname, tid, _F)
            o.append("AttributeBegin")
            o.append(f'  NamedMaterial "{name}"')
            if flags & sd.TRI_FLIP:
                o.append("  ReverseOrientation")
            if first is not None:
                l0 = d.lights[first]
                assert all(tuple(d.lights[int(i)].rgb) == tuple(l0.rgb) for i in lights)
                o.append(f'  AreaLightSource "diffuse" "rgb L" [{_nums(l0.rgb)}] "bool twosided" "{"true" if l0.two_sided else "false"}"')
            nv = d._P[m].shape[0]
            if d._idx[m].shape[0] > self.ply_over and not (flags & sd.TRI_HAS_S):
                # a large mesh goes into a binary PLY next to the scene file (Shape "plymesh", rc/shapes/plymesh.rs); tangents have no PLY property
                from .ingest import write_ply
                name = f"{self.stem}_mesh{m}.ply"
                write_ply(os.path.join(self.dir, name), d._P[m], d._idx[m] - nv0, N=d._N[m] if flags & sd.TRI_HAS_N else None, UV=d._UV[m] if flags & sd.TRI_HAS_UV else None)
                shape = f'  Shape "plymesh" "string filename" "{name}"' + masks
            else:
                shape = f'  Shape "trianglemesh" "integer indices" [{" ".join(str(int(x)) for x in (d._idx[m] - nv0).reshape(-1))}] "point P" [{_nums(d._P[m])}]'
                if flags & sd.TRI_HAS_N:
                    shape += f' "normal N" [{_nums(d._N[m])}]'
                if flags & sd.TRI_HAS_UV:
                    shape += f' "float uv" [{_nums(d._UV[m])}]'
                if flags & sd.TRI_HAS_S:
                    shape += f' "vector S" [{_nums(d._S[m])}]'
                shape += masks
            o.append(shape)
            o.append("AttributeEnd")
            nv0 += nv
        for k, sp in enumerate(getattr(d, "spheres", [])):  # Shape "sphere" under its object-to-world transform, with its area light if it has one
            name = self.material(sp.material)
            o.append("AttributeBegin")
            o.append(f'  NamedMaterial "{name}"')
            o.append(f"  Transform [{_nums(np.asarray(sp.o2w, np.float32).T)}]")
            if sp.reverse_orientation:
                o.append("  ReverseOrientation")
            if sp.light >= 0:
                l = d.lights[sp.light]
                o.append(f'  AreaLightSource "diffuse" "rgb L" [{_nums(l.rgb)}] "bool twosided" "{"true" if l.two_sided else "false"}"')
            o.append("  " + self.quadric_shape(sp))
            o.append("AttributeEnd")
        # ObjectBegin .. ObjectEnd per object (one Shape per run of triangles that share material and flags: the meshes add_object was given), then
        # one ObjectInstance per placement under its primitive-to-world matrix
        for k, ob in enumerate(getattr(d, "objects", [])):
            # named materials and textures live in the graphics state, which ObjectEnd / AttributeEnd pop (api.rs:1019-1051): whatever the definition names is made
            # before it (a material first used inside a definition and again after it came back as "no such named material, using matte": scripts/fuzz_pbrt.py)
            for mid in sorted({int(x) for x in ob.mat} | {int(q.material) for q in (getattr(ob, "quadrics", None) or [])}):
                self.material(mid)
            if getattr(ob, "alpha", None) is not None:
                for tid in sorted({int(x) for x in np.asarray(ob.alpha).reshape(-1) if x >= 0}):
                    self.param("alpha", tid, _F)
            o.append(f'ObjectBegin "object{k}"')
            nt = ob.idx.shape[0]
            t0 = 0
            while t0 < nt:
                t1 = t0
                emit = getattr(ob, "emit", None)
                alpha = getattr(ob, "alpha", None)
                while (t1 + 1 < nt and ob.mat[t1 + 1] == ob.mat[t0] and ob.flags[t1 + 1] == ob.flags[t0] and ob.idx[t1 + 1].min() >= ob.idx[t0:t1 + 1].min()
                       and (emit is None or emit[t1 + 1] == emit[t0]) and (alpha is None or tuple(alpha[t1 + 1]) == tuple(alpha[t0]))):
                    t1 += 1
                tri = ob.idx[t0:t1 + 1]
                v0, v1 = int(tri.min()), int(tri.max()) + 1
                flags = int(ob.flags[t0])
                o.append("  AttributeBegin")
                o.append(f'    NamedMaterial "{self.material(int(ob.mat[t0]))}"')
                if flags & sd.TRI_FLIP:
                    o.append("    ReverseOrientation")
                if emit is not None and emit[t0] >= 0:  # an area light inside an object definition: shown, never sampled (rc/api.rs:954-964)
                    rgb, two_sided = d.emitters[int(emit[t0])]
                    o.append(f'    AreaLightSource "diffuse" "rgb L" [{_nums(rgb)}] "bool twosided" "{"true" if two_sided else "false"}"')
                shape = f'    Shape "trianglemesh" "integer indices" [{" ".join(str(int(x)) for x in (tri - v0).reshape(-1))}] "point P" [{_nums(ob.P[v0:v1])}]'
                if flags & sd.TRI_HAS_N:
                    shape += f' "normal N" [{_nums(ob.N[v0:v1])}]'
                if flags & sd.TRI_HAS_UV:
                    shape += f' "float uv" [{_nums(ob.UV[v0:v1])}]'
                if flags & sd.TRI_HAS_S:
                    shape += f' "vector S" [{_nums(ob.S[v0:v1])}]'
                if alpha is not None:  # "alpha" / "shadowalpha" of a mesh inside the definition
                    for pname, tid in zip(("alpha", "shadowalpha"), (int(x) for x in alpha[t0])):
                        if tid >= 0:
                            shape += " " + self.param(pname, tid, _F)
                o.append(shape)
                o.append("  AttributeEnd")
                t0 = t1 + 1
            for sp in (getattr(ob, "quadrics", None) or []):  # the definition's quadrics, each under the CTM it was given inside ObjectBegin .. ObjectEnd
                o.append("  AttributeBegin")
                o.append(f'    NamedMaterial "{self.material(sp.material)}"')
                o.append(f"    ConcatTransform [{_nums(np.asarray(sp.o2w, np.float32).T)}]")
                if sp.reverse_orientation:
                    o.append("    ReverseOrientation")
                if sp.light <= -2:
                    rgb, two_sided = d.emitters[-2 - int(sp.light)]
                    o.append(f'    AreaLightSource "diffuse" "rgb L" [{_nums(rgb)}] "bool twosided" "{"true" if two_sided else "false"}"')
                o.append("    " + self.quadric_shape(sp))
                o.append("  AttributeEnd")
            o.append("ObjectEnd")
        for i in getattr(d, "instances", []):
            o.append("AttributeBegin")
            o.append(f"  Transform [{_nums(np.asarray(i.o2w, np.float32).T)}]")
            o.append(f'  ObjectInstance "object{i.obj}"')
            o.append("AttributeEnd")
        for _, l in other:
            self.light(l)
        o.append("WorldEnd")
        return "\n".join(o) + "\n"


def write_pbrt(desc: sd.SceneDesc, path: str, ply_over: int = 1 << 62) -> str:
    """Writes `path` (and the PFM images it names, next to it); returns the scene text. Meshes of more than `ply_over` triangles
    are written as binary PLY files and referenced with Shape "plymesh"."""
    w = _Writer(desc, path, ply_over)
    text = w.write()
    with open(path, "w") as fh:
        fh.write(text)
    return text
