"""Host-level scene description shared by every backend binding.

This is the *unflattened* input the reference's `RealApi` accumulates before `world_end`
(rc/api.rs:174-176 `RenderOptions.primitives/lights`, :913-966 `shape()`): triangle meshes already
in world space (rc/shapes/mesh.rs:61), one `DiffuseAreaLight` per emissive triangle in shape order
(rc/api.rs:933-946), material/texture tables, the light list, and the Camera/Film/Sampler/Integrator
parameters with the defaults of their `create()` functions.

Pure numpy + dataclasses: no backend code is imported here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# --- enums shared (by value) with include/rtx_hip.h and oracle/ -------------------------------
TEX_CONST, TEX_SCALE, TEX_MIX, TEX_IMAGE, TEX_CHECKER, TEX_UV, TEX_FBM = 0, 1, 2, 3, 4, 5, 6
(MAT_MATTE, MAT_PLASTIC, MAT_METAL, MAT_MIRROR, MAT_GLASS, MAT_UBER, MAT_SUBSTRATE, MAT_MIX,
 MAT_TRANSLUCENT, MAT_DISNEY) = range(10)
LIGHT_DIFFUSE_AREA, LIGHT_POINT, LIGHT_DISTANT, LIGHT_INFINITE = 0, 1, 2, 3
FILTER_BOX, FILTER_TRIANGLE, FILTER_GAUSSIAN, FILTER_MITCHELL = 0, 1, 2, 3
WRAP_REPEAT, WRAP_BLACK, WRAP_CLAMP = 0, 1, 2
SAMPLER_REF, SAMPLER_KEYED = 0, 1
# tri_flags bits
TRI_FLIP, TRI_HAS_N, TRI_HAS_UV, TRI_HAS_S = 1, 2, 4, 8
# order of the 16 material parameter slots
MAT_SLOTS = ("kd", "ks", "kr", "kt", "sigma", "roughness", "urough", "vrough", "eta", "k", "opacity",
             "reflect", "transmit", "amount", "m1", "m2")

# Metal's default eta / k: Spectrum::from_sampled over the measured copper tables and the CIE 1931 observer (rc/material/metal.rs:25-29,
# 84-200; rc/spectrum.rs:108-126; rc/cie.rs). These are the f32 values that arithmetic gives - the C++ host computes them (rtxh_copper in
# rtx_host.cpp, used by the .pbrt loader) and tests/test_pbrt_cpu.py checks these constants against it and against a numpy restatement.
COPPER_ETA = (0.19999069, 0.9220846, 1.0998759)
COPPER_K = (3.9046354, 2.4476333, 2.1376526)


@dataclass
class Texture:
    kind: int = TEX_CONST
    value: Sequence[float] = (0.0, 0.0, 0.0)
    tex1: int = -1
    tex2: int = -1
    amount: int = -1
    mip: int = -1
    mapping: Sequence[float] = (1.0, 1.0, 0.0, 0.0)  # su sv du dv (rc/texture/mod.rs:38-61)


@dataclass
class MipImage:
    data: np.ndarray  # (h, w, 3) float32, already y-flipped / gamma-decoded / scaled (imagemap.rs:52-84)
    trilinear: bool = False
    max_aniso: float = 8.0
    wrap: int = WRAP_REPEAT


@dataclass
class Material:
    kind: int = MAT_MATTE
    params: dict = field(default_factory=dict)  # slot name -> texture id (or material id for m1/m2)
    remap_roughness: bool = True
    bump: int = -1  # "bumpmap" float texture id (rc/material/mod.rs:50-92)

    def slots(self) -> np.ndarray:
        return np.array([int(self.params.get(k, -1)) for k in MAT_SLOTS], dtype=np.int32)


@dataclass
class Light:
    kind: int = LIGHT_DIFFUSE_AREA
    tri: int = -1
    sphere: int = -1  # area light on analytic sphere number `sphere` instead of on a triangle
    rgb: Sequence[float] = (1.0, 1.0, 1.0)
    two_sided: bool = False
    vec: Sequence[float] = (0.0, 0.0, 0.0)  # point: position; distant: from - to
    mip: int = -1
    l2w: Optional[np.ndarray] = None  # 4x4 (infinite)
    w2l: Optional[np.ndarray] = None


@dataclass
class SphereShape:  # Shape "sphere" (rc/shapes/sphere.rs:53-68)
    o2w: np.ndarray  # 4x4 object-to-world, float32
    w2o: np.ndarray
    radius: float = 1.0
    z_min: float = -1.0
    z_max: float = 1.0
    phi_max: float = 360.0
    reverse_orientation: bool = False
    material: int = 0
    light: int = -1  # index into the light list of its DiffuseAreaLight, or -1
    kind: int = 0    # 0 sphere, 1 disk (z_min = height, z_max = inner radius; rc/shapes/disk.rs), 2 cylinder (rc/shapes/cylinder.rs)


@dataclass
class Camera:
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    look: Sequence[float] = (0.0, 0.0, 1.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fov: float = 90.0            # camera.rs:108
    lens_radius: float = 0.0     # :84
    focal_distance: float = 1e6  # :85
    frame_aspect: Optional[float] = None                 # "frameaspectratio" (:86-97); None = xres / yres
    screen_window: Optional[Sequence[float]] = None      # "screenwindow" xmin xmax ymin ymax (:98-107); overrides the aspect ratio


@dataclass
class Film:
    xres: int = 1280  # film.rs:124-125
    yres: int = 720
    crop: Sequence[float] = (0.0, 1.0, 0.0, 1.0)
    filter_kind: int = FILTER_BOX
    filter_params: Sequence[float] = (0.5, 0.5, 0.0, 0.0)  # xwidth ywidth alpha|B C
    scale: float = 1.0
    max_sample_luminance: float = float("inf")


@dataclass
class Sampler:
    spp: int = 16   # zerotwosequence.rs:59
    dims: int = 4   # :60


@dataclass
class Integrator:
    max_depth: int = 5           # path.rs:50
    rr_threshold: float = 1.0    # :51
    light_strategy: str = "spatial"  # :52
    pixel_bounds: Optional[Sequence[int]] = None  # x0 x1 y0 y1


@dataclass
class ObjectDef:
    """An object definition: a triangle soup in object space (vertices as the shapes' own CTMs leave them) and per-triangle material / flags."""
    P: np.ndarray
    idx: np.ndarray
    N: Optional[np.ndarray]
    UV: Optional[np.ndarray]
    S: Optional[np.ndarray]
    mat: np.ndarray
    flags: np.ndarray
    emit: Optional[np.ndarray] = None  # per triangle: index into SceneDesc.emitters or -1 (None: no triangle of the object emits)
    # what else the definition holds (rc/api.rs:1019-1051 collects every primitive): quadrics in OBJECT space (SphereShape; light = -1 or -2 - k: emitter k of
    # SceneDesc.emitters, in no light list) - primitive ids inside the object after its triangles - and {alpha, shadowalpha} texture ids per triangle
    quadrics: Optional[list] = None
    alpha: Optional[np.ndarray] = None


@dataclass
class InstanceDef:
    obj: int
    o2w: np.ndarray
    w2o: np.ndarray


class SceneDesc:
    def __init__(self):
        self._P: List[np.ndarray] = []
        self._N: List[np.ndarray] = []
        self._UV: List[np.ndarray] = []
        self._S: List[np.ndarray] = []
        self._idx: List[np.ndarray] = []
        self._mat: List[np.ndarray] = []
        self._light: List[np.ndarray] = []
        self._flags: List[np.ndarray] = []
        self._alpha: List[np.ndarray] = []  # per triangle (alpha, shadowalpha) float-texture ids or -1 (mesh.rs:134-156)
        self._nv = 0
        self._nt = 0
        self.textures: List[Texture] = []
        self.mipmaps: List[MipImage] = []
        self.spheres: List[SphereShape] = []
        self.objects: List[ObjectDef] = []
        self.emitters = []  # (rgb, two_sided) of emitters that are in no light list (emitting meshes inside object definitions)
        self.instances: List[InstanceDef] = []
        self.materials: List[Material] = []
        self.lights: List[Light] = []
        self.camera = Camera()
        self.film = Film()
        self.sampler = Sampler()
        self.integrator = Integrator()
        self.max_prims_per_node = 4  # bvh/mod.rs:76
        self.name = "scene"

    # ---- textures -----------------------------------------------------------------------
    def const_tex(self, v) -> int:
        if np.isscalar(v):
            v = (float(v),) * 3
        self.textures.append(Texture(TEX_CONST, tuple(float(x) for x in v)))
        return len(self.textures) - 1

    def scale_tex(self, t1: int, t2: int) -> int:
        self.textures.append(Texture(TEX_SCALE, tex1=t1, tex2=t2))
        return len(self.textures) - 1

    def mix_tex(self, t1: int, t2: int, amount: int) -> int:
        self.textures.append(Texture(TEX_MIX, tex1=t1, tex2=t2, amount=amount))
        return len(self.textures) - 1

    def add_mip(self, data, trilinear=False, max_aniso=8.0, wrap=WRAP_REPEAT) -> int:
        data = np.ascontiguousarray(data, dtype=np.float32)
        assert data.ndim == 3 and data.shape[2] == 3
        h, w = data.shape[:2]
        self.mipmaps.append(MipImage(data, trilinear, max_aniso, wrap))
        return len(self.mipmaps) - 1

    def image_tex(self, mip: int, su=1.0, sv=1.0, du=0.0, dv=0.0) -> int:
        self.textures.append(Texture(TEX_IMAGE, mip=mip, mapping=(su, sv, du, dv)))
        return len(self.textures) - 1

    def checker_tex(self, t1, t2, su=1.0, sv=1.0, du=0.0, dv=0.0, aa="closedform") -> int:  # checkerboard.rs:44-95 (dimension 2, "uv" mapping)
        self.textures.append(Texture(TEX_CHECKER, tex1=self._t(_f(t1)), tex2=self._t(_f(t2)), amount=0 if aa == "none" else 1, mapping=(su, sv, du, dv)))
        return len(self.textures) - 1

    def uv_tex(self, su=1.0, sv=1.0, du=0.0, dv=0.0) -> int:  # uv.rs:21-40
        self.textures.append(Texture(TEX_UV, mapping=(su, sv, du, dv)))
        return len(self.textures) - 1

    def fbm_tex(self, omega=0.5, octaves=8) -> int:  # fbm.rs:25-44 (identity texture-to-world)
        self.textures.append(Texture(TEX_FBM, value=(float(omega), 0.0, 0.0), amount=int(octaves)))
        return len(self.textures) - 1

    def _t(self, v) -> int:
        """int => existing texture id; number/tuple => new constant texture."""
        if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
            return int(v)
        return self.const_tex(v)

    # ---- materials (defaults = the reference's create() functions) -----------------------
    def matte(self, kd=0.5, sigma=0.0) -> int:  # matte.rs:22-34
        self.materials.append(Material(MAT_MATTE, {"kd": self._t(_f(kd)), "sigma": self._t(float(sigma))}))
        return len(self.materials) - 1

    def plastic(self, kd=0.25, ks=0.25, roughness=0.1, remap=True) -> int:  # plastic.rs:25-41
        self.materials.append(Material(MAT_PLASTIC, {"kd": self._t(_f(kd)), "ks": self._t(_f(ks)),
                                                      "roughness": self._t(float(roughness))}, remap))
        return len(self.materials) - 1

    def metal(self, eta=COPPER_ETA, k=COPPER_K, roughness=0.01, urough=None, vrough=None, remap=True) -> int:  # metal.rs:23-47
        p = {"eta": self._t(_f(eta)), "k": self._t(_f(k)), "roughness": self._t(float(roughness))}
        if urough is not None:
            p["urough"] = self._t(float(urough))
        if vrough is not None:
            p["vrough"] = self._t(float(vrough))
        self.materials.append(Material(MAT_METAL, p, remap))
        return len(self.materials) - 1

    def mirror(self, kr=0.9) -> int:  # mirror.rs:22
        self.materials.append(Material(MAT_MIRROR, {"kr": self._t(_f(kr))}))
        return len(self.materials) - 1

    def glass(self, kr=1.0, kt=1.0, index=1.5, urough=0.0, vrough=0.0, remap=True) -> int:  # glass.rs:30-38
        self.materials.append(Material(MAT_GLASS, {"kr": self._t(_f(kr)), "kt": self._t(_f(kt)), "eta": self._t(float(index)),
                                                    "urough": self._t(float(urough)), "vrough": self._t(float(vrough))}, remap))
        return len(self.materials) - 1

    def uber(self, kd=0.25, ks=0.25, kr=0.0, kt=0.0, roughness=0.1, urough=None, vrough=None, index=1.5,
             opacity=1.0, remap=True) -> int:  # uber.rs:32-44
        p = {"kd": self._t(_f(kd)), "ks": self._t(_f(ks)), "kr": self._t(_f(kr)), "kt": self._t(_f(kt)),
             "roughness": self._t(float(roughness)), "eta": self._t(float(index)), "opacity": self._t(_f(opacity))}
        if urough is not None:
            p["urough"] = self._t(float(urough))
        if vrough is not None:
            p["vrough"] = self._t(float(vrough))
        self.materials.append(Material(MAT_UBER, p, remap))
        return len(self.materials) - 1

    def substrate(self, kd=0.5, ks=0.5, urough=0.1, vrough=0.1, remap=True) -> int:  # substrate.rs:24-29
        self.materials.append(Material(MAT_SUBSTRATE, {"kd": self._t(_f(kd)), "ks": self._t(_f(ks)),
                                                        "urough": self._t(float(urough)), "vrough": self._t(float(vrough))}, remap))
        return len(self.materials) - 1

    def translucent(self, kd=0.25, ks=0.25, reflect=0.5, transmit=0.5, roughness=0.1, remap=True) -> int:  # translucent.rs:28-34
        self.materials.append(Material(MAT_TRANSLUCENT, {"kd": self._t(_f(kd)), "ks": self._t(_f(ks)), "reflect": self._t(_f(reflect)),
                                                          "transmit": self._t(_f(transmit)), "roughness": self._t(float(roughness))}, remap))
        return len(self.materials) - 1

    def disney(self, color=0.5, metallic=0.0, eta=1.5, roughness=0.5, speculartint=0.0, anisotropic=0.0, sheen=0.0, sheentint=0.5,
               clearcoat=0.0, clearcoatgloss=1.0, spectrans=0.0, scatterdistance=0.0, thin=False, flatness=0.0, difftrans=1.0) -> int:  # disney.rs:43-80
        t = lambda v: self._t(_f(v))
        self.materials.append(Material(MAT_DISNEY, {"kd": t(color), "ks": t(metallic),
                                                    "eta": t(eta), "roughness": t(roughness), "kr": t(speculartint), "urough": t(anisotropic), "kt": t(sheen),
                                                    "sigma": t(sheentint), "vrough": t(clearcoat), "k": t(clearcoatgloss), "opacity": t(spectrans),
                                                    "reflect": t(scatterdistance), "transmit": t(flatness), "amount": t(difftrans), "m1": 1 if thin else 0}))
        return len(self.materials) - 1

    def set_bump(self, material: int, tex) -> int:
        """Attach a "bumpmap" float texture to a (non-mix) material; returns the material id."""
        assert self.materials[material].kind != MAT_MIX
        self.materials[material].bump = self._t(_f(tex))
        return material

    def mix(self, m1: int, m2: int, amount=0.5) -> int:  # mixmat.rs:22-30
        self.materials.append(Material(MAT_MIX, {"m1": int(m1), "m2": int(m2), "amount": self._t(_f(amount))}))
        return len(self.materials) - 1

    # ---- geometry ------------------------------------------------------------------------
    def add_mesh(self, P, idx, material: int, N=None, UV=None, S=None, reverse_orientation=False,
                 emission=None, two_sided=False, alpha=None, shadow_alpha=None) -> int:
        """Append a triangle mesh whose points are already in world space (identity CTM).
        Returns the index of its first triangle. One DiffuseAreaLight per triangle when `emission`
        is given (rc/api.rs:933-946), appended to the light list in triangle order."""
        P = np.ascontiguousarray(P, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, 3)
        nv, nt = P.shape[0], idx.shape[0]
        assert idx.min() >= 0 and idx.max() < nv
        flags = (TRI_FLIP if reverse_orientation else 0)
        self._P.append(P)
        for lst, arr, w, bit in ((self._N, N, 3, TRI_HAS_N), (self._UV, UV, 2, TRI_HAS_UV), (self._S, S, 3, TRI_HAS_S)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, w)
                assert a.shape[0] == nv
                flags |= bit
            else:
                a = np.zeros((nv, w), dtype=np.float32)
            lst.append(a)
        self._idx.append(idx + self._nv)
        self._mat.append(np.full(nt, material, dtype=np.int32))
        self._flags.append(np.full(nt, flags, dtype=np.uint8))
        # "alpha" / "shadowalpha": a float texture id, or a number (only 0 makes a mask: the constant-0 texture, mesh.rs:142-144, 154-156)
        am = [(-1 if a is None else (int(a) if isinstance(a, (int, np.integer)) and not isinstance(a, bool) else (self.const_tex(0.0) if float(a) == 0.0 else -1)))
              for a in (alpha, shadow_alpha)]
        self._alpha.append(np.tile(np.int32(am), (nt, 1)))
        first = self._nt
        if emission is not None:
            ids = np.arange(len(self.lights), len(self.lights) + nt, dtype=np.int32)
            for t in range(nt):
                self.lights.append(Light(LIGHT_DIFFUSE_AREA, tri=first + t, rgb=tuple(float(x) for x in emission), two_sided=two_sided))
            self._light.append(ids)
        else:
            self._light.append(np.full(nt, -1, dtype=np.int32))
        self._nv += nv
        self._nt += nt
        return first

    def add_sphere(self, center=(0.0, 0.0, 0.0), radius=1.0, material: int = 0, o2w=None, z_min=None, z_max=None, phi_max=360.0,
                   reverse_orientation=False, emission=None, two_sided=False) -> int:
        """Shape "sphere" (rc/shapes/sphere.rs): an analytic sphere - optionally clipped in z and phi - under `o2w` (default: a translation to
        `center`). With `emission` it carries one DiffuseAreaLight (one light per Shape, rc/api.rs:933-946). Returns the sphere's index."""
        if o2w is None:
            o2w = np.eye(4, dtype=np.float32)
            o2w[:3, 3] = np.float32(center)
        o2w = np.ascontiguousarray(o2w, np.float32)
        w2o = np.ascontiguousarray(np.linalg.inv(o2w.astype(np.float64)), np.float32)
        light = -1
        if emission is not None:
            self.lights.append(Light(LIGHT_DIFFUSE_AREA, tri=-1, sphere=len(self.spheres), rgb=tuple(float(x) for x in emission), two_sided=two_sided))
            light = len(self.lights) - 1
        self.spheres.append(SphereShape(o2w, w2o, float(radius), float(-radius if z_min is None else z_min), float(radius if z_max is None else z_max),
                                        float(phi_max), bool(reverse_orientation), int(material), light))
        return len(self.spheres) - 1

    def _add_quadric(self, kind, o2w, radius, a, b, phi_max, material, reverse_orientation, emission, two_sided) -> int:
        o2w = np.ascontiguousarray(o2w, np.float32)
        w2o = np.ascontiguousarray(np.linalg.inv(o2w.astype(np.float64)), np.float32)
        light = -1
        if emission is not None:
            self.lights.append(Light(LIGHT_DIFFUSE_AREA, tri=-1, sphere=len(self.spheres), rgb=tuple(float(x) for x in emission), two_sided=two_sided))
            light = len(self.lights) - 1
        self.spheres.append(SphereShape(o2w, w2o, float(radius), float(a), float(b), float(phi_max), bool(reverse_orientation), int(material), light, kind))
        return len(self.spheres) - 1

    def add_disk(self, o2w, radius=1.0, material: int = 0, height=0.0, inner_radius=0.0, phi_max=360.0, reverse_orientation=False, emission=None, two_sided=False) -> int:
        """Shape "disk" (rc/shapes/disk.rs): the annulus inner_radius <= r <= radius of the plane z = height of object space."""
        return self._add_quadric(1, o2w, radius, height, inner_radius, phi_max, material, reverse_orientation, emission, two_sided)

    def add_cylinder(self, o2w, radius=1.0, material: int = 0, z_min=-1.0, z_max=1.0, phi_max=360.0, reverse_orientation=False, emission=None, two_sided=False) -> int:
        """Shape "cylinder" (rc/shapes/cylinder.rs): the open cylinder of `radius` around the object-space z axis between z_min and z_max."""
        return self._add_quadric(2, o2w, radius, z_min, z_max, phi_max, material, reverse_orientation, emission, two_sided)

    def add_object(self, meshes, quadrics=None) -> int:
        """ObjectBegin ... ObjectEnd (rc/api.rs:1019-1051): triangle meshes in OBJECT space - an iterable of dicts with the keys of `add_mesh`
        (P, idx, material, N, UV, S, reverse_orientation, emission, two_sided, alpha, shadow_alpha) - and `quadrics`: dicts with kind (0 sphere, 1 disk, 2 cylinder),
        o2w (the CTM inside the definition), radius, z_min, z_max (disk: height, inner radius), phi_max, material, reverse_orientation, emission, two_sided. A mesh with `emission` is a shape under an AreaLightSource
        inside the object definition: it keeps its DiffuseAreaLight (it glows when a camera ray or a specular bounce reaches it) but the light never enters
        the scene's list (rc/api.rs:954-964) - an entry of `self.emitters`, not of `self.lights`. Returns the object's index; nothing is rendered until
        `add_instance` places it."""
        Ps, Is, Ns, UVs, Ss, mats, flags, emits, alphas, nv0 = [], [], [], [], [], [], [], [], [], 0
        for m in meshes:
            P = np.ascontiguousarray(m["P"], dtype=np.float32).reshape(-1, 3)
            idx = np.ascontiguousarray(m["idx"], dtype=np.int32).reshape(-1, 3)
            nv, nt = P.shape[0], idx.shape[0]
            assert idx.min() >= 0 and idx.max() < nv
            f = TRI_FLIP if m.get("reverse_orientation", False) else 0
            Ps.append(P)
            for lst, key, w, bit in ((Ns, "N", 3, TRI_HAS_N), (UVs, "UV", 2, TRI_HAS_UV), (Ss, "S", 3, TRI_HAS_S)):
                a = m.get(key)
                if a is not None:
                    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, w)
                    assert a.shape[0] == nv
                    f |= bit
                else:
                    a = np.zeros((nv, w), dtype=np.float32)
                lst.append(a)
            Is.append(idx + nv0)
            mats.append(np.full(nt, int(m["material"]), dtype=np.int32))
            flags.append(np.full(nt, f, dtype=np.uint8))
            em = -1
            if m.get("emission") is not None:
                self.emitters.append((tuple(float(x) for x in m["emission"]), bool(m.get("two_sided", False))))
                em = len(self.emitters) - 1
            emits.append(np.full(nt, em, dtype=np.int32))
            am = [(-1 if a is None else (int(a) if isinstance(a, (int, np.integer)) and not isinstance(a, bool) else (self.const_tex(0.0) if float(a) == 0.0 else -1)))
                  for a in (m.get("alpha"), m.get("shadow_alpha"))]
            alphas.append(np.tile(np.int32(am), (nt, 1)))
            nv0 += nv
        qs = []
        for q in quadrics or []:
            o2w = np.ascontiguousarray(q["o2w"], np.float32).reshape(4, 4)
            w2o = np.ascontiguousarray(np.linalg.inv(o2w.astype(np.float64)), np.float32)
            light = -1
            if q.get("emission") is not None:
                self.emitters.append((tuple(float(x) for x in q["emission"]), bool(q.get("two_sided", False))))
                light = -2 - (len(self.emitters) - 1)
            r = float(q.get("radius", 1.0)); kind = int(q.get("kind", 0))
            qs.append(SphereShape(o2w, w2o, r, float(q.get("z_min", -r if kind == 0 else 0.0)), float(q.get("z_max", r if kind == 0 else 0.0)), float(q.get("phi_max", 360.0)),
                                  bool(q.get("reverse_orientation", False)), int(q["material"]), light, kind))
        if not Ps:  # an object of quadrics only
            assert qs, "an object without primitives"
            self.objects.append(ObjectDef(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), None, None, None, np.zeros(0, np.int32), np.zeros(0, np.uint8), None, qs, None))
            return len(self.objects) - 1
        fl = np.ascontiguousarray(np.concatenate(flags), dtype=np.uint8)
        self.objects.append(ObjectDef(
            np.ascontiguousarray(np.concatenate(Ps)), np.ascontiguousarray(np.concatenate(Is)),
            np.ascontiguousarray(np.concatenate(Ns)) if (fl & TRI_HAS_N).any() else None,
            np.ascontiguousarray(np.concatenate(UVs)) if (fl & TRI_HAS_UV).any() else None,
            np.ascontiguousarray(np.concatenate(Ss)) if (fl & TRI_HAS_S).any() else None,
            np.ascontiguousarray(np.concatenate(mats)), fl,
            np.ascontiguousarray(np.concatenate(emits)) if any((e >= 0).any() for e in emits) else None,
            qs or None, np.ascontiguousarray(np.concatenate(alphas), dtype=np.int32) if any((a >= 0).any() for a in alphas) else None))
        return len(self.objects) - 1

    def add_instance(self, obj: int, o2w) -> int:
        """ObjectInstance (rc/api.rs:1053-1090): a TransformedPrimitive over object `obj` with primitive_to_world = `o2w` (rc/primitive.rs:79-118).
        The object's BVH is traversed in object space; nothing is copied. Returns the instance's index."""
        assert 0 <= obj < len(self.objects)
        o2w = np.ascontiguousarray(o2w, np.float32).reshape(4, 4)
        w2o = np.ascontiguousarray(np.linalg.inv(o2w.astype(np.float64)), np.float32)
        self.instances.append(InstanceDef(int(obj), o2w, w2o))
        return len(self.instances) - 1

    def add_quad(self, p0, p1, p2, p3, material: int, **kw) -> int:
        return self.add_mesh([p0, p1, p2, p3], [[0, 1, 2], [0, 2, 3]], material, **kw)

    def add_ply(self, path, material: int, **kw) -> int:
        """Shape "plymesh" (rc/shapes/plymesh.rs): vertices, normals, uvs and (split) faces from a PLY file."""
        from .ingest import read_ply
        m = read_ply(path)
        return self.add_mesh(m["P"], m["idx"], material, N=m["N"], UV=m["UV"], **kw)

    def add_pfm_mip(self, path, **kw) -> int:
        """An image-map / environment MIP pyramid from a PFM file (rc/imageio.rs:179-246)."""
        from .ingest import read_pfm
        return self.add_mip(read_pfm(path), **kw)

    # ---- lights --------------------------------------------------------------------------
    def point_light(self, pos, I=(1.0, 1.0, 1.0)) -> int:
        self.lights.append(Light(LIGHT_POINT, rgb=tuple(I), vec=tuple(pos)))
        return len(self.lights) - 1

    def distant_light(self, frm, to, L=(1.0, 1.0, 1.0)) -> int:
        d = np.float32(frm) - np.float32(to)
        self.lights.append(Light(LIGHT_DISTANT, rgb=tuple(L), vec=tuple(float(x) for x in d)))
        return len(self.lights) - 1

    def infinite_light(self, mip: int, l2w=None) -> int:
        l2w = np.eye(4, dtype=np.float32) if l2w is None else np.ascontiguousarray(l2w, dtype=np.float32)
        w2l = np.linalg.inv(l2w.astype(np.float64)).astype(np.float32)
        self.lights.append(Light(LIGHT_INFINITE, mip=mip, l2w=l2w, w2l=w2l))
        return len(self.lights) - 1

    # ---- flattened views -------------------------------------------------------------------
    @property
    def n_tris(self) -> int:
        return self._nt

    @property
    def n_verts(self) -> int:
        return self._nv

    def arrays(self):
        """(P, idx, N|None, UV|None, S|None, tri_material, tri_light, tri_flags) as contiguous arrays."""
        if not self._P:  # a scene of spheres / object instances only: no top-level triangle
            z = lambda w, dt: np.zeros((0, w) if w else (0,), dt)
            return z(3, np.float32), z(3, np.int32), None, None, None, z(0, np.int32), z(0, np.int32), z(0, np.uint8)
        P = np.ascontiguousarray(np.concatenate(self._P), dtype=np.float32)
        idx = np.ascontiguousarray(np.concatenate(self._idx), dtype=np.int32)
        flags = np.ascontiguousarray(np.concatenate(self._flags), dtype=np.uint8)
        N = np.ascontiguousarray(np.concatenate(self._N)) if (flags & TRI_HAS_N).any() else None
        UV = np.ascontiguousarray(np.concatenate(self._UV)) if (flags & TRI_HAS_UV).any() else None
        S = np.ascontiguousarray(np.concatenate(self._S)) if (flags & TRI_HAS_S).any() else None
        mat = np.ascontiguousarray(np.concatenate(self._mat), dtype=np.int32)
        light = np.ascontiguousarray(np.concatenate(self._light), dtype=np.int32)
        return P, idx, N, UV, S, mat, light, flags

    def alpha_ids(self):
        """(n_tris, 2) int32 {alpha, shadowalpha} texture ids, or None when no mesh carries a mask."""
        a = np.ascontiguousarray(np.concatenate(self._alpha), dtype=np.int32) if self._alpha else None
        return a if a is not None and (a >= 0).any() else None


def _f(v):
    """Normalise a python colour argument: texture ids (ints) pass through, numbers -> float."""
    if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
        return int(v)
    if np.isscalar(v):
        return float(v)
    return tuple(float(x) for x in v)
