"""ctypes binding of the product libraries (csrc/_build/librtx_host.so -> librtx_hip.so).

`HostScene` plays the role of rustracer's `RealApi` + `world_end` (rc/api.rs:913-1010): it feeds the
unflattened scene description to the C++ host layer (SAH BVH build, flattening, camera/film set-up)
and renders through the HIP backend. Nothing here computes radiance and nothing falls back to a CPU
path: without a GPU `render`/`trace` raise `BackendError`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_BUILD = os.path.join(_CSRC, "_build")
HOST_LIB = os.path.join(_BUILD, "librtx_host.so")
HIP_LIB = os.path.join(_BUILD, "librtx_hip.so")

RT_FLAG_COUNT_TRAVERSAL = 1
RT_FLAG_FILM_ON_DEVICE = 2
RT_FLAG_TIME_KERNELS = 4
RT_FLAG_COUNT_AS_RENDERED = 8
RT_FLAG_REF_STREAM = 16


class BackendError(RuntimeError):
    pass


STAMP = os.path.join(_BUILD, "source.sha")


def source_sha() -> str:
    """sha256 over everything the two libraries are built from (csrc/*.{h,hip,cpp,inl}, the Makefile, include/*.h)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".h", ".hip", ".cpp", ".inl")) or f == "Makefile"]
    files += [os.path.join(_HERE, "..", "include", f) for f in ("rtx_hip.h", "rtx_host.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force: bool = False) -> None:
    """Compile both libraries for gfx950 (hipcc cross-compiles without a GPU). Whether the built libraries are current is decided by CONTENT: the sha256 of the
    sources is stamped beside them (csrc/_build/source.sha) - the libraries are untracked and travel to the GPU box by snapshot, where file times mean nothing."""
    sha = source_sha()
    stamped = open(STAMP).read().strip() if os.path.exists(STAMP) else None
    if force or stamped != sha or not (os.path.exists(HOST_LIB) and os.path.exists(HIP_LIB)):
        subprocess.check_call(["make", "-C", _CSRC, "-s"] + (["-B"] if force or (stamped is not None and stamped != sha) else []))
        with open(STAMP, "w") as f:
            f.write(sha + "\n")


class RenderParams(C.Structure):
    _fields_ = [
        ("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_float * 4),
        ("filter_kind", C.c_int32), ("filter_params", C.c_float * 4),
        ("film_scale", C.c_float), ("max_sample_luminance", C.c_float),
        ("cam_to_world", C.c_float * 16), ("cam_to_world_inv", C.c_float * 16),
        ("fov", C.c_float), ("lens_radius", C.c_float), ("focal_distance", C.c_float),
        ("spp", C.c_int32), ("sampler_dims", C.c_int32),
        ("max_depth", C.c_int32), ("rr_threshold", C.c_float), ("light_strategy", C.c_int32),
        ("pixel_bounds", C.c_int32 * 4), ("rank", C.c_int32), ("world_size", C.c_int32), ("flags", C.c_uint32),
        ("screen_window", C.c_float * 4), ("has_pixel_bounds", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in (
        "camera_rays", "rays_closest", "rays_shadow", "rays_mis", "nodes_closest", "nodes_shadow", "nodes_mis",
        "tris_closest", "tris_shadow", "tris_mis", "paths_scrubbed")] +
        [(n, C.c_double) for n in ("ms_total", "ms_sampler", "ms_raygen", "ms_trace_closest", "ms_trace_any", "ms_trace_mis",
                                   "ms_shade", "ms_resolve", "ms_film", "ms_lightdist")] +
        [(n, C.c_uint64) for n in ("launches_trace_closest", "n_passes", "vertices_lambert_const", "vertices_lambert", "vertices_two_lobe", "vertices_generic")] +
        [(n, C.c_double) for n in ("ms_shade_lambert_const", "ms_shade_lambert", "ms_shade_two_lobe", "ms_shade_generic", "ms_shade_bin", "ms_shade_miss")] +
        [("shade_section_cycles", C.c_uint64 * 32)] +
        [(n, C.c_uint64) for n in ("rays_mis_any", "nodes_mis_any", "tris_mis_any")] + [("ms_trace_mis_any", C.c_double), ("ms_gather", C.c_double), ("rays_mis_not_cast", C.c_uint64), ("rays_tail_not_cast", C.c_uint64)] +
        [(n, C.c_uint64) for n in ("launches_trace_path", "launches_trace_shadow", "launches_trace_mis", "launches_trace_mis_any", "launches_shade")])

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "shade_section_cycles" else getattr(self, n)) for n, _ in self._fields_}


_lib = None
_hip = None


def lib():
    global _lib, _hip
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise BackendError(f"{HOST_LIB} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hip = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        _lib = C.CDLL(HOST_LIB)
        L = _lib
        L.rtxh_scene_new.restype = C.c_void_p
        L.rtxh_scene_free.argtypes = [C.c_void_p]
        L.rtxh_last_error.restype = C.c_char_p
        _hip.rt_last_error.restype = C.c_char_p
        _hip.rt_version.restype = C.c_char_p
        # the mirrors below are positional: a stale library (or a stale mirror) must not be used at all
        for name, mirror in (("rt_stats", Stats), ("rtxh_render_params", RenderParams), ("rtxh_pbrt_result", PbrtResult)):
            n = L.rtxh_sizeof(name.encode())
            if n != C.sizeof(mirror):
                _lib = _hip = None
                raise BackendError(f"{HOST_LIB}: sizeof({name}) = {n}, the ctypes mirror {mirror.__name__} has {C.sizeof(mirror)} bytes - rebuild (make -C rustracer_amd/csrc)")
    return _lib


def hip_lib():
    lib()
    return _hip


def device_available() -> bool:
    return bool(hip_lib().rt_device_available())


def _p(a, t=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _check(rc, what):
    if rc < 0:
        msg = lib().rtxh_last_error()
        raise BackendError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else '?'}")
    return rc


def look_at(pos, look, up):
    m = np.zeros((4, 4), np.float32)
    mi = np.zeros((4, 4), np.float32)
    lib().rtxh_look_at(_p(np.float32(pos)), _p(np.float32(look)), _p(np.float32(up)), _p(m), _p(mi))
    return m, mi


def screen_window_of(camera):
    """PerspectiveCamera::create (camera.rs:86-107): explicit "screenwindow", else from "frameaspectratio", else (0, 0, 0, 0) = the default."""
    if getattr(camera, "screen_window", None) is not None:
        return tuple(camera.screen_window)
    fr = getattr(camera, "frame_aspect", None)
    if fr is None:
        return (0.0, 0.0, 0.0, 0.0)
    fr = np.float32(fr)
    return (-fr, fr, -1.0, 1.0) if fr > 1.0 else (-1.0, 1.0, np.float32(-1.0) / fr, np.float32(1.0) / fr)


def render_params(desc, rank=0, world_size=1, flags=0) -> RenderParams:
    p = RenderParams()
    f, c, s, it = desc.film, desc.camera, desc.sampler, desc.integrator
    p.xres, p.yres = f.xres, f.yres
    p.crop[:] = [float(x) for x in f.crop]
    p.filter_kind = f.filter_kind
    p.filter_params[:] = [float(x) for x in f.filter_params]
    p.film_scale = f.scale
    p.max_sample_luminance = f.max_sample_luminance
    w2c, c2w = look_at(c.pos, c.look, c.up)  # CTM = identity * LookAt (api.rs:637); camera_to_world = CTM.inverse() (:726)
    w2c, c2w = w2c + np.float32(0.0), c2w + np.float32(0.0)  # the product with the identity CTM turns a -0 entry into +0
    p.cam_to_world[:] = c2w.reshape(-1).tolist()
    p.cam_to_world_inv[:] = w2c.reshape(-1).tolist()
    p.fov, p.lens_radius, p.focal_distance = c.fov, c.lens_radius, c.focal_distance
    p.screen_window[:] = [float(x) for x in screen_window_of(c)]
    p.spp, p.sampler_dims = s.spp, s.dims
    p.max_depth, p.rr_threshold = it.max_depth, it.rr_threshold
    p.light_strategy = 1 if it.light_strategy == "uniform" else 0
    pb = it.pixel_bounds
    p.pixel_bounds[:] = list(pb) if pb is not None else [0, 0, 0, 0]
    p.has_pixel_bounds = 1 if pb is not None else 0
    p.rank, p.world_size, p.flags = rank, world_size, flags
    return p


class HostScene:
    def __init__(self, desc, device_bvh=False, device_ingest=False):
        """device_bvh=True: the tree comes from the GPU's linear builder (rt_bvh_build) instead of the host SAH build.
        device_ingest=True: MIP pyramids and environment-map sampling tables are built on the GPU (bit-identical tables)."""
        L = lib()
        L.rtxh_set_device_ingest(1 if device_ingest else 0)
        try:
            self._build(desc, device_bvh)
        finally:
            L.rtxh_set_device_ingest(0)

    def _build(self, desc, device_bvh):
        L = lib()
        self.desc = desc
        self.h = C.c_void_p(L.rtxh_scene_new())
        P, idx, N, UV, S, mat, light, flags = desc.arrays()
        self._keep = (P, idx, N, UV, S, mat, light, flags)
        if idx.shape[0]:  # (a scene may hold spheres / object instances only)
            _check(L.rtxh_scene_set_mesh(self.h, _p(P), P.shape[0], _p(idx, C.c_int32), idx.shape[0], _p(N), _p(UV), _p(S),
                                         _p(mat, C.c_int32), _p(light, C.c_int32), _p(flags, C.c_uint8)), "set_mesh")
        alpha = desc.alpha_ids() if hasattr(desc, "alpha_ids") else None
        if alpha is not None:
            self._keep += (alpha,)
            _check(L.rtxh_scene_set_alpha(self.h, _p(alpha, C.c_int32)), "set_alpha")
        for m in desc.mipmaps:
            h, w = m.data.shape[:2]
            _check(L.rtxh_scene_add_mipmap(self.h, w, h, _p(m.data), int(m.trilinear), C.c_float(m.max_aniso), m.wrap), "add_mipmap")
        for t in desc.textures:
            _check(L.rtxh_scene_add_texture(self.h, t.kind, _p(np.float32(t.value)), t.tex1, t.tex2, t.amount, t.mip, _p(np.float32(t.mapping))), "add_texture")
        for m in desc.materials:
            _check(L.rtxh_scene_add_material(self.h, m.kind, _p(m.slots(), C.c_int32), int(m.remap_roughness), int(m.bump)), "add_material")
        for sp in getattr(desc, "spheres", []):
            _check(L.rtxh_scene_add_quadric(self.h, int(getattr(sp, "kind", 0)), _p(sp.o2w), _p(sp.w2o), C.c_float(sp.radius), C.c_float(sp.z_min), C.c_float(sp.z_max),
                                            C.c_float(sp.phi_max), int(sp.reverse_orientation), sp.material, sp.light), "add_quadric")
        for rgb, two_sided in getattr(desc, "emitters", []):
            _check(L.rtxh_scene_add_emitter(self.h, _p(np.float32(rgb)), int(two_sided)), "add_emitter")
        for k, o in enumerate(getattr(desc, "objects", [])):
            self._keep += (o,)
            _check(L.rtxh_scene_add_object(self.h, _p(o.P), o.P.shape[0], _p(o.idx, C.c_int32), o.idx.shape[0], _p(o.N), _p(o.UV), _p(o.S),
                                           _p(o.mat, C.c_int32), _p(o.flags, C.c_uint8)), "add_object")
            if getattr(o, "emit", None) is not None:
                _check(L.rtxh_scene_object_emitters(self.h, k, _p(o.emit, C.c_int32)), "object_emitters")
            for q in getattr(o, "quadrics", None) or []:  # quadrics of the object definition, in object space; light -2 - k: unlisted emitter k
                _check(L.rtxh_object_add_quadric(self.h, k, int(q.kind), _p(q.o2w), _p(q.w2o), C.c_float(q.radius), C.c_float(q.z_min), C.c_float(q.z_max), C.c_float(q.phi_max),
                                                 int(q.reverse_orientation), q.material, (-2 - q.light) if q.light <= -2 else -1), "object_add_quadric")
            if getattr(o, "alpha", None) is not None:
                _check(L.rtxh_object_set_alpha(self.h, k, _p(o.alpha, C.c_int32)), "object_set_alpha")
        for i in getattr(desc, "instances", []):
            _check(L.rtxh_scene_add_instance(self.h, i.obj, _p(i.o2w), _p(i.w2o)), "add_instance")
        for l in desc.lights:
            l2w = None if l.l2w is None else np.ascontiguousarray(l.l2w, np.float32)
            w2l = None if l.w2l is None else np.ascontiguousarray(l.w2l, np.float32)
            tri = l.tri if getattr(l, "sphere", -1) < 0 else -2 - l.sphere  # -2 - k: the area light sits on sphere k
            _check(L.rtxh_scene_add_light(self.h, l.kind, tri, _p(np.float32(l.rgb)), int(l.two_sided), _p(np.float32(l.vec)), l.mip, _p(l2w), _p(w2l)), "add_light")
        self.bvh_build_ms = None
        if device_bvh:
            ms = C.c_float()
            _check(L.rtxh_scene_commit_device_bvh(self.h, desc.max_prims_per_node, C.byref(ms)), "commit_device_bvh")
            self.bvh_build_ms = ms.value
        else:
            _check(L.rtxh_scene_commit(self.h, desc.max_prims_per_node), "commit")

    def __del__(self):
        try:
            lib().rtxh_scene_free(self.h)
        except Exception:
            pass

    # ---- CPU-side products -----------------------------------------------------------------
    def bvh(self):
        L = lib()
        nn, npr = C.c_int32(), C.c_int32()
        L.rtxh_scene_bvh_sizes(self.h, C.byref(nn), C.byref(npr))
        bounds = np.zeros((nn.value, 6), np.float32)
        offset = np.zeros(nn.value, np.uint32)
        nprims = np.zeros(nn.value, np.uint16)
        axis = np.zeros(nn.value, np.uint8)
        ordered = np.zeros(npr.value, np.int32)
        L.rtxh_scene_bvh_get(self.h, _p(bounds), _p(offset, C.c_uint32), _p(nprims, C.c_uint16), _p(axis, C.c_uint8), _p(ordered, C.c_int32))
        return dict(bounds=bounds, offset=offset, n_prims=nprims, axis=axis, ordered=ordered)

    def bvh_sizes(self):
        nn, npr = C.c_int32(), C.c_int32()
        lib().rtxh_scene_bvh_sizes(self.h, C.byref(nn), C.byref(npr))
        return nn.value, npr.value

    def link_tables(self, mid=False):
        """The link tables rt_scene_create gives the stackless LDS walks of this scene (rt_link_tables: host only, no device), as
        dict(kept=[9 rows x n_nodes], full=..., start_kept=[9], start_full=[9], rays=[9], tests_all=[9], tests_kept=[9])."""
        nn = self.bvh_sizes()[0]
        words = 9 * nn + 9
        kept, full, stats = np.zeros(words, np.uint32), np.zeros(words, np.uint32), np.zeros(27, np.float64)
        _check(lib().rtxh_scene_link_tables(self.h, C.c_int32(1 if mid else 0), _p(kept, C.c_uint32), _p(full, C.c_uint32), C.c_uint64(words), _p(stats, C.c_double)), "rtxh_scene_link_tables")

        def split(t):
            rows = np.concatenate([t[:8 * nn].reshape(8, nn), t[8 * nn + 8:9 * nn + 8].reshape(1, nn)])
            return rows, np.concatenate([t[8 * nn:8 * nn + 8], t[9 * nn + 8:9 * nn + 9]])
        (rk, sk), (rf, sf) = split(kept), split(full)
        return dict(kept=rk, full=rf, start_kept=sk, start_full=sf, rays=stats[:9], tests_all=stats[9:18], tests_kept=stats[18:27])

    def lds_resident(self):
        """Does the traversal kernel keep this scene's nodes and primitives in LDS? Asked of the uploaded scene (rt_scene_query: what rt_scene_create decided,
        not re-derived here)."""
        return bool(_check(lib().rtxh_scene_query(self.h, 0), "scene_query"))

    def _render_params(self, **kw):
        return render_params(self.desc, **kw)

    def n_lights(self):
        return len(self.desc.lights)

    def table(self, name):
        return scene_table(self.h, name)

    def setup(self, **kw):
        p = self._render_params(**kw)
        r2c = np.zeros((4, 4), np.float32)
        dxdy = np.zeros(6, np.float32)
        table = np.zeros(256, np.float32)
        sb = np.zeros(4, np.int32)
        cr = np.zeros(4, np.int32)
        _check(lib().rtxh_camera_film_setup(C.byref(p), _p(r2c), _p(dxdy), _p(table), _p(sb, C.c_int32), _p(cr, C.c_int32)), "camera_film_setup")
        return dict(raster_to_camera=r2c, dx_camera=dxdy[:3].copy(), dy_camera=dxdy[3:].copy(), filter_table=table, sample_bounds=sb, cropped=cr, params=p)

    def mip_levels(self, mip):
        L = lib()
        w, h = C.c_int32(), C.c_int32()
        n = _check(L.rtxh_mip_level(self.h, mip, 0, C.byref(w), C.byref(h), None), "mip_level")
        out = []
        for lvl in range(n):
            L.rtxh_mip_level(self.h, mip, lvl, C.byref(w), C.byref(h), None)
            a = np.zeros((h.value, w.value, 3), np.float32)
            L.rtxh_mip_level(self.h, mip, lvl, C.byref(w), C.byref(h), _p(a))
            out.append(a)
        return out

    # ---- GPU ---------------------------------------------------------------------------------
    def upload(self, device=-1):
        _check(lib().rtxh_scene_upload(self.h, device), "upload")

    def render(self, rank=0, world_size=1, count_traversal=False, time_kernels=False, device_out=None, stream=0, count_as_rendered=False, ref_stream=False):
        """renderer::render on the GPU. Returns (film_xyzw (H,W,4) over the cropped bounds, stats dict).
        `device_out`: optional torch CUDA tensor (H,W,4) float32 that receives the film in HBM. `ref_stream`: the reference's own sampler stream - one RNG stream per
        16 x 16 tile, one lane per tile (RT_FLAG_REF_STREAM; slow by construction) - instead of the pixel-keyed one."""
        flags = ((RT_FLAG_COUNT_TRAVERSAL if count_traversal else 0) | (RT_FLAG_TIME_KERNELS if time_kernels else 0) |
                 (RT_FLAG_COUNT_AS_RENDERED if count_as_rendered else 0) | (RT_FLAG_REF_STREAM if ref_stream else 0))
        st = self.setup(rank=rank, world_size=world_size)
        cr = st["cropped"]
        w, h = int(cr[2] - cr[0]), int(cr[3] - cr[1])
        p = st["params"]
        stats = Stats()
        if device_out is not None:
            assert tuple(device_out.shape) == (h, w, 4) and device_out.is_contiguous()
            p.flags = flags | RT_FLAG_FILM_ON_DEVICE
            _check(lib().rtxh_render(self.h, C.byref(p), C.c_void_p(stream), C.c_void_p(device_out.data_ptr()), C.byref(stats)), "render")
            return device_out, stats.as_dict()
        p.flags = flags
        film = np.zeros((h, w, 4), np.float32)
        _check(lib().rtxh_render(self.h, C.byref(p), C.c_void_p(stream), _p(film), C.byref(stats)), "render")
        return film, stats.as_dict()

    def render_multi(self, devices, chunks_per_device=1, count_traversal=False, time_kernels=False, count_as_rendered=False, device_out=None):
        """The frame on several GPUs of this process (rt_multi_render): one host thread per entry of `devices`, chunks of tile rows pulled from a
        shared queue, rows gathered on devices[0]. Returns (film, total stats, [per-device stats]).
        `device_out`: optional torch CUDA tensor (H,W,4) float32 on devices[0] that receives the merged film in HBM."""
        flags = ((RT_FLAG_COUNT_TRAVERSAL if count_traversal else 0) | (RT_FLAG_TIME_KERNELS if time_kernels else 0) |
                 (RT_FLAG_COUNT_AS_RENDERED if count_as_rendered else 0))
        st = self.setup()
        cr = st["cropped"]
        w, h = int(cr[2] - cr[0]), int(cr[3] - cr[1])
        p = st["params"]
        dev = np.ascontiguousarray(devices, np.int32)
        total, per = Stats(), (Stats * len(dev))()
        if device_out is not None:
            assert tuple(device_out.shape) == (h, w, 4) and device_out.is_contiguous()
            p.flags = flags | RT_FLAG_FILM_ON_DEVICE
            _check(lib().rtxh_render_multi(self.h, C.byref(p), _p(dev, C.c_int32), len(dev), int(chunks_per_device), C.c_void_p(device_out.data_ptr()), C.byref(total), per), "render_multi")
            return device_out, total.as_dict(), [s.as_dict() for s in per]
        p.flags = flags
        film = np.zeros((h, w, 4), np.float32)
        _check(lib().rtxh_render_multi(self.h, C.byref(p), _p(dev, C.c_int32), len(dev), int(chunks_per_device), _p(film), C.byref(total), per), "render_multi")
        return film, total.as_dict(), [s.as_dict() for s in per]

    def trace(self, rays, any_hit=False, count=True):
        """count=True: the visit-counting kernels (one node per step, the reference's sequence); count=False: the
        kernels rt_render launches (same hit records, no counters)."""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        cnt = np.zeros(2, np.uint64)
        pc = _p(cnt, C.c_uint64) if count else None
        if any_hit:
            occ = np.zeros(n, np.uint32)
            _check(lib().rtxh_trace(self.h, _p(rays), C.c_uint64(n), 1, occ.ctypes.data_as(C.POINTER(C.c_float)), pc), "trace_any")
            return dict(occluded=occ > 0, nodes=int(cnt[0]), tris=int(cnt[1]))
        out = np.zeros((n, 4), np.float32)
        _check(lib().rtxh_trace(self.h, _p(rays), C.c_uint64(n), 0, _p(out), pc), "trace_closest")
        return dict(t=out[:, 0].copy(), prim=out[:, 1].copy().view(np.int32), b0=out[:, 2].copy(), b1=out[:, 3].copy(), nodes=int(cnt[0]), tris=int(cnt[1]))

    def trace_device(self, d_rays_ptr, n, d_hits_ptr, reps=10, stream=0):
        ms = C.c_float()
        _check(lib().rtxh_trace_device(self.h, C.c_void_p(d_rays_ptr), C.c_uint64(n), C.c_void_p(d_hits_ptr), reps, C.c_void_p(stream), C.byref(ms)), "trace_device")
        return ms.value

    def light_distribution(self):
        nv = np.zeros(3, np.int32)
        _check(lib().rtxh_light_distribution(self.h, _p(nv, C.c_int32), None, None, None), "light_distribution")
        total = int(nv[0]) * int(nv[1]) * int(nv[2])
        if total == 0:
            return dict(n_voxels=nv)
        nl = self.n_lights()
        func = np.zeros((total, nl), np.float32)
        cdf = np.zeros((total, nl + 1), np.float32)
        fint = np.zeros(total, np.float32)
        _check(lib().rtxh_light_distribution(self.h, _p(nv, C.c_int32), _p(func), _p(cdf), _p(fint)), "light_distribution")
        return dict(n_voxels=nv, func=func, cdf=cdf, func_int=fint)


def copper():
    """(eta rgb, k rgb): the defaults of Material "metal" (Metal::create, rc/material/metal.rs:25-29)."""
    e, k = (C.c_float * 3)(), (C.c_float * 3)()
    lib().rtxh_copper(e, k)
    return np.float32(list(e)), np.float32(list(k))


def spectrum_from_sampled(wavelengths_nm, values):
    """Spectrum::from_sampled (rc/spectrum.rs:108-126) -> RGB."""
    lam, v = np.ascontiguousarray(wavelengths_nm, np.float32), np.ascontiguousarray(values, np.float32)
    out = (C.c_float * 3)()
    _check(lib().rtxh_spectrum_from_sampled(_p(lam), _p(v), len(lam), out), "spectrum_from_sampled")
    return np.float32(list(out))


def spectrum_blackbody(temperature, scale=1.0):
    """A "blackbody" parameter's value (rc/paramset.rs:291-310) -> RGB."""
    out = (C.c_float * 3)()
    _check(lib().rtxh_spectrum_blackbody(C.c_float(temperature), C.c_float(scale), out), "spectrum_blackbody")
    return np.float32(list(out))


class PbrtResult(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("params", RenderParams), ("max_prims_per_node", C.c_int32), ("n_warnings", C.c_int32),
                ("film_filename", C.c_char * 512)]


_TEXTURE_DT = np.dtype([("kind", "<i4"), ("value", "<f4", 3), ("tex1", "<i4"), ("tex2", "<i4"), ("amount", "<i4"), ("image", "<i4"), ("mapping", "<f4", 4)])
_MATERIAL_DT = np.dtype([("kind", "<i4"), ("slot", "<i4", 16), ("remap_roughness", "<i4"), ("bump", "<i4")])
_LIGHT_DT = np.dtype([("kind", "<i4"), ("tri", "<i4"), ("rgb", "<f4", 3), ("two_sided", "<i4"), ("vec", "<f4", 3), ("mip", "<i4"), ("l2w", "<f4", 12), ("w2l", "<f4", 12)])
_TABLES = {"textures": (0, _TEXTURE_DT), "materials": (1, _MATERIAL_DT), "lights": (2, _LIGHT_DT), "P": (3, np.dtype(("<f4", 3))), "N": (4, np.dtype(("<f4", 3))),
           "UV": (5, np.dtype(("<f4", 2))), "S": (6, np.dtype(("<f4", 3))), "indices": (7, np.dtype(("<i4", 3))), "tri_material": (8, np.dtype("<i4")),
           "tri_light": (9, np.dtype("<i4")), "tri_flags": (10, np.dtype("u1")), "env_func": (11, np.dtype("<f4")), "env_cdf": (12, np.dtype("<f4")),
           "env_row_int": (13, np.dtype("<f4")), "env_marg_cdf": (14, np.dtype("<f4")),
           "instances": (15, np.dtype([("object", "<i4"), ("o2w", "<f4", (4, 4)), ("w2o", "<f4", (4, 4))])),
           "emitters": (16, np.dtype([("rgb", "<f4", (3,)), ("two_sided", "<i4")])),
           "quadrics": (17, np.dtype([("o2w", "<f4", (4, 4)), ("w2o", "<f4", (4, 4)), ("radius", "<f4"), ("z_min", "<f4"), ("z_max", "<f4"), ("theta_min", "<f4"), ("theta_max", "<f4"),
                                      ("phi_max", "<f4"), ("reverse_orientation", "<i4"), ("swaps_handedness", "<i4"), ("kind", "<i4"), ("height", "<f4"), ("inner_radius", "<f4"),
                                      ("material", "<i4"), ("light", "<i4")]))}
_OBJECT_TABLES = {"P": (0, np.dtype(("<f4", 3))), "N": (1, np.dtype(("<f4", 3))), "UV": (2, np.dtype(("<f4", 2))), "S": (3, np.dtype(("<f4", 3))),
                  "indices": (4, np.dtype(("<i4", 3))), "tri_material": (5, np.dtype("<i4")), "tri_flags": (6, np.dtype("u1")), "tri_emitter": (7, np.dtype("<i4"))}


def scene_table(handle, name):
    """Copy of one unflattened table of an rtxh_scene (rtxh_scene_inspect)."""
    if isinstance(name, tuple) and name[1] in ("quadrics", "tri_alpha"):  # what else the object definition holds: RTXH_TABLE_OBJECT_EXTRA_BASE + 2 * object + {0, 1}
        which, dt = 100000 + 2 * int(name[0]) + (0 if name[1] == "quadrics" else 1), (_TABLES["quadrics"][1] if name[1] == "quadrics" else np.dtype(("<i4", 2)))
    elif isinstance(name, tuple):  # (object index, table): the object-space soup of one ObjectBegin block
        j, dt = _OBJECT_TABLES[name[1]]
        which = 1000 + 8 * int(name[0]) + j
    else:
        which, dt = _TABLES[name]
    n = C.c_uint64()
    _check(lib().rtxh_scene_inspect(handle, which, None, C.c_uint64(0), C.byref(n)), "scene_inspect")
    out = np.zeros(n.value, dt)
    if n.value:
        _check(lib().rtxh_scene_inspect(handle, which, out.ctypes.data_as(C.c_void_p), C.c_uint64(out.nbytes), C.byref(n)), "scene_inspect")
    return out


class PbrtScene(HostScene):
    """A scene read from a pbrt-v3 file by the C++ host (rtxh_pbrt_load): what `rustracer scene.pbrt` builds up to
    `renderer::render` (rc/pbrt/mod.rs:16-27, rc/api.rs:977-1010). Same methods as HostScene."""

    def __init__(self, path=None, text=None, base_dir="", device_ingest=False, flatten_instances=False):
        """flatten_instances=True: every ObjectInstance is written out as world-space triangles (single-level traversal kernels) instead of the
        reference's one tree per object (rc/primitive.rs:79-118), which is the default."""
        L = lib()
        res = PbrtResult()
        L.rtxh_set_device_ingest(1 if device_ingest else 0)  # MIP pyramids / environment tables on the GPU (bit-identical)
        L.rtxh_set_flatten_instances(1 if flatten_instances else 0)
        try:
            if path is not None:
                _check(L.rtxh_pbrt_load(os.fsencode(path), C.byref(res)), "pbrt_load")
            else:
                _check(L.rtxh_pbrt_parse(text.encode(), os.fsencode(base_dir), C.byref(res)), "pbrt_parse")
        finally:
            L.rtxh_set_device_ingest(0)
            L.rtxh_set_flatten_instances(0)
        self.h = C.c_void_p(res.scene)
        self.desc = None
        self.params = RenderParams.from_buffer_copy(res.params)
        self.max_prims_per_node = res.max_prims_per_node
        self.n_warnings = res.n_warnings
        self.first_warning = (L.rtxh_last_error() or b"").decode(errors="replace") if res.n_warnings else ""  # (the loader leaves the first warning's text there)
        self.film_filename = res.film_filename.decode(errors='replace')

    def _render_params(self, rank=0, world_size=1, flags=0):
        p = RenderParams.from_buffer_copy(self.params)
        p.rank, p.world_size, p.flags = rank, world_size, flags
        return p

    def n_lights(self):
        return len(scene_table(self.h, "lights"))


def pbrt_tokens(text: str):
    """The host lexer's token list: [("K", word) | ("N", float) | ("S", string) | ("[", None) | ("]", None)]."""
    buf = C.create_string_buffer(16 * len(text) + 64)
    n = _check(lib().rtxh_pbrt_tokens(text.encode(), buf, C.c_uint64(len(buf))), "pbrt_tokens")
    out = []
    for line in buf.value.decode().split("\n")[:n]:
        if line in ("[", "]"):
            out.append((line, None))
        elif line[0] == "N":
            out.append(("N", float(line[2:])))
        else:
            out.append((line[0], line[2:]))
    return out


def sampler_tables(spp, dims, pixel0, n_pixels, plain=False):
    """K0 on the GPU: (scrambles (n,3*dims) u32, perms (n, 2*dims, spp) u16). plain=True: the single-kernel
    in-order statement of the algorithm (cross-check of the segmented sampler rt_render uses)."""
    spp2 = 1
    while spp2 < spp:
        spp2 *= 2
    sc = np.zeros((n_pixels, 3 * dims), np.uint32)
    pm = np.zeros((n_pixels, 2 * dims, spp2), np.uint16)
    fn = hip_lib().rt_sampler_tables_plain if plain else hip_lib().rt_sampler_tables
    rc = fn(spp, dims, C.c_uint64(pixel0), C.c_uint64(n_pixels), _p(sc, C.c_uint32), _p(pm, C.c_uint16))
    if rc < 0:
        raise BackendError(f"rt_sampler_tables failed ({rc}): {hip_lib().rt_last_error().decode(errors='replace')}")
    return sc, pm


def offset_ray_origin(p, p_error, n, w):
    """offset_ray_origin (rc/geometry/mod.rs:203-220) on the device for (k, 3) float32 arrays: the parity hook of the ulp stepping every spawned ray goes through."""
    a = [np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (p, p_error, n, w)]
    out = np.zeros_like(a[0])
    rc = hip_lib().rt_offset_ray_origin(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), C.c_uint64(a[0].shape[0]), _p(out))
    if rc < 0:
        raise BackendError(f"rt_offset_ray_origin failed ({rc}): {hip_lib().rt_last_error().decode(errors='replace')}")
    return out


def film_to_rgb(film_xyzw: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """Film::write_image's pixel maths (rc/film.rs:196-234) in float32 numpy: XYZ -> RGB, / weight, clamp, * scale."""
    f = np.asarray(film_xyzw, np.float32)
    X, Y, Z, Wt = f[..., 0], f[..., 1], f[..., 2], f[..., 3]
    c = np.float32
    r = c(3.240479) * X - c(1.537150) * Y - c(0.498535) * Z
    g = c(-0.969256) * X + c(1.875991) * Y + c(0.041556) * Z
    b = c(0.055648) * X - c(0.204043) * Y + c(1.057311) * Z
    rgb = np.stack([r, g, b], -1)
    nz = Wt != 0
    inv = np.where(nz, c(1.0) / np.where(nz, Wt, c(1.0)), c(1.0)).astype(np.float32)
    rgb = np.where(nz[..., None], np.maximum(c(0.0), rgb * inv[..., None]), rgb)
    return (rgb * c(scale)).astype(np.float32)


def rgb_to_png8(rgb: np.ndarray) -> np.ndarray:
    """write_image_png's quantisation (rc/imageio.rs:52-63): clamp(255 * gamma_correct(v) + 0.5, 0, 255) as u8, sRGB curve of
    rc/spectrum.rs:387-393."""
    v = np.asarray(rgb, np.float32)
    c = np.float32
    with np.errstate(invalid="ignore"):
        g = np.where(v <= c(0.0031308), c(12.92) * v, c(1.055) * np.power(np.maximum(v, c(0.0)), c(1.0) / c(2.4), dtype=np.float32) - c(0.055)).astype(np.float32)
    q = np.clip(c(255.0) * g + c(0.5), c(0.0), c(255.0))
    return np.nan_to_num(q, nan=0.0).astype(np.uint8)   # Rust's `as u8` maps NaN to 0
