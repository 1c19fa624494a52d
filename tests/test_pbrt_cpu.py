"""pbrt-v3 scene files through the C++ host (`rtxh_pbrt_load`) - CPU only.

Two kinds of checks:
  * round trips: a SceneDesc written by `rustracer_amd.pbrt_export` and read back by the C++ parser must give the same
    triangle soup, BVH, render parameters, lights and (resolved) material / texture trees as the same SceneDesc handed to
    the host call by call (`HostScene`), for the benchmark scenes and the whole material / texture / light zoo;
  * directive semantics on hand-written files: the lexer's known answers (rc/pbrt/lexer.rs:277-337), CTM composition
    order, attribute / transform stacks, named coordinate systems, Include, named materials, parameter defaults of every
    create(), and the reference's error cases (rc/api.rs:179-256, 418-470).
"""
import os
import sys

import numpy as np
import pytest

from rustracer_amd import host, scene_desc as sd
from rustracer_amd.pbrt_export import write_pbrt

F32 = np.float32


# ------------------------------------------------------------------------------------------------ comparison helpers
class _Cmp:
    """Walks the material / texture tables of two host scenes side by side."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.ta, self.tb = a.table("textures"), b.table("textures")
        self.ma, self.mb = a.table("materials"), b.table("materials")
        self.mips_seen = {}

    def mip(self, ia, ib, exact):
        if (ia, ib) in self.mips_seen:
            return
        self.mips_seen[(ia, ib)] = True
        la, lb = self.a.mip_levels(ia), self.b.mip_levels(ib)
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            assert x.shape == y.shape
            if exact:
                assert np.array_equal(x, y)
            else:  # float image maps go through Spectrum::y() when read (imagemap.rs:214-216)
                assert np.allclose(x[..., 0], y[..., 0], rtol=2e-6, atol=1e-7)

    def tex(self, ia, ib, exact=True):
        assert (ia < 0) == (ib < 0), (ia, ib)
        if ia < 0:
            return
        x, y = self.ta[ia], self.tb[ib]
        assert x["kind"] == y["kind"], (x, y)
        k = int(x["kind"])
        if k == sd.TEX_CONST:
            if exact:
                assert np.array_equal(x["value"], y["value"]), (x, y)
            else:
                assert x["value"][0] == y["value"][0]
        elif k in (sd.TEX_SCALE, sd.TEX_MIX):
            self.tex(x["tex1"], y["tex1"], exact)
            self.tex(x["tex2"], y["tex2"], exact)
            if k == sd.TEX_MIX:
                self.tex(x["amount"], y["amount"], False)
        elif k == sd.TEX_IMAGE:
            assert np.array_equal(x["mapping"], y["mapping"])
            self.mip(int(x["image"]), int(y["image"]), exact)
        elif k == sd.TEX_CHECKER:
            assert np.array_equal(x["mapping"], y["mapping"]) and x["amount"] == y["amount"]
            self.tex(x["tex1"], y["tex1"])
            self.tex(x["tex2"], y["tex2"])
        elif k == sd.TEX_UV:
            assert np.array_equal(x["mapping"], y["mapping"])
        elif k == sd.TEX_FBM:
            assert x["value"][0] == y["value"][0] and x["amount"] == y["amount"]
        else:
            raise AssertionError(k)

    FLOAT_SLOTS = {sd.MAT_MATTE: {4}, sd.MAT_PLASTIC: {5}, sd.MAT_METAL: {5, 6, 7}, sd.MAT_GLASS: {6, 7, 8}, sd.MAT_UBER: {5, 6, 7, 8}, sd.MAT_SUBSTRATE: {6, 7},
                   sd.MAT_TRANSLUCENT: {5}, sd.MAT_DISNEY: {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13}, sd.MAT_MIRROR: set(), sd.MAT_MIX: set()}

    def mat(self, ia, ib):
        x, y = self.ma[ia], self.mb[ib]
        assert x["kind"] == y["kind"]
        k = int(x["kind"])
        if k != sd.MAT_MIX:
            assert x["remap_roughness"] == y["remap_roughness"]
            self.tex(int(x["bump"]), int(y["bump"]), False)
        for slot in range(14):
            self.tex(int(x["slot"][slot]), int(y["slot"][slot]), slot not in self.FLOAT_SLOTS[k])
        if k == sd.MAT_MIX:
            self.mat(int(x["slot"][14]), int(y["slot"][14]))
            self.mat(int(x["slot"][15]), int(y["slot"][15]))
        elif k == sd.MAT_DISNEY:
            assert x["slot"][14] == y["slot"][14]


def assert_same_scene(p, h):
    """p: PbrtScene, h: HostScene of the SceneDesc the file was written from."""
    for name in ("P", "N", "UV", "S", "indices", "tri_light", "tri_flags"):
        assert np.array_equal(p.table(name), h.table(name)), name
    bp, bh = p.bvh(), h.bvh()
    for k in bp:
        assert np.array_equal(bp[k], bh[k]), k
    sp, sh = p.setup(), h.setup()
    for name, _ in host.RenderParams._fields_:   # value equality: CTM = identity * LookAt turns a -0 of the LookAt matrix into +0 (api.rs:637)
        x, y = getattr(sp["params"], name), getattr(sh["params"], name)
        assert np.array_equal(np.array(x), np.array(y)), name
    for k in ("raster_to_camera", "dx_camera", "dy_camera", "filter_table", "sample_bounds", "cropped"):
        assert np.array_equal(sp[k], sh[k]), k
    c = _Cmp(p, h)
    tm_p, tm_h = p.table("tri_material"), h.table("tri_material")
    for a, b in sorted(set(zip(tm_p.tolist(), tm_h.tolist()))):
        c.mat(a, b)
    lp, lh = p.table("lights"), h.table("lights")
    assert len(lp) == len(lh)
    for k in ("kind", "tri", "rgb", "two_sided", "vec", "l2w"):
        assert np.array_equal(lp[k], lh[k]), k
    assert np.allclose(lp["w2l"], lh["w2l"], rtol=1e-5, atol=1e-6)
    for a, b in zip(lp, lh):
        if a["kind"] == sd.LIGHT_INFINITE:
            c.mip(int(a["mip"]), int(b["mip"]), True)


def _round_trip(desc, tmp_path):
    path = os.path.join(str(tmp_path), f"{desc.name}.pbrt")
    write_pbrt(desc, path)
    p = host.PbrtScene(path)
    assert p.n_warnings == 0
    assert p.max_prims_per_node == desc.max_prims_per_node
    assert p.film_filename == f"rt-{desc.name}.png"
    assert_same_scene(p, host.HostScene(desc))
    return p


# ------------------------------------------------------------------------------------------------ round trips
def test_cornell_round_trip(tmp_path):
    from rustracer_amd.scenes import cornell_box
    _round_trip(cornell_box(96, 64, 32), tmp_path)


def test_mis_plates_round_trip(tmp_path):
    from rustracer_amd.scenes import mis_plates
    _round_trip(mis_plates(64, 48, 8, sphere_level=1), tmp_path)


def test_room_env_round_trip(tmp_path):
    from rustracer_amd.scenes import room_env
    _round_trip(room_env(64, 36, 4, detail=2, tex_size=32, env_size=32), tmp_path)


def test_large_meshes_round_trip_through_plymesh(tmp_path):
    from rustracer_amd.scenes import blob_scene
    d = blob_scene(nu=48, nv=24, xres=32, yres=32, spp=4)
    path = os.path.join(str(tmp_path), "blob.pbrt")
    text = write_pbrt(d, path, ply_over=100)
    assert 'Shape "plymesh"' in text and os.path.exists(os.path.join(str(tmp_path), "blob_mesh0.ply"))
    assert_same_scene(host.PbrtScene(path), host.HostScene(d))


def _zoo_cases():
    from test_gpu_materials import MATERIALS
    # float checkerboard / uv textures (bump maps of two zoo entries) do not exist in the reference's make_float_texture (api.rs:1201-1216)
    return [m for m in MATERIALS if m not in ("mirror_bump_checker", "mix_bump_both")]


@pytest.mark.parametrize("material", _zoo_cases())
def test_material_zoo_round_trip(tmp_path, material):
    from test_gpu_materials import _zoo
    d = _zoo(material)
    d.name = material
    _round_trip(d, tmp_path)


@pytest.mark.parametrize("light", ["area_two_sided", "point", "distant", "infinite"])
def test_light_zoo_round_trip(tmp_path, light):
    from test_gpu_materials import _zoo
    d = _zoo("plastic", light)
    d.film.filter_kind, d.film.filter_params = {"point": (sd.FILTER_GAUSSIAN, (1.5, 2.5, 1.7, 0.0)), "distant": (sd.FILTER_MITCHELL, (2.0, 3.0, 0.4, 0.3)),
                                                 "infinite": (sd.FILTER_TRIANGLE, (1.5, 1.0, 0.0, 0.0))}.get(light, (sd.FILTER_BOX, (0.5, 0.5, 0.0, 0.0)))
    d.film.crop = (0.1, 0.9, 0.25, 0.75)
    d.film.scale, d.film.max_sample_luminance = 2.0, 50.0
    d.camera.lens_radius, d.camera.focal_distance = 0.05, 5.0
    d.integrator.light_strategy, d.integrator.rr_threshold, d.integrator.pixel_bounds = "uniform", 0.25, (2, 30, 3, 20)
    d.max_prims_per_node = 2
    d.name = light
    _round_trip(d, tmp_path)


# ------------------------------------------------------------------------------------------------ lexer known answers
def test_lexer_known_answers():
    # rc/pbrt/lexer.rs:316-337: float, string, keyword / bracket and comment parsers
    assert host.pbrt_tokens("-1.23e2") == [("N", -123.0)]
    assert host.pbrt_tokens('"this is a string"') == [("S", "this is a string")]
    assert host.pbrt_tokens("Accelerator") == [("K", "Accelerator")]
    assert host.pbrt_tokens("[") == [("[", None)]
    assert host.pbrt_tokens("#foo\n") == []
    toks = host.pbrt_tokens('LookAt 0 0 5 0 0 0 0 1 0\nCamera "perspective" "float fov" [50] # trailing\n  #whole line\nWorldBegin\nShape "sphere"\nWorldEnd')
    assert toks[0] == ("K", "LookAt") and [t[1] for t in toks[1:10]] == [0, 0, 5, 0, 0, 0, 0, 1, 0]
    assert toks[10:] == [("K", "Camera"), ("S", "perspective"), ("S", "float fov"), ("[", None), ("N", 50.0), ("]", None), ("K", "WorldBegin"), ("K", "Shape"),
                         ("S", "sphere"), ("K", "WorldEnd")]
    assert host.pbrt_tokens('"a""b"[1 2]') == [("S", "a"), ("S", "b"), ("[", None), ("N", 1.0), ("N", 2.0), ("]", None)]
    with pytest.raises(host.BackendError, match="unterminated"):
        host.pbrt_tokens('Shape "open')


# ------------------------------------------------------------------------------------------------ directive semantics
HEADER = 'Sampler "02sequence"\n'
TRI = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\n'


def _parse(text, base_dir=""):
    return host.PbrtScene(text=text, base_dir=str(base_dir))


def _mm(a, b):  # Matrix4x4 product in f32, the summation order of matrix.rs:157-168
    r = np.zeros((4, 4), F32)
    for i in range(4):
        for j in range(4):
            r[i, j] = F32(F32(F32(a[i, 0] * b[0, j]) + F32(a[i, 1] * b[1, j])) + F32(a[i, 2] * b[2, j])) + F32(a[i, 3] * b[3, j])
    return r


def _pt(m, p):  # Transform * Point3f (transform.rs:264-286), w == 1 here
    x, y, z = (F32(v) for v in p)
    return np.array([F32(F32(F32(m[r, 0] * x) + F32(m[r, 1] * y)) + F32(m[r, 2] * z)) + m[r, 3] for r in range(3)], F32)


def _translate(x, y, z):
    m = np.eye(4, dtype=F32)
    m[:3, 3] = (x, y, z)
    return m


def _scale(x, y, z):
    return np.diag(np.array([x, y, z, 1], F32))


def test_defaults_of_every_create():
    p = _parse(HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    q = p.params
    assert (q.xres, q.yres) == (1280, 720) and list(q.crop) == [0, 1, 0, 1]                       # film.rs:124-126
    assert q.filter_kind == 0 and list(q.filter_params)[:2] == [0.5, 0.5]                         # api.rs:283 + boxfilter.rs
    assert q.film_scale == 1.0 and q.max_sample_luminance == float("inf")
    assert q.fov == 90.0 and q.lens_radius == 0.0 and q.focal_distance == F32(1e6)                # camera.rs:84-108
    assert (q.spp, q.sampler_dims) == (16, 4)                                                     # zerotwosequence.rs:58-63
    assert (q.max_depth, q.rr_threshold, q.light_strategy) == (5, 1.0, 0)                         # path.rs:49-53
    assert list(q.pixel_bounds) == [0, 0, 0, 0] and q.has_pixel_bounds == 0
    assert np.array_equal(np.array(q.cam_to_world), np.eye(4, dtype=F32).reshape(-1))
    assert p.max_prims_per_node == 4 and p.film_filename == "image.png"                           # bvh/mod.rs:76, film.rs:118-123
    m = p.table("materials")
    t = p.table("textures")
    assert len(m) == 1 and m[0]["kind"] == sd.MAT_MATTE                                           # api.rs:304 default material "matte"
    assert np.array_equal(t[m[0]["slot"][0]]["value"], F32([0.5, 0.5, 0.5])) and t[m[0]["slot"][4]]["value"][0] == 0.0   # matte.rs:24-25
    assert p.n_lights() == 0 and p.n_warnings == 0


@pytest.mark.parametrize("name, strategy", [("uniform", 1), ("spatial", 0), ("power", 0), ("anything-else", 0)])
def test_every_light_sample_strategy_but_uniform_is_spatial(name, strategy):
    # PathIntegrator::preprocess (path.rs:86-94) compares the string with "uniform" only; pbrt scene files carry "power"
    p = _parse(f'Integrator "path" "string lightsamplestrategy" "{name}"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    assert p.params.light_strategy == strategy and p.n_warnings == 0


def test_halton_default_sampler_is_an_error_like_the_reference():
    with pytest.raises(host.BackendError, match='Sampler "halton" unknown'):   # api.rs:285 default + :205-215
        _parse("WorldBegin\n" + TRI + "WorldEnd\n")


def test_ctm_composition_and_stacks():
    text = HEADER + """WorldBegin
Translate 1 2 3
Scale 2 2 2
""" + TRI + """AttributeBegin
  Translate 10 0 0
  TransformBegin
    Scale -1 1 1
""" + TRI + """  TransformEnd
""" + TRI + """AttributeEnd
""" + TRI + "WorldEnd\n"
    p = _parse(text)
    P = p.table("P")
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], F32)
    ctm0 = _mm(_translate(1, 2, 3), _scale(2, 2, 2))          # cur_transform = cur_transform * t (api.rs:536-556)
    ctm1 = _mm(ctm0, _translate(10, 0, 0))
    ctm2 = _mm(ctm1, _scale(-1, 1, 1))
    for k, m in enumerate((ctm0, ctm2, ctm1, ctm0)):
        assert np.array_equal(P[3 * k:3 * k + 3], np.stack([_pt(m, v) for v in base])), k
    flags = p.table("tri_flags")
    assert [int(f) & sd.TRI_FLIP for f in flags] == [0, sd.TRI_FLIP, 0, 0]   # swaps_handedness of the mirrored CTM (transform.rs:255-261)


def test_reverse_orientation_is_graphics_state():
    p = _parse(HEADER + "WorldBegin\nAttributeBegin\nReverseOrientation\n" + TRI + "Scale 1 1 -1\n" + TRI + "AttributeEnd\n" + TRI + "WorldEnd\n")
    assert [int(f) & sd.TRI_FLIP for f in p.table("tri_flags")] == [sd.TRI_FLIP, 0, 0]   # reverse ^ swaps_handedness (mesh.rs:175-180)


def test_rotate_and_concat_transform_and_lookat_camera():
    text = """LookAt 1 2 3  0 0 0  0 1 0
Camera "perspective" "float fov" [40]
""" + HEADER + """WorldBegin
Rotate 90 0 0 1
ConcatTransform [1 0 0 0  0 1 0 0  0 0 1 0  5 6 7 1]
""" + TRI + """CoordSysTransform "camera"
""" + TRI + "WorldEnd\n"
    p = _parse(text)
    P = p.table("P")
    s, c = F32(np.sin(np.radians(F32(90)))), F32(np.cos(np.radians(F32(90))))
    rot = np.eye(4, dtype=F32)
    rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1] = c, -s, s, c     # Transform::rotate about z (transform.rs:30-56): m[0][0] = 0 + 1*cos
    ctm = _mm(rot, _translate(5, 6, 7))                          # the file gives the matrix column-major (api.rs:596-600)
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], F32)
    assert np.allclose(P[:3], np.stack([_pt(ctm, v) for v in base]), rtol=0, atol=1e-6)
    w2c, c2w = host.look_at((1, 2, 3), (0, 0, 0), (0, 1, 0))
    assert np.array_equal(np.array(p.params.cam_to_world, F32).reshape(4, 4), c2w)       # camera_to_world = CTM.inverse() (api.rs:726)
    assert np.array_equal(P[3:6], np.stack([_pt(c2w, v) for v in base]))                 # named coordinate system "camera" (:728)
    assert p.params.fov == 40.0


def test_camera_screen_window_parameters(tmp_path):
    from rustracer_amd.scenes import cornell_box
    p = _parse('Camera "perspective" "float frameaspectratio" [2]\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    assert list(p.params.screen_window) == [-2.0, 2.0, -1.0, 1.0]                                   # camera.rs:86-97
    p = _parse('Camera "perspective" "float frameaspectratio" [0.5] "float screenwindow" [-1 2 -3 4]\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    assert list(p.params.screen_window) == [-1.0, 2.0, -3.0, 4.0]                                   # :98-107 overrides
    p = _parse('Camera "perspective"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    assert list(p.params.screen_window) == [0.0, 0.0, 0.0, 0.0]                                     # default: derived from the resolution at set-up
    d = cornell_box(48, 32, 4)
    d.camera.frame_aspect, d.camera.screen_window, d.name = 0.75, (-0.6, 0.9, -1.1, 0.8), "win"
    _round_trip(d, tmp_path)


def test_world_begin_resets_the_ctm_and_options_are_refused_in_the_world_block():
    p = _parse("Translate 100 0 0\n" + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n")
    assert np.array_equal(p.table("P")[1], F32([1, 0, 0]))                               # api.rs:741
    with pytest.raises(host.BackendError, match="Options cannot be set inside world block"):
        _parse(HEADER + 'WorldBegin\nFilm "image"\n' + TRI + "WorldEnd\n")
    with pytest.raises(host.BackendError, match="must be inside world block"):
        _parse(HEADER + TRI + "WorldBegin\nWorldEnd\n")
    with pytest.raises(host.BackendError, match="WorldEnd"):
        _parse(HEADER + "WorldBegin\n" + TRI)


def test_named_materials_textures_and_lookup_order():
    text = HEADER + """WorldBegin
Texture "grid" "spectrum" "checkerboard" "float uscale" [4] "float vscale" [6] "rgb tex1" [0.9 0.1 0.1] "rgb tex2" [0.1 0.1 0.9]
Texture "bumps" "float" "scale" "float tex1" [0.25] "float tex2" [2]
MakeNamedMaterial "a" "string type" "plastic" "texture Kd" "grid" "float roughness" [0.3] "bool remaproughness" "false" "texture bumpmap" "bumps"
MakeNamedMaterial "b" "string type" "mirror"
MakeNamedMaterial "ab" "string type" "mix" "string namedmaterial1" "a" "string namedmaterial2" "b" "rgb amount" [0.2 0.4 0.6]
Material "matte" "rgb Kd" [0.1 0.2 0.3] "float sigma" [20]
""" + TRI + TRI.replace('"point P"', '"rgb Kd" [0.7 0.8 0.9] "point P"') + """NamedMaterial "ab"
""" + TRI + """Material "glass" "float eta" [1.7]
""" + TRI + """NamedMaterial "missing"
""" + TRI + "WorldEnd\n"
    p = _parse(text)
    m, t, tm = p.table("materials"), p.table("textures"), p.table("tri_material")
    m0, m1, m2, m3, m4 = (m[i] for i in tm)
    assert m0["kind"] == sd.MAT_MATTE and np.array_equal(t[m0["slot"][0]]["value"], F32([0.1, 0.2, 0.3])) and t[m0["slot"][4]]["value"][0] == 20.0
    assert np.array_equal(t[m1["slot"][0]]["value"], F32([0.7, 0.8, 0.9]))       # shape parameters override the Material's (paramset.rs:421-424)
    assert t[m1["slot"][4]]["value"][0] == 20.0
    assert m2["kind"] == sd.MAT_MIX and np.array_equal(t[m2["slot"][13]]["value"], F32([0.2, 0.4, 0.6]))
    a, b = m[m2["slot"][14]], m[m2["slot"][15]]
    assert a["kind"] == sd.MAT_PLASTIC and a["remap_roughness"] == 0 and b["kind"] == sd.MAT_MIRROR
    grid = t[a["slot"][0]]
    assert grid["kind"] == sd.TEX_CHECKER and list(grid["mapping"]) == [4, 6, 0, 0] and grid["amount"] == 1
    assert np.array_equal(t[a["slot"][1]]["value"], F32([0.25, 0.25, 0.25])) and t[a["slot"][5]]["value"][0] == F32(0.3)
    bump = t[a["bump"]]
    assert bump["kind"] == sd.TEX_SCALE and t[bump["tex1"]]["value"][0] == 0.25 and t[bump["tex2"]]["value"][0] == 2.0
    assert np.array_equal(t[b["slot"][2]]["value"], F32([0.9, 0.9, 0.9]))        # mirror.rs:22
    assert m3["kind"] == sd.MAT_GLASS and t[m3["slot"][8]]["value"][0] == F32(1.7)   # "eta" before "index" (glass.rs:32-34)
    assert m4["kind"] == sd.MAT_MATTE and p.n_warnings == 1                     # api.rs:318-326: unknown named material -> matte, logged
    # the Material directive cleared the named material (api.rs:890), then NamedMaterial took over again


def test_attribute_end_restores_material_and_area_light():
    text = HEADER + """WorldBegin
AttributeBegin
  Material "mirror"
  AreaLightSource "diffuse" "rgb L" [2 3 4] "rgb scale" [0.5 0.5 2] "bool twosided" "true"
""" + TRI + TRI + """AttributeEnd
LightSource "point" "rgb I" [1 2 3] "point from" [1 1 1]
""" + TRI + """AttributeBegin
  Translate 0 0 5
  LightSource "point" "point from" [1 1 1] "rgb scale" [2 2 2]
  LightSource "distant" "point from" [0 0 0] "point to" [0 0 2]
AttributeEnd
WorldEnd
"""
    p = _parse(text)
    l = p.table("lights")
    assert list(l["kind"]) == [sd.LIGHT_DIFFUSE_AREA, sd.LIGHT_DIFFUSE_AREA, sd.LIGHT_POINT, sd.LIGHT_POINT, sd.LIGHT_DISTANT]   # appearance order (api.rs:904, 958-963)
    assert list(l["tri"][:2]) == [0, 1] and np.array_equal(l["rgb"][0], F32([1, 1.5, 8])) and list(l["two_sided"][:2]) == [1, 1]  # L * scale (diffuse.rs:44-50)
    assert list(p.table("tri_light")) == [0, 1, -1]
    m, tm = p.table("materials"), p.table("tri_material")
    assert m[tm[0]]["kind"] == sd.MAT_MIRROR and m[tm[2]]["kind"] == sd.MAT_MATTE
    assert np.array_equal(l["vec"][2], F32([1, 1, 1])) and np.array_equal(l["rgb"][2], F32([1, 2, 3]))
    assert np.array_equal(l["vec"][3], F32([1, 1, 6])) and np.array_equal(l["rgb"][3], F32([2, 2, 2]))     # translate(from) * l2w (point.rs:33-34)
    assert np.array_equal(l["vec"][4], F32([0, 0, -1]))                                                     # l2w * (from - to), normalised (distant.rs:27,39-40)


def test_include_and_plymesh_and_imagemap(tmp_path):
    from rustracer_amd.ingest import write_pfm, write_ply
    P = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], F32)
    write_ply(os.path.join(str(tmp_path), "quad.ply"), P, np.array([[0, 1, 2], [0, 2, 3]], np.int32), UV=P[:, :2].copy())
    rng = np.random.default_rng(5)
    img = rng.random((4, 8, 3), dtype=F32)
    write_pfm(os.path.join(str(tmp_path), "tex.pfm"), img)
    os.makedirs(os.path.join(str(tmp_path), "geo"))
    with open(os.path.join(str(tmp_path), "geo", "inc.pbrt"), "w") as f:
        f.write('Texture "img" "spectrum" "imagemap" "string filename" "tex.pfm" "float scale" [0.5] "string wrap" "clamp" "bool trilinear" "true"\n'
                'Material "matte" "texture Kd" "img"\nShape "plymesh" "string filename" "quad.ply"\n')
    main = os.path.join(str(tmp_path), "main.pbrt")
    with open(main, "w") as f:
        f.write(HEADER + 'WorldBegin\nTranslate 0 0 2\nInclude "geo/inc.pbrt"\n' + TRI + "WorldEnd\n")
    p = host.PbrtScene(main)
    assert len(p.table("indices")) == 3 and np.array_equal(p.table("P")[:4], P + F32([0, 0, 2]))
    assert [int(f) for f in p.table("tri_flags")] == [sd.TRI_HAS_UV, sd.TRI_HAS_UV, 0]
    t = p.table("textures")
    kd = t[p.table("materials")[p.table("tri_material")[0]]["slot"][0]]
    assert kd["kind"] == sd.TEX_IMAGE
    lv = p.mip_levels(int(kd["image"]))
    assert np.array_equal(lv[0], F32(0.5) * img[::-1])           # y flip + scale, no gamma for .pfm (imagemap.rs:52-84, 124-127)
    # a missing image is a 1x1 grey texel, logged, as in the reference (imagemap.rs:62-69)
    q = _parse(HEADER + 'WorldBegin\nTexture "img" "spectrum" "imagemap" "string filename" "nope.pfm"\nMaterial "matte" "texture Kd" "img"\n' + TRI + "WorldEnd\n", tmp_path)
    assert q.n_warnings == 1 and np.array_equal(q.mip_levels(0)[0], np.full((1, 1, 3), 0.18, F32))


def test_png_imagemap_is_gamma_decoded_by_default(tmp_path):
    from rustracer_amd.ingest import write_png
    a = np.random.default_rng(2).integers(0, 256, (4, 4, 3))
    write_png(os.path.join(str(tmp_path), "t.png"), a, 2, 8)
    base = HEADER + 'WorldBegin\nTexture "img" "spectrum" "imagemap" "string filename" "t.png" %s\nMaterial "matte" "texture Kd" "img"\n' + TRI + "WorldEnd\n"
    v = (a.astype(F32) / F32(255))[::-1]
    lin = np.where(v <= F32(0.04045), v / F32(12.92), ((v + F32(0.055)) * F32(1.0) / F32(1.055)) ** F32(2.4)).astype(F32)   # spectrum.rs:379-385
    assert np.allclose(_parse(base % "", tmp_path).mip_levels(0)[0], lin, rtol=3e-7, atol=0)          # gamma defaults to true for png / tga (imagemap.rs:124-127)
    assert np.array_equal(_parse(base % '"bool gamma" "false"', tmp_path).mip_levels(0)[0], v)
    f = _parse((base % '"float scale" [2]').replace('"spectrum"', '"float"').replace('"texture Kd" "img"', '"texture sigma" "img"'), tmp_path).mip_levels(0)[0]
    y = F32(2) * lin
    assert np.allclose(f[..., 0], F32(0.212671) * y[..., 0] + F32(0.715160) * y[..., 1] + F32(0.072169) * y[..., 2], rtol=1e-6)   # float maps keep y() (imagemap.rs:214-216)


def test_infinite_light_texels_carry_the_scale(tmp_path):
    from rustracer_amd.ingest import write_pfm
    img = np.random.default_rng(9).random((8, 16, 3), dtype=F32)
    write_pfm(os.path.join(str(tmp_path), "env.pfm"), img)
    p = _parse(HEADER + 'WorldBegin\nLightSource "infinite" "string mapname" "env.pfm" "rgb L" [2 1 0.5] "rgb scale" [0.5 1 2]\n' + TRI + "WorldEnd\n", tmp_path)
    assert np.array_equal(p.mip_levels(0)[0], img * (F32([2, 1, 0.5]) * F32([0.5, 1, 2])))   # infinite.rs:60, 117-127: no flip, texel * (L * scale)
    q = _parse(HEADER + 'WorldBegin\nLightSource "infinite" "rgb L" [3 2 1]\n' + TRI + "WorldEnd\n", tmp_path)
    assert np.array_equal(q.mip_levels(0)[0], F32([[[3, 2, 1]]])) and q.n_lights() == 1       # no map: one texel of `power` (:62-69)


def test_object_instances_are_written_out():
    mesh = ('Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0 0 0 1 0 0 1 1 0 0 1 0] "normal N" [0 0 1 0 0 1 0.6 0 0.8 0 0.6 0.8] '
            '"vector S" [1 0 0 1 0 0 0.8 0 -0.6 1 0 0] "float uv" [0 0 1 0 1 1 0 1]\n')
    text = HEADER + """WorldBegin
Material "mirror"
""" + TRI + """ObjectBegin "thing"
  Material "plastic"
  Translate 0 0 1
""" + mesh + """  ReverseOrientation
""" + TRI + """ObjectEnd
ObjectBegin "empty"
ObjectEnd
""" + TRI + """AttributeBegin
  Translate 5 0 0
  Scale 2 1 1
  ObjectInstance "thing"
AttributeEnd
ObjectInstance "empty"
Scale 1 -3 1
ObjectInstance "thing"
WorldEnd
"""
    p = host.PbrtScene(text=text, flatten_instances=True)
    P, N, S, UV = p.table("P"), p.table("N"), p.table("S"), p.table("UV")
    idx, flags, tm, m = p.table("indices"), p.table("tri_flags"), p.table("tri_material"), p.table("materials")
    assert len(idx) == 2 + 2 * 3 and len(P) == 3 + 3 + 2 * 7
    assert [m[i]["kind"] for i in tm] == [sd.MAT_MIRROR, sd.MAT_MIRROR] + [sd.MAT_PLASTIC] * 6      # ObjectEnd restores the graphics state (api.rs:1046)
    obj_p = np.array([[0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1], [0, 0, 1], [1, 0, 1], [0, 1, 1]], F32)   # instance space: CTM inside the definition
    obj_n = np.array([[0, 0, 1], [0, 0, 1], [0.6, 0, 0.8], [0, 0.6, 0.8]] + [[0, 0, 0]] * 3, F32)
    obj_s = np.array([[1, 0, 0], [1, 0, 0], [0.8, 0, -0.6], [1, 0, 0]] + [[0, 0, 0]] * 3, F32)
    m1 = _mm(_translate(5, 0, 0), _scale(2, 1, 1))
    m2 = _scale(1, -3, 1)
    for v0, mtx, inv_diag in ((6, m1, (0.5, 1, 1)), (13, m2, (1, F32(1) / F32(-3), 1))):
        assert np.array_equal(P[v0:v0 + 7], np.stack([_pt(mtx, v) for v in obj_p]))
        assert np.allclose(N[v0:v0 + 7], obj_n * F32(inv_diag), rtol=1e-6, atol=0)                 # inverse transpose (transform.rs:244-253)
        assert np.array_equal(S[v0:v0 + 7], obj_s * np.diag(mtx)[:3])
        assert np.array_equal(UV[v0:v0 + 4], F32([[0, 0], [1, 0], [1, 1], [0, 1]]))
    base = sd.TRI_HAS_N | sd.TRI_HAS_UV | sd.TRI_HAS_S
    assert [int(f) for f in flags] == [0, 0, base, base, sd.TRI_FLIP, base | sd.TRI_FLIP, base | sd.TRI_FLIP, 0]   # the mirrored instance toggles the flip
    assert np.array_equal(idx[2:5], [[6, 7, 8], [6, 8, 9], [10, 11, 12]]) and np.array_equal(idx[5:], [[13, 14, 15], [13, 15, 16], [17, 18, 19]])
    assert (p.table("tri_light") == -1).all()


def test_object_instances_keep_the_references_form_by_default():
    """ObjectBegin / ObjectInstance as the reference holds them (rc/api.rs:1019-1090, rc/primitive.rs:79-118): the object's soup stays in object space - one
    copy however often it is used -, each instance is the CTM at its ObjectInstance with the inverse the Transform carries; an object that is never
    instantiated (or is empty) leaves nothing behind."""
    text = HEADER + """WorldBegin
Material "mirror"
""" + TRI + """ObjectBegin "thing"
  Material "plastic"
  Translate 0 0 1
""" + TRI + """  ReverseOrientation
""" + TRI + """ObjectEnd
ObjectBegin "unused"
""" + TRI + """ObjectEnd
ObjectBegin "empty"
ObjectEnd
AttributeBegin
  Translate 5 0 0
  Scale 2 1 1
  ObjectInstance "thing"
AttributeEnd
ObjectInstance "empty"
Scale 1 -3 1
ObjectInstance "thing"
WorldEnd
"""
    p = _parse(text)
    assert len(p.table("indices")) == 1 and len(p.table("P")) == 3                       # the top level holds its own triangle only
    inst = p.table("instances")
    assert len(inst) == 2 and list(inst["object"]) == [0, 0]                               # both uses share object 0; "unused" and "empty" were never handed over
    m1 = _mm(_translate(5, 0, 0), _scale(2, 1, 1))
    assert np.array_equal(inst["o2w"][0], m1) and np.array_equal(inst["o2w"][1], _scale(1, -3, 1))
    assert np.allclose(inst["w2o"][0] @ inst["o2w"][0], np.eye(4), atol=1e-6) and np.allclose(inst["w2o"][1] @ inst["o2w"][1], np.eye(4), atol=1e-6)
    oP, oi, of, om = p.table((0, "P")), p.table((0, "indices")), p.table((0, "tri_flags")), p.table((0, "tri_material"))
    assert np.array_equal(oP, F32([[0, 0, 1], [1, 0, 1], [0, 1, 1]] * 2)) and np.array_equal(oi, [[0, 1, 2], [3, 4, 5]])   # the CTM inside the definition applied, the instance's not
    assert [int(f) for f in of] == [0, sd.TRI_FLIP]
    m = p.table("materials")
    assert [m[i]["kind"] for i in om] == [sd.MAT_PLASTIC] * 2 and m[p.table("tri_material")[0]]["kind"] == sd.MAT_MIRROR
    b = p.bvh()
    assert len(b["ordered"]) == 3 and sorted(b["ordered"]) == [0, 1, 2]                   # one triangle and two instance boxes in the top-level tree
    with pytest.raises(host.BackendError, match="Unable to find instance"):
        _parse(text.replace('ObjectInstance "empty"', 'ObjectInstance "nope"'))


def test_area_lights_inside_object_definitions_stay_out_of_the_light_list():
    """A shape under an AreaLightSource inside ObjectBegin .. ObjectEnd keeps its area light and the scene's light list does not get it
    (rc/api.rs:934-964: `prims` go to the instance, `area_lights` are dropped): the triangles name an unlisted emitter, in both instancing forms."""
    text = (HEADER + 'WorldBegin\nLightSource "point" "rgb I" [1 1 1]\n' + TRI + 'ObjectBegin "lamp"\n' + TRI +
            'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [3 2 1] "rgb scale" [2 2 2] "bool twosided" "true"\nTranslate 0 0 1\n' + TRI + TRI + 'AttributeEnd\n' + TRI +
            'ObjectEnd\nObjectInstance "lamp"\nTranslate 4 0 0\nObjectInstance "lamp"\nWorldEnd\n')
    p = _parse(text)
    assert p.n_lights() == 1 and len(p.table("instances")) == 2
    assert list(p.table((0, "tri_emitter"))) == [-1, 0, 1, -1]                                   # one area light per Shape statement (make_area_light, api.rs:934-942)
    em = p.table("emitters")
    assert len(em) == 2 and np.array_equal(em["rgb"][0], F32([6, 4, 2])) and em["two_sided"][1] == 1
    q = host.PbrtScene(text=text, flatten_instances=True)
    assert q.n_lights() == 1 and list(q.table("tri_light")) == [-1] + [-1, -2, -3, -1] * 2      # -2 - k: unlisted emitter k
    assert len(q.table("emitters")) == 2


def test_quadrics_inside_object_definitions_are_placed_by_every_instance():
    """The reference wraps whatever an object holds in a TransformedPrimitive (rc/api.rs:1053-1090, primitive.rs:79-118). Round 6: so does the host - a quadric of an
    object definition stays IN the object, in object space (its own object_to_world = the CTM inside the definition), and every ObjectInstance places the object.
    Written out (flatten_instances) an instance of it is the same quadric under instance_to_world * object_to_world: the same tables as the scene with the quadrics
    written at the top level under the composed transforms. An area light on it stays out of the light list (api.rs:954-964) either way."""
    obj = ('ObjectBegin "o"\n' + TRI + 'AttributeBegin\nTranslate 0 1 0\nScale 1 2 1\nShape "sphere" "float radius" 0.5 "float zmax" 0.25\nAttributeEnd\n'
           'Rotate 30 0 0 1\nShape "cylinder" "float radius" 0.2\nObjectEnd\n')
    text = HEADER + 'WorldBegin\nLightSource "point" "rgb I" [1 1 1]\n' + TRI + obj + 'Translate 3 0 0\nObjectInstance "o"\nRotate 45 0 1 0\nTranslate 0 0 2\nObjectInstance "o"\nWorldEnd\n'
    flat = (HEADER + 'WorldBegin\nLightSource "point" "rgb I" [1 1 1]\n' + TRI +
            'AttributeBegin\nTranslate 3 0 0\nAttributeBegin\nTranslate 0 1 0\nScale 1 2 1\nShape "sphere" "float radius" 0.5 "float zmax" 0.25\nAttributeEnd\n'
            'Rotate 30 0 0 1\nShape "cylinder" "float radius" 0.2\nAttributeEnd\n'
            'AttributeBegin\nTranslate 3 0 0\nRotate 45 0 1 0\nTranslate 0 0 2\nAttributeBegin\nTranslate 0 1 0\nScale 1 2 1\nShape "sphere" "float radius" 0.5 "float zmax" 0.25\nAttributeEnd\n'
            'Rotate 30 0 0 1\nShape "cylinder" "float radius" 0.2\nAttributeEnd\nWorldEnd\n')
    p, q = _parse(text), _parse(flat)
    # two-level (the default): the object holds its triangle AND its two quadrics, in object space; nothing at the top level
    assert len(p.table("instances")) == 2 and len(p.table((0, "indices"))) == 1 and len(p.table("quadrics")) == 0
    oq = p.table((0, "quadrics"))
    assert len(oq) == 2 and list(oq["kind"]) == [0, 2] and list(oq["light"]) == [-1, -1]
    one = _parse(HEADER + 'WorldBegin\n' + TRI + 'AttributeBegin\nTranslate 0 1 0\nScale 1 2 1\nShape "sphere" "float radius" 0.5 "float zmax" 0.25\nAttributeEnd\n'
                 'Rotate 30 0 0 1\nShape "cylinder" "float radius" 0.2\nWorldEnd\n').table("quadrics")   # the same statements at the top level of an identity CTM
    for k in oq.dtype.names:
        assert np.array_equal(oq[k], one[k]), k
    # written out: the quadrics at the top level under the composed transforms
    f = host.PbrtScene(text=text, flatten_instances=True)
    sp, sq = f.table("quadrics"), q.table("quadrics")
    assert len(sp) == 4 and list(sp["kind"]) == [0, 2, 0, 2] and len(f.table("instances")) == 0
    for k in sp.dtype.names:
        if k in ("o2w", "w2o"):
            assert np.allclose(sp[k], sq[k], rtol=0, atol=1e-6), k   # (the file spells the product out statement by statement: the same matrices up to the order of the roundings)
        else:
            assert np.array_equal(sp[k], sq[k]), k
    # an emitting quadric of an object: unlisted, one emitter per Shape statement, shared by its placements
    text_e = text.replace('Shape "sphere"', 'AreaLightSource "diffuse" "rgb L" [5 4 3]\nShape "sphere"')
    e = _parse(text_e)
    assert e.n_lights() == 1 and len(e.table("emitters")) == 1 and np.array_equal(e.table("emitters")["rgb"][0], F32([5, 4, 3]))
    assert list(e.table((0, "quadrics"))["light"]) == [-2, -1]  # -2 - k: unlisted emitter k; the cylinder emits nothing
    ef = host.PbrtScene(text=text_e, flatten_instances=True)
    assert list(ef.table("quadrics")["light"]) == [-2, -1, -2, -1]


def test_a_redefined_object_is_what_later_instances_place():
    """ObjectBegin on a name already in use replaces the definition (instances.insert(name, Vec::new()), rc/api.rs:1030): instances placed afterwards use
    the new one, in the two-level form as in the written-out form."""
    tri2 = TRI.replace("Shape", "Translate 0 0 7\nShape")
    text = HEADER + 'WorldBegin\n' + TRI + 'ObjectBegin "a"\n' + TRI + 'ObjectEnd\nObjectInstance "a"\nObjectBegin "a"\n' + tri2 + TRI + 'ObjectEnd\nTranslate 1 0 0\nObjectInstance "a"\nWorldEnd\n'
    p = _parse(text)
    inst = p.table("instances")
    assert len(inst) == 2 and list(inst["object"]) == [0, 1]
    assert len(p.table((0, "indices"))) == 1 and len(p.table((1, "indices"))) == 2
    assert p.table((1, "P"))[0, 2] == 7.0
    q = host.PbrtScene(text=text, flatten_instances=True)
    assert len(q.table("indices")) == 1 + 1 + 2
    zs = sorted(float(z) for z in q.table("P")[:, 2])
    assert zs.count(7.0) == 6  # both triangles of the second definition (the Translate stays in force inside it), none of the first


@pytest.mark.parametrize("text, message", [
    (HEADER + 'WorldBegin\nObjectInstance "nothing"\n' + TRI + "WorldEnd\n", "Unable to find instance named nothing"),
    (HEADER + 'WorldBegin\nObjectBegin "a"\nObjectBegin "b"\nObjectEnd\nObjectEnd\n' + TRI + "WorldEnd\n", "inside of instance definition"),
    (HEADER + 'WorldBegin\nObjectEnd\n' + TRI + "WorldEnd\n", "outside of instance definition"),
    ('Film "other"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n", 'Film "other" unknown'),
    ('PixelFilter "sinc"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n", 'Filter "sinc" unknown'),
    ('Camera "orthographic"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n", 'Camera "orthographic" unknown'),
    ('Integrator "whitted"\n' + HEADER + "WorldBegin\n" + TRI + "WorldEnd\n", "not supported"),
    (HEADER + 'WorldBegin\nShape "cone"\nWorldEnd\n', "not supported"),
    (HEADER + 'WorldBegin\nLightSource "spot"\n' + TRI + "WorldEnd\n", "not supported"),
    (HEADER + 'WorldBegin\nAreaLightSource "sphere"\n' + TRI + "WorldEnd\n", "unknown"),
    (HEADER + 'WorldBegin\nMakeNamedMaterial "m" "rgb Kd" [1 1 1]\n' + TRI + "WorldEnd\n", 'No parameter string "type"'),
    (HEADER + 'WorldBegin\nMaterial "matte" "spectrum Kd" [400 1 700 1]\n' + TRI + "WorldEnd\n", "inline samples"),
    (HEADER + 'WorldBegin\nMaterial "matte" "xyz Kd" [1 1 1]\n' + TRI + "WorldEnd\n", "not supported"),
    (HEADER + "WorldBegin\nFrobnicate 1 2\nWorldEnd\n", "unknown directive"),
    (HEADER + 'WorldBegin\nShape "trianglemesh" "integer indices" [0 1 5] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n', "out of range"),
    (HEADER + "WorldBegin\nWorldEnd\n", "no triangles"),
])
def test_errors(text, message):
    with pytest.raises(host.BackendError, match=message):
        _parse(text)


def test_unknown_material_and_coordinate_system_are_logged_not_fatal():
    p = _parse(HEADER + 'WorldBegin\nCoordSysTransform "nowhere"\nMaterial "velvet"\n' + TRI + "AttributeEnd\nTransformEnd\nWorldEnd\n")
    assert p.n_warnings == 4 and p.table("materials")[0]["kind"] == sd.MAT_MATTE   # api.rs:663-667, 1178-1181, 757-760, 779-782


def test_damaged_files_fail_cleanly(tmp_path):
    """NUL bytes, truncation and random damage end in BackendError (or a valid scene), never in a hang or a C++ exception."""
    import random
    from rustracer_amd.scenes import cornell_box
    path = os.path.join(str(tmp_path), "c.pbrt")
    text = write_pbrt(cornell_box(16, 16, 4), path).encode()
    rnd = random.Random(5)
    cases = [text.replace(b"[0 1 2 0 2 3]", b"[0 \0\0\0\0 0 2 3]", 1), text[: len(text) // 2], text.replace(b'"', b"", 1), b"\0" * 64, b"Shape " * 1000]
    for _ in range(60):
        b = bytearray(text)
        for _ in range(rnd.choice([1, 2, 8])):
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        cases.append(bytes(b))
    for k, c in enumerate(cases):
        p = os.path.join(str(tmp_path), f"d{k}.pbrt")
        open(p, "wb").write(c)
        try:
            host.PbrtScene(p)
        except host.BackendError:
            pass


# ------------------------------------------------------------------------------------------------ sampled spectra (rc/spectrum.rs, rc/cie.rs)
def _spectrum_tables():
    """The data tables of rtx_spectrum_tables.inl, parsed from the file the C++ host compiles."""
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rustracer_amd", "csrc", "rtx_spectrum_tables.inl")).read()
    t = {m.group(1): np.array([float(x.rstrip("f")) for x in re.findall(r"[-+0-9.e]+f", m.group(2))], F32)
         for m in re.finditer(r"static const float (\w+)\[\d+\] = \{(.*?)\};", text, re.S)}
    t["y_int"] = F32(re.search(r"kCieYIntegral = ([0-9.]+)f", text).group(1))
    return t


def _from_sampled_numpy(t, lam, v):
    """Spectrum::from_sampled (rc/spectrum.rs:108-126) + interpolate_spectrum_samples (:196-211) + from_xyz (:91-96), in f32, operation by operation."""
    lam, v = np.asarray(lam, F32), np.asarray(v, F32)
    xyz = [F32(0), F32(0), F32(0)]
    for i in range(471):
        l = F32(360 + i)
        if l <= lam[0]:
            val = v[0]
        elif l >= lam[-1]:
            val = v[-1]
        else:
            off = min(max(int(np.searchsorted(lam, l, side="right")) - 1, 0), len(lam) - 2)  # find_interval(n, |i| lambda[i] <= l)
            tt = F32(l - lam[off]) / F32(lam[off + 1] - lam[off])
            val = F32(v[off] * F32(F32(1) - tt)) + F32(v[off + 1] * tt)
        for k, tab in enumerate(("kCieX", "kCieY", "kCieZ")):
            xyz[k] = F32(xyz[k] + F32(val * t[tab][i]))
    scale = F32(F32(830) - F32(360)) / F32(t["y_int"] * F32(471))
    x, y, z = (F32(c * scale) for c in xyz)
    m = [(3.240479, -1.537150, -0.498535), (-0.969256, 1.875991, 0.041556), (0.055648, -0.204043, 1.057311)]
    return np.array([F32(F32(F32(F32(a) * x) + F32(F32(b) * y)) + F32(F32(c) * z)) for a, b, c in m], F32)


def test_cie_tables_are_the_1931_standard_observer():
    """Pins the DATA independently of the reference: published properties of the CIE 1931 2-degree colour matching functions."""
    t = _spectrum_tables()
    x, y, z = t["kCieX"], t["kCieY"], t["kCieZ"]
    assert len(x) == len(y) == len(z) == 471
    assert y.max() == 1.0 and 360 + int(y.argmax()) == 555                      # photopic peak V(555 nm) = 1
    assert 360 + int(x.argmax()) == 599 and abs(float(x.max()) - 1.0622) < 1e-3  # x-bar: main lobe 1.0622 at 599 nm ...
    assert 360 + int(x[:140].argmax()) == 442 and abs(float(x[:140].max()) - 0.3501) < 1e-3  # ... secondary lobe 0.3501 at 442 nm
    assert 360 + int(z.argmax()) == 446 and abs(float(z.max()) - 1.7826) < 1e-3  # z-bar: 1.7826 at 446 nm
    assert abs(float(y.astype(np.float64).sum()) - float(t["y_int"])) < 1e-3      # CIE_Y_INTEGRAL = sum of y-bar = 106.8569
    # equal-energy white has x = y = 1/3: the three integrals agree to 5e-4
    sx, sy, sz = (float(a.astype(np.float64).sum()) for a in (x, y, z))
    assert abs(sx / sy - 1) < 5e-4 and abs(sz / sy - 1) < 5e-4
    cw, cn, ck = t["kCopperWavelengths"], t["kCopperN"], t["kCopperK"]
    assert len(cw) == 56 and np.all(np.diff(cw) > 0) and cw[0] < 300 and cw[-1] > 880
    # copper is red because its reflectance edge sits near 560-600 nm: n drops below 0.5 and k rises past it in the red
    red, blue = cw > 620, (cw > 420) & (cw < 480)
    assert cn[red].max() < 0.5 and cn[blue].min() > 1.0 and ck[red].min() > 3.0


def test_from_sampled_matches_a_numpy_restatement_and_physics():
    t = _spectrum_tables()
    eta, k = host.copper()
    assert np.array_equal(eta, F32(sd.COPPER_ETA)) and np.array_equal(k, F32(sd.COPPER_K))  # scene_desc's defaults ARE the reference arithmetic's values
    assert np.array_equal(eta, _from_sampled_numpy(t, t["kCopperWavelengths"], t["kCopperN"]))
    assert np.array_equal(k, _from_sampled_numpy(t, t["kCopperWavelengths"], t["kCopperK"]))
    assert np.allclose(eta, (0.200438, 0.924033, 1.102212), rtol=5e-3) and np.allclose(k, (3.912949, 2.452848, 2.142188), rtol=5e-3)  # the values in the literature
    rng = np.random.default_rng(5)
    lam = np.sort(rng.uniform(300, 900, 40)).astype(F32)
    v = rng.uniform(0, 2, 40).astype(F32)
    assert np.array_equal(host.spectrum_from_sampled(lam, v), _from_sampled_numpy(t, lam, v))
    flat = host.spectrum_from_sampled(F32([400, 700]), F32([1, 1]))   # a constant SPD: Y = (830 - 360) / 471, the Riemann-sum scale of spectrum.rs:119-120
    assert abs(float(0.212671 * flat[0] + 0.715160 * flat[1] + 0.072169 * flat[2]) - 470.0 / 471.0) < 1e-5
    with pytest.raises(host.BackendError, match="increase"):
        host.spectrum_from_sampled(F32([500, 400]), F32([1, 1]))
    # blackbody: Planck locus chromaticities (x, y): 6500 K -> (0.3135, 0.3237), 2856 K (illuminant A) -> (0.4476, 0.4074)
    for temp, (cx, cy) in [(6500.0, (0.3135, 0.3237)), (2856.0, (0.4476, 0.4074))]:
        r, g, b = (float(c) for c in host.spectrum_blackbody(temp, 1.0))
        X = 0.412453 * r + 0.357580 * g + 0.180423 * b; Y = 0.212671 * r + 0.715160 * g + 0.072169 * b; Z = 0.019334 * r + 0.119193 * g + 0.950227 * b
        assert abs(X / (X + Y + Z) - cx) < 2e-3 and abs(Y / (X + Y + Z) - cy) < 2e-3, (temp, X / (X + Y + Z), Y / (X + Y + Z))
    assert np.allclose(host.spectrum_blackbody(5000.0, 3.0), 3.0 * host.spectrum_blackbody(5000.0, 1.0), rtol=1e-6)


def test_metal_default_spectrum_files_and_blackbody_parameters(tmp_path):
    spd = tmp_path / "green.spd"
    spd.write_text("# wavelength value\n400 0.1\n500 0.2 550 0.9\n600 0.2\n700 0.1\n")
    p = _parse(HEADER + 'WorldBegin\nMaterial "metal"\n' + TRI + 'Material "matte" "spectrum Kd" "green.spd"\n' + TRI
               + 'AreaLightSource "diffuse" "blackbody L" [6500 2]\n' + TRI + "WorldEnd\n", base_dir=str(tmp_path))
    m, t = p.table("materials"), p.table("textures")
    metal = [x for x in m if x["kind"] == sd.MAT_METAL][0]
    eta, k = host.copper()
    assert np.array_equal(t[metal["slot"][sd.MAT_SLOTS.index("eta")]]["value"], eta) and np.array_equal(t[metal["slot"][sd.MAT_SLOTS.index("k")]]["value"], k)
    matte_kd = [t[x["slot"][0]]["value"] for x in m if x["kind"] == sd.MAT_MATTE]
    want = host.spectrum_from_sampled(F32([400, 500, 550, 600, 700]), F32([0.1, 0.2, 0.9, 0.2, 0.1]))
    assert any(np.array_equal(v, want) for v in matte_kd) and want[1] > want[0] and want[1] > want[2]
    lights = p.table("lights")
    assert np.array_equal(F32(lights[0]["rgb"]), host.spectrum_blackbody(6500.0, 2.0))
    q = _parse(HEADER + 'WorldBegin\nMaterial "matte" "spectrum Kd" "missing.spd"\n' + TRI + "WorldEnd\n", base_dir=str(tmp_path))
    assert q.n_warnings >= 1 and np.array_equal(q.table("textures")[q.table("materials")[0]["slot"][0]]["value"], F32([0, 0, 0]))  # paramset.rs:257-266: black


def test_instancing_past_the_triangle_budget_is_refused(monkeypatch):
    """ObjectInstance written out on request: a file that instantiates its way past the budget fails with a message naming it (the two-level default copies nothing)."""
    body = 'WorldBegin\nObjectBegin "o"\n' + TRI * 3 + "ObjectEnd\n" + "".join(f'AttributeBegin\nTranslate {i} 0 0\nObjectInstance "o"\nAttributeEnd\n' for i in range(8)) + "WorldEnd\n"
    assert len(host.PbrtScene(text=HEADER + body, flatten_instances=True).table("indices")) == 24
    two_level = _parse(HEADER + body)
    assert len(two_level.table("indices")) == 0 and len(two_level.table("instances")) == 8 and len(two_level.table((0, "indices"))) == 3
    import subprocess, sys, textwrap
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        from rustracer_amd import host
        try:
            host.PbrtScene(text={HEADER + body!r}, flatten_instances=True)
        except host.BackendError as e:
            print("REFUSED", e)
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RTX_INSTANCE_TRIANGLE_BUDGET="10")).stdout
    assert "REFUSED" in out and "exceed 10 triangles" in out


def test_instanced_scene_round_trips_through_a_file(tmp_path):
    """SceneDesc objects / instances -> ObjectBegin .. ObjectEnd / ObjectInstance -> the loader's two-level form: the object-space soups come back array for
    array, each instance with its matrix (the inverse is the loader's own f32 Gauss-Jordan, equal to the description's to rounding), the same top-level tree."""
    from rustracer_amd.scene_desc import SceneDesc
    from rustracer_amd.scenes.procedural import icosphere
    d = SceneDesc()
    a, b = d.matte((0.5, 0.5, 0.5)), d.plastic((0.2, 0.3, 0.6), (0.3, 0.3, 0.3), 0.1)
    d.add_quad((-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5), a)
    P, F = icosphere(1, (0, 0, 0), 0.5)
    n = (P / F32(0.5)).astype(F32)
    o0 = d.add_object([dict(P=P, idx=F, material=b, N=n), dict(P=F32([[0, 1, 0], [1, 1, 0], [0, 1, 1]]), idx=[[0, 1, 2]], material=a, reverse_orientation=True)])
    o1 = d.add_object([dict(P=F32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), idx=[[0, 1, 2]], material=a)])
    for k, obj in enumerate((o0, o1, o0)):
        m = np.eye(4, dtype=F32); m[:3, 3] = (2.0 * k - 2.0, 0.7, 0.5 * k); m[0, 0] = -1.5 if k == 2 else 1.0
        d.add_instance(obj, m)
    d.add_quad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), d.matte((0, 0, 0)), emission=(10, 10, 10))
    path = os.path.join(str(tmp_path), "inst.pbrt")
    write_pbrt(d, path)
    p, h = host.PbrtScene(path=path), host.HostScene(d)
    inst = p.table("instances")
    assert list(inst["object"]) == [0, 1, 0] and np.array_equal(inst["o2w"], np.stack([i.o2w for i in d.instances]))
    assert np.allclose(inst["w2o"], np.stack([i.w2o for i in d.instances]), rtol=0, atol=1e-6)
    for k, ob in enumerate(d.objects):
        for name, want in (("P", ob.P), ("indices", ob.idx), ("tri_flags", ob.flags), ("N", ob.N)):
            got = p.table((k, name))
            assert (len(got) == 0) if want is None else np.array_equal(got, want), (k, name)
    bp, bh = p.bvh(), h.bvh()
    assert all(np.array_equal(bp[k], bh[k]) for k in bp)
    for name in ("P", "indices", "tri_flags", "tri_light"):
        assert np.array_equal(p.table(name), h.table(name)), name


def test_random_scenes_round_trip_through_a_file(tmp_path, monkeypatch):
    """scripts/fuzz_pbrt.py on a few scenes of each generator (rooms over every material / texture / light class; two-level scenes whose objects hold quadrics and masked
    meshes): written out, loaded, equal table by table to the scene handed over call by call. It found the exporter leaving an object's quadrics and masks out and defining
    a named material inside the ObjectBegin block that used it first - popped with the block, "no such named material, using matte" for every later user."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_pbrt.py"), "14", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 different" in r.stdout and "different" in r.stdout.split("objects:")[1] and " 14 of 14 scenes equal" in r.stdout.split("objects:")[1], r.stdout

