"""CPU-side tests (-m "not gpu"): oracle vs golden fixtures, product host layer vs oracle, C-ABI exports."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import bits, random_rays

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def cornell():
    from rustracer_amd.scenes import cornell_box
    return cornell_box(32, 32, 16)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "cornell_32x32_16spp.npz"))


# ---------------------------------------------------------------- oracle vs committed golden vectors
def test_oracle_reproduces_golden(orc, cornell, gold):
    o = orc.OracleScene(cornell)
    b = o.bvh()
    for k in ("bounds", "offset", "n_prims", "axis", "ordered"):
        assert np.array_equal(b[k], gold["bvh_" + k]), k
    hc = o.trace(gold["rays"])
    assert np.array_equal(hc["prim"], gold["hit_prim"]) and np.array_equal(bits(hc["t"]), bits(gold["hit_t"]))
    assert np.array_equal(bits(hc["b0"]), bits(gold["hit_b0"])) and np.array_equal(bits(hc["b1"]), bits(gold["hit_b1"]))
    assert (hc["nodes"], hc["tris"]) == (int(gold["hit_nodes"]), int(gold["hit_tris"]))
    ha = o.trace(gold["rays_any"], any_hit=True)
    assert np.array_equal(ha["occluded"], gold["occluded"])
    film, st = o.render(mode=1, n_threads=3)  # keyed mode is thread-count independent
    assert np.array_equal(bits(film), bits(gold["film_xyzw"]))
    assert [st[k] for k in ("camera_rays", "rays_closest", "rays_shadow", "rays_mis")] == gold["stats"].tolist()


def test_oracle_sampler_tables_golden(orc):
    g = np.load(os.path.join(GOLD, "sampler_tables_keyed.npz"))
    for spp in (16, 64):
        for px in (0, 1, 777):
            t1, t2, _ = orc.sampler_tables(spp, 4, 1, px)
            assert np.array_equal(bits(t1), bits(g[f"t1_{spp}_{px}"])) and np.array_equal(bits(t2), bits(g[f"t2_{spp}_{px}"]))


def test_oracle_ref_and_keyed_modes_agree_statistically(orc):
    from rustracer_amd.scenes import cornell_box
    o = orc.OracleScene(cornell_box(48, 48, 64))
    ref, _ = o.render(mode=0)
    key, _ = o.render(mode=1)
    a, b = orc.film_to_rgb(ref), orc.film_to_rgb(key)
    # two unbiased estimators of the same image: means agree, per-pixel difference is Monte-Carlo noise
    assert abs(a.mean() - b.mean()) / b.mean() < 0.02
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 0.25


def test_c1_config_on_the_cpu_port(orc):
    """BASELINE.json configs[0] exactly - cornell-box 400x400, PathIntegrator maxdepth 5, 64 spp, on the CPU (the reference's own runnable case):
    the oracle in the reference's tile-sequential sampler mode on every host thread, and the pixel-keyed mode the GPU parity is defined on. Both are
    unbiased estimators of the same image; the frame is independent of the thread count because a tile's stream is seeded by its index."""
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(400, 400, 64)
    assert (d.integrator.max_depth, d.sampler.spp, d.film.xres, d.film.yres) == (5, 64, 400, 400)
    o = orc.OracleScene(d)
    ref, st = o.render(mode=0)
    # 64 unit weights per pixel, except where a sample fell exactly on a pixel edge and went to both neighbours (film.rs:313-321)
    assert st["camera_rays"] == 400 * 400 * 64 and ref[..., 3].sum() >= 400 * 400 * 64 and (np.abs(ref[..., 3] - 64) <= 4).all() and (ref[..., 3] == 64).mean() > 0.99
    ref1, _ = o.render(mode=0, n_threads=3)
    assert np.array_equal(ref[..., 3], ref1[..., 3]) and np.allclose(ref, ref1, rtol=1e-5)  # only such edge splats cross tiles: the merge order barely matters
    key, _ = o.render(mode=1)
    a, b = orc.film_to_rgb(ref), orc.film_to_rgb(key)
    assert abs(a.mean() - b.mean()) / b.mean() < 0.01
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 0.2


def test_oracle_single_path_probe_matches_film(orc):
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(8, 8, 4)
    o = orc.OracleScene(d)
    film, _ = o.render(mode=1, n_threads=1)
    rgb = orc.film_to_rgb(film)
    px, py = 3, 5
    acc = np.zeros(3, np.float32)
    for s in range(4):
        acc = acc + o.li_keyed(px, py, s)
    assert np.allclose(acc / np.float32(4), rgb[py, px], rtol=2e-5, atol=1e-7)


# ---------------------------------------------------------------- product host layer vs oracle (bit-exact)
def _scenes():
    from rustracer_amd.scenes import cornell_box, random_soup
    return [cornell_box(40, 24, 4), random_soup(3000, seed=5), random_soup(257, seed=9, max_prims=1), random_soup(64, seed=2, degenerate=True)]


@pytest.mark.parametrize("idx", range(4))
def test_host_bvh_matches_oracle(orc, host, idx):
    d = _scenes()[idx]
    bo, bh = orc.OracleScene(d).bvh(), host.HostScene(d).bvh()
    assert len(bo["bounds"]) == len(bh["bounds"])
    for k in ("offset", "n_prims", "axis", "ordered"):
        assert np.array_equal(bo[k], bh[k]), k
    assert np.array_equal(bits(bo["bounds"]), bits(bh["bounds"]))
    # structural sanity of the flattened tree (rc/bvh/mod.rs:314-358)
    n = len(bh["bounds"])
    leaves = bh["n_prims"] > 0
    assert sorted(bh["ordered"].tolist()) == list(range(d.n_tris))
    assert int(bh["n_prims"][leaves].sum()) == d.n_tris
    assert np.all(bh["offset"][~leaves] > np.nonzero(~leaves)[0]) and np.all(bh["offset"][~leaves] < n)


def test_host_camera_film_setup_matches_oracle(orc, host):
    from rustracer_amd.scenes import cornell_box
    from rustracer_amd import scene_desc as sd
    for filt in ((sd.FILTER_BOX, (0.5, 0.5, 0, 0)), (sd.FILTER_GAUSSIAN, (2.0, 2.0, 2.0, 0)), (sd.FILTER_MITCHELL, (2.0, 2.0, 1 / 3, 1 / 3)), (sd.FILTER_TRIANGLE, (2.0, 1.5, 0, 0))):
        d = cornell_box(50, 30, 4)
        d.film.filter_kind, d.film.filter_params = filt
        d.film.crop = (0.1, 0.9, 0.2, 1.0)
        d.camera.lens_radius = 0.5
        so, sh = orc.OracleScene(d).setup(), host.HostScene(d).setup()
        for k in ("raster_to_camera", "dx_camera", "dy_camera", "filter_table"):
            assert np.array_equal(bits(so[k]), bits(sh[k])), (filt, k)
        assert np.array_equal(so["sample_bounds"], sh["sample_bounds"]) and np.array_equal(so["cropped"], sh["cropped"])
    m_o, mi_o = orc.look_at((1, 2, 3), (4, 0, 9), (0, 1, 0))
    m_h, mi_h = host.look_at((1, 2, 3), (4, 0, 9), (0, 1, 0))
    assert np.array_equal(bits(m_o), bits(m_h)) and np.array_equal(bits(mi_o), bits(mi_h))


def test_host_mip_pyramid(host):
    from rustracer_amd.scene_desc import SceneDesc
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (8, 16, 3)).astype(np.float32)
    s = SceneDesc()
    m = s.add_mip(img)
    s.add_quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), s.matte(s.image_tex(m)))
    lv = host.HostScene(s).mip_levels(0)
    assert [a.shape[:2] for a in lv] == [(8, 16), (4, 8), (2, 4), (1, 2), (1, 1)]  # 1 + log2(16) levels (rc/mipmap.rs:159)
    assert np.array_equal(lv[0], img)
    e = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) * np.float32(0.25)  # :176-180
    assert np.array_equal(bits(lv[1]), bits(e.astype(np.float32)))


# ---------------------------------------------------------------- C ABI
def test_c_abi_exports_every_declared_symbol(host):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for header, lib in (("rtx_hip.h", host.hip_lib()), ("rtx_host.h", host.lib())):
        text = open(os.path.join(root, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = set(re.findall(r"\b(rtx?h?_[a-z0-9_]+)\s*\(", text))
        assert len(names) >= 10
        for n in names:
            assert hasattr(lib, n), f"{header}: {n} not exported"
    assert b"gfx950" in host.hip_lib().rt_version()


def test_product_fails_loudly_without_gpu(host, cornell):
    if host.device_available():
        pytest.skip("a GPU is present")
    h = host.HostScene(cornell)
    with pytest.raises(host.BackendError, match="no CPU fallback"):
        h.render()
    with pytest.raises(host.BackendError, match="no CPU fallback"):
        h.trace(random_rays(4, [0, 0, 0], [1, 1, 1], 0))
    with pytest.raises(host.BackendError):
        host.sampler_tables(16, 4, 0, 4)


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dirpath, _, files in os.walk(os.path.join(root, "rustracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|#include\s+\"[^\"]*orc_|liborc", t):
                    bad.append(f)
    assert not bad, bad


@pytest.mark.parametrize("shape,wrap", [((3, 5), 0), ((17, 33), 1), ((60, 100), 2), ((16, 12), 0), ((7, 8), 2)])
def test_mip_of_a_non_power_of_two_image_is_resampled_like_the_reference(orc, host, shape, wrap):
    """MIPMap::new's 4-tap Lanczos zoom (rc/mipmap.rs:75-139, 362-408): host == oracle bit for bit, sizes = next powers of two."""
    from rustracer_amd.scene_desc import SceneDesc
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    s = SceneDesc()
    m = s.add_mip(img, wrap=wrap)
    s.add_quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), s.matte(s.image_tex(m)))
    lo, lh = orc.OracleScene(s).mip_levels(0), host.HostScene(s).mip_levels(0)
    p2 = lambda v: 1 << (int(v) - 1).bit_length()
    assert lo[0].shape[:2] == (p2(shape[0]), p2(shape[1])) and len(lo) == len(lh)
    for a, b in zip(lo, lh):
        assert a.shape == b.shape and np.array_equal(bits(a), bits(b))
    assert (lo[0] >= 0).all() and np.isfinite(lo[0]).all()  # clamped to [0, inf) after the t pass (:133)
    # a constant image stays constant where the 4 taps see it whole (weights are normalised, :376-378)
    c = SceneDesc(); c.add_mip(np.full(shape + (3,), 0.5, np.float32), wrap=0 if wrap == 1 else wrap)
    c.add_quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), c.matte(0.5))
    assert np.allclose(host.HostScene(c).mip_levels(0)[0], 0.5, atol=1e-6)


def test_round_up_pow2_reference_kat(orc):
    assert orc.round_up_pow2(1023) == 1024 and orc.round_up_pow2(1024) == 1024  # rc/lib.rs:341-345
    assert [orc.round_up_pow2(v) for v in (1, 2, 3, 5, 17, 1025)] == [1, 2, 4, 8, 32, 2048]


def test_png8_quantisation_matches_write_image_png(host):
    # rc/imageio.rs:59-62 with rc/spectrum.rs:387-393, value by value in Python floats rounded to f32 at each step
    v = np.array([-1.0, 0.0, 0.001, 0.0031308, 0.0031309, 0.18, 0.5, 1.0, 1.5, np.nan, np.inf], np.float32)
    want = []
    for x in v:
        if np.isnan(x):
            want.append(0); continue
        g = np.float32(12.92) * x if x <= np.float32(0.0031308) else np.float32(1.055) * np.float32(np.power(np.float32(x), np.float32(1.0) / np.float32(2.4))) - np.float32(0.055)
        want.append(int(min(max(np.float32(255.0) * np.float32(g) + np.float32(0.5), 0.0), 255.0)))
    assert host.rgb_to_png8(v).tolist() == want
    assert want[:2] == [0, 0] and want[7:9] == [255, 255] and want[5] == 118


@pytest.mark.parametrize("camera", [dict(frame_aspect=2.0), dict(frame_aspect=0.5), dict(screen_window=(-0.5, 1.25, -0.75, 0.5)), dict(frame_aspect=3.0, screen_window=(-1.0, 1.0, -1.0, 1.0))])
def test_screen_window_and_frame_aspect_match_the_oracle(host, orc, camera):
    """PerspectiveCamera::create's "frameaspectratio" / "screenwindow" (camera.rs:86-107): same raster-to-camera matrix and differentials from both hosts."""
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(48, 32, 4)
    for k, v in camera.items():
        setattr(d.camera, k, v)
    a, b = host.HostScene(d).setup(), orc.OracleScene(d).setup()
    for k in ("raster_to_camera", "dx_camera", "dy_camera"):
        assert np.array_equal(a[k], b[k]), k
    base = host.HostScene(cornell_box(48, 32, 4)).setup()
    assert not np.array_equal(a["raster_to_camera"], base["raster_to_camera"]) or camera == dict(frame_aspect=1.5)


def test_kernel_register_and_scratch_budgets(host):
    """Occupancy is decided by registers on this chip (waves per SIMD = 512 / allocated VGPRs, 8-register granules; MI355X guide) and a change
    in one device function can move a kernel across a step without any test failing: alpha masks once took k_shade<5> from 218 to 258 VGPRs (2 -> 1
    wave per SIMD) because their texture evaluator became reachable from every shade kernel. The budgets below are read from the BUILT library."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_budget", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "kernel_budget.py"))
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    res = kb.kernel_resources(host.HIP_LIB)
    # (round 5: k_shade has a sixth template argument, LDSREC; the forms with LDSREC = 0 keep their names below)
    res = {(re.sub(r", 0>$", ">", k) if k.startswith(("rtx::k_shade<", "rtx::k_trace<")) else k): v for k, v in res.items()}  # (trailing defaults: LDSREC = 0, MID = 0)
    budget = {  # kernel prefix -> (max VGPRs, max scratch bytes)
        "rtx::k_shade<1, false, false, false, false>": (128, 32),       # FOUR waves per SIMD (round 4), no out-of-line call; round 5: the two registers of a voxel's 32-byte distribution record cost five spilled dwords
        # the textured front-ends bound to three waves (round 4): a few dozen spilled dwords buy the third wave (S4 shade 3027 -> 2801 ms)
        "rtx::k_shade<3, false, false, false, false>": (168, 64),
        "rtx::k_shade<5, false, false, false, false>": (168, 128),
        "rtx::k_shade<6, false, false, false, false>": (168, 224),
        "rtx::k_shade<0, false, false, false, false>": (256, 2048),     # 257 (one accumulation register added by a callee) is ONE wave per SIMD
        "rtx::k_trace<false, false, 256, 16, 0>": (72, 0),    # the LDS-resident closest-hit kernel of the headline: 7 waves
        "rtx::k_trace<true, false, 256, 16, 0>": (64, 0),     # ... and its shadow-ray twin: 8 waves
        "rtx::k_trace<true, false, 1024, 16, 0, 1>": (64, 0), # a mid-size scene's occlusion rays: one 1024-lane workgroup per CU around 157 KB of LDS
        "rtx::k_trace<false, false, 1024, 16, 0, 1>": (72, 0),  # ... and its closest-hit rays (bounds + eight link rows in LDS, triangles from HBM)
        "rtx::k_trace_pair<false, 128, 32, 0>": (80, 0),     # HBM scenes: 6 waves; no scratch (an indexed load per node visit once hid here)
        "rtx::k_trace_quad<128, 32, 0>": (72, 0),   # (its 16 KB stack column per 128 lanes holds it to five waves per SIMD; 66 registers with the two node tests)
        "rtx::k_trace_top<false, 512, 0>": (80, 0), "rtx::k_trace_top<true, 512, 0>": (80, 0),   # 512 lanes per workgroup: 6 waves
        # GENERAL instantiations (quadrics, masked triangles, object instances): the out-of-line quadric / mask evaluators cost them the waves (2 per SIMD),
        # the plain instantiations above must not notice that these exist
        "rtx::k_trace_pair<false, 128, 32, 1>": (192, 128), "rtx::k_trace_quad<128, 32, 1>": (192, 128),
        # ... and without the mask evaluator (scenes whose meshes carry no alpha mask): three waves
        "rtx::k_trace_pair<false, 128, 32, 2>": (168, 32), "rtx::k_trace_quad<128, 32, 2>": (128, 32), "rtx::k_trace<false, false, 256, 16, 2>": (128, 32),
        # ... and with neither masks nor quadrics (instances over plain triangles): four waves
        "rtx::k_trace_pair<false, 128, 32, 3>": (128, 32), "rtx::k_trace_quad<128, 32, 3>": (128, 0),  # (round 6, stream pointers with the non-temporal hint: five spilled dwords in the nested form)
        "rtx::k_shade<3, true, false, false, false>": (256, 512), "rtx::k_shade<5, true, false, false, false>": (256, 512), "rtx::k_shade<6, true, false, false, false>": (256, 512),   # (they spill some: still one wave more than the generic kernel's code)
        # LEAN forms (area lights, constant textures): no out-of-line evaluator, three waves, a few spilled dwords at most
        "rtx::k_shade<3, false, true, false, false>": (168, 0), "rtx::k_shade<5, false, true, false, false>": (168, 32), "rtx::k_shade<6, false, true, false, false>": (168, 64),
        "rtx::k_shade<3, false, false, true, false>": (168, 32),   # BOUNCED: the Lambert front-end past the camera vertices, everything inline, three waves
        # QLIGHTS (round 4): the LEAN forms with inline cone sampling of sphere lights - three waves like the LEAN forms
        "rtx::k_shade<3, false, true, false, true>": (168, 0), "rtx::k_shade<5, false, true, false, true>": (168, 32), "rtx::k_shade<6, false, true, false, true>": (168, 64),
        # the two-level walk as one loop (round 4): four waves for closest hit, six for occlusion rays, nothing spilled
        "rtx::k_trace_inst<false, 128, 32>": (128, 0), "rtx::k_trace_inst<true, 128, 32>": (80, 0),
        "rtx::k_resolve<false>": (88, 256), "rtx::k_raygen": (72, 0), "rtx::k_film_accumulate": (96, 0),  # (round 6: eight samples' records in flight per lane)
    }
    for name, (vg, sc) in budget.items():
        r = res[name]
        spills_allowed = 32 if name.startswith("rtx::k_shade<0") else (80 if name.startswith("rtx::k_shade<") and name[13] in "356" and name.endswith("false, false, false, false>") else 64 if name.startswith("rtx::k_shade<") and (name.endswith("true, false, false, false>") or name.endswith("false, true, false, false>") or name.endswith("false, false, true>")) else ((8 if name.startswith("rtx::k_trace<") else 2) if name.endswith(", 2>") else (8 if name.startswith("rtx::k_shade<1,") or name.endswith(", 3>") else 0)))  # the generic kernel fills its 256 registers: a few spilled values, never a second wave lost
        assert r["vgpr"] <= vg and r["scratch"] <= sc and r["vgpr_spills"] <= spills_allowed, (name, r)
    # round 5: the forms that keep the scene's small tables in LDS (LDSREC: 1 everything, 2 materials + textures, 3 lights + materials + textures + image headers): the register
    # bounds of the forms they replace, a handful of spilled dwords in the four-wave kernels, and LDS that leaves the waves the bound promises (4 x 33 KB, 3 x 10 KB)
    lds_budget = {  # name -> (max VGPRs, max scratch bytes, max spilled dwords, max LDS bytes)
        "rtx::k_shade<1, false, false, false, false, 1>": (128, 32, 8, 36864), "rtx::k_shade<1, false, false, false, false, 3>": (128, 32, 8, 16384),
        "rtx::k_shade<3, false, false, false, false, 3>": (168, 64, 72, 16384), "rtx::k_shade<3, false, false, true, false, 3>": (168, 32, 64, 16384),
        "rtx::k_shade<5, false, false, false, false, 3>": (168, 160, 72, 16384), "rtx::k_shade<6, false, false, false, false, 3>": (168, 224, 72, 16384),
        "rtx::k_shade<3, false, true, false, false, 2>": (168, 0, 0, 6144), "rtx::k_shade<5, false, true, false, false, 2>": (168, 32, 64, 6144), "rtx::k_shade<6, false, true, false, false, 2>": (168, 64, 64, 6144),
        "rtx::k_shade<3, false, true, false, true, 1>": (168, 0, 0, 32768), "rtx::k_shade<5, false, true, false, true, 1>": (168, 32, 64, 32768), "rtx::k_shade<6, false, true, false, true, 1>": (168, 64, 64, 32768),
    }
    for name, (vg, sc, sp, lds) in lds_budget.items():
        r = res[name]
        assert r["vgpr"] <= vg and r["scratch"] <= sc and r["vgpr_spills"] <= sp and r["lds"] <= lds, (name, r)


def test_build_is_decided_by_a_content_stamp_not_by_file_times(host, monkeypatch, tmp_path):
    """host.build(): the libraries travel to the GPU box by snapshot, where mtimes mean nothing - whether they are current is the sha256 of
    the sources stamped beside them. A matching stamp: no make; a stamp of other sources: make -B and a new stamp; touching a file changes nothing."""
    import subprocess
    host.build()
    assert open(host.STAMP).read().strip() == host.source_sha()
    calls = []
    monkeypatch.setattr(subprocess, "check_call", lambda cmd, **kw: calls.append(cmd))
    os.utime(os.path.join(os.path.dirname(host.HIP_LIB), "..", "rtx_kernels.h"))  # newer than the libraries: irrelevant
    host.build()
    assert calls == []
    stamp = open(host.STAMP).read()
    try:
        open(host.STAMP, "w").write("0" * 64 + "\n")
        host.build()
        assert len(calls) == 1 and calls[0][0] == "make" and "-B" in calls[0]
        assert open(host.STAMP).read().strip() == host.source_sha()
    finally:
        open(host.STAMP, "w").write(stamp)


def test_second_sobol_matrix_is_a_taylor_shift_over_gf2():
    """The device evaluates the (0,2)-sequence's second dimension without the generator-matrix loop (rtx_dev_math.h sobol1_bits): the XOR of the matrix columns
    over the set bits of the Gray code equals the bit reversal of a five-step Taylor shift. Checked against the loop (`v ^= v >> 1`,
    rc/sampler/lowdiscrepancy.rs:104-112) for every index a 65536-spp table can hold and for random 32-bit arguments."""
    def loop(n):  # sobol_2d's inner loop on the index itself (no Gray code): v starts at 1 << 31
        v, out = np.uint32(1 << 31), np.zeros_like(n)
        n = n.copy()
        for _ in range(32):
            out ^= np.where(n & np.uint32(1), v, np.uint32(0))
            n >>= np.uint32(1)
            v ^= v >> np.uint32(1)
        return out

    def brev(x):
        x = ((x >> np.uint32(1)) & np.uint32(0x55555555)) | ((x & np.uint32(0x55555555)) << np.uint32(1))
        x = ((x >> np.uint32(2)) & np.uint32(0x33333333)) | ((x & np.uint32(0x33333333)) << np.uint32(2))
        x = ((x >> np.uint32(4)) & np.uint32(0x0f0f0f0f)) | ((x & np.uint32(0x0f0f0f0f)) << np.uint32(4))
        x = ((x >> np.uint32(8)) & np.uint32(0x00ff00ff)) | ((x & np.uint32(0x00ff00ff)) << np.uint32(8))
        return (x >> np.uint32(16)) | (x << np.uint32(16))

    def shift(n):
        n = n.copy()
        for s, m in ((16, 0xffff0000), (8, 0xff00ff00), (4, 0xf0f0f0f0), (2, 0xcccccccc), (1, 0xaaaaaaaa)):
            n ^= (n & np.uint32(m)) >> np.uint32(s)
        return brev(n)

    g = np.concatenate([np.arange(1 << 17, dtype=np.uint32), np.random.default_rng(7).integers(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32)])
    assert np.array_equal(loop(g), shift(g))
