"""One material / texture / light kind at a time on the GPU (generic k_shade<0> path) against the oracle.

A small closed room with a smooth-shaded sphere (per-vertex normals and uvs), a tilted two-sided emitter and one
extra light of the kind under test. Same gates as the other parity suites: film weights exact, linear-RGB film
within 1e-3 relative L2, ray counts within a few flipped paths.
"""
import numpy as np
import pytest

from util import rel_l2

pytestmark = pytest.mark.gpu
L2_GATE = 1e-3


def _zoo(material: str, light: str = "area", res=(40, 32), spp=8, max_depth=5):
    from rustracer_amd.scene_desc import SceneDesc, WRAP_BLACK, WRAP_CLAMP, WRAP_REPEAT
    from rustracer_amd.scenes.procedural import checker_fbm_image, icosphere, sky_image
    s = SceneDesc()
    img = s.add_mip(checker_fbm_image(32, 5, (0.9, 0.3, 0.2), (0.2, 0.3, 0.9), 4), trilinear=False, max_aniso=8.0, wrap=WRAP_REPEAT)
    img_tri = s.add_mip(checker_fbm_image(16, 6), trilinear=True, wrap=WRAP_CLAMP)
    img_blk = s.add_mip(checker_fbm_image(16, 7), trilinear=False, max_aniso=2.0, wrap=WRAP_BLACK)
    img_npot = s.add_mip(checker_fbm_image(32, 8)[:20, :24], trilinear=False, max_aniso=8.0, wrap=WRAP_REPEAT)  # 24 x 20: Lanczos-zoomed to 32 x 32 (mipmap.rs:75-139)
    t_img, t_tri, t_blk = s.image_tex(img, 3, 2, 0.1, 0.2), s.image_tex(img_tri, 2, 2), s.image_tex(img_blk, 1.5, 1.5, -0.2, 0.0)
    t_npot = s.image_tex(img_npot, 2, 3)
    mats = {
        "matte": lambda: s.matte((0.6, 0.5, 0.4)),
        "oren_nayar": lambda: s.matte((0.6, 0.5, 0.4), sigma=30.0),
        "matte_image_ewa": lambda: s.matte(t_img),
        "matte_image_trilinear_clamp": lambda: s.matte(t_tri),
        "matte_image_black_wrap": lambda: s.matte(t_blk),
        "matte_image_npot": lambda: s.matte(t_npot),
        "matte_checker_closedform": lambda: s.matte(s.checker_tex(t_img, (0.1, 0.1, 0.6), 6, 5, 0.1, 0.3)),
        "matte_checker_none_nested": lambda: s.matte(s.scale_tex(s.checker_tex((0.9, 0.8, 0.2), t_tri, 7, 7, aa="none"), s.mix_tex(s.uv_tex(2, 2), s.const_tex((0.8, 0.8, 0.8)), s.const_tex(0.6)))),
        "matte_uv": lambda: s.matte(s.uv_tex(3.0, 2.0, 0.2, 0.1)),
        "matte_fbm": lambda: s.matte(s.scale_tex(s.fbm_tex(0.6, 6), s.const_tex((0.9, 0.7, 0.5)))),
        "plastic_fbm_checker_roughness": lambda: s.plastic(s.checker_tex(s.fbm_tex(0.5, 4), (0.5, 0.2, 0.2), 3, 3), (0.4, 0.4, 0.4), 0.2),
        "matte_bump_fbm": lambda: s.set_bump(s.matte((0.6, 0.5, 0.4)), s.scale_tex(s.fbm_tex(0.6, 5), s.const_tex(0.05))),
        "plastic_bump_image": lambda: s.set_bump(s.plastic((0.3, 0.3, 0.5), (0.4, 0.4, 0.4), 0.1), s.scale_tex(t_img, s.const_tex(0.02))),
        "mirror_bump_checker": lambda: s.set_bump(s.mirror(0.9), s.checker_tex(0.0, 0.01, 8, 8)),
        "mix_bump_both": lambda: s.mix(s.set_bump(s.matte((0.7, 0.2, 0.2)), s.scale_tex(s.fbm_tex(0.5, 4), s.const_tex(0.03))),
                                       s.set_bump(s.metal(roughness=0.1), s.scale_tex(s.uv_tex(9, 9), s.const_tex(0.02))), 0.4),
        "disney_default": lambda: s.disney((0.6, 0.3, 0.2)),
        "disney_metal_aniso_clearcoat": lambda: s.disney((0.8, 0.6, 0.2), metallic=0.9, roughness=0.3, anisotropic=0.7, speculartint=0.5, clearcoat=0.8, clearcoatgloss=0.7),
        "disney_sheen_textured": lambda: s.disney(t_img, roughness=0.6, sheen=1.0, sheentint=0.8, metallic=0.1),
        "disney_spectrans": lambda: s.disney((0.7, 0.8, 0.9), spectrans=0.8, roughness=0.15, eta=1.4),
        "disney_thin": lambda: s.disney((0.5, 0.7, 0.4), thin=True, flatness=0.4, difftrans=1.2, spectrans=0.3, roughness=0.4),
        "disney_scatterdistance": lambda: s.disney((0.6, 0.5, 0.5), scatterdistance=(0.1, 0.1, 0.1), sheen=0.5),
        "mix_disney_bump": lambda: s.mix(s.set_bump(s.disney((0.3, 0.5, 0.7), clearcoat=1.0), s.scale_tex(s.fbm_tex(0.5, 4), s.const_tex(0.02))), s.glass(), (0.6, 0.6, 0.6)),
        "matte_scale_mix_tex": lambda: s.matte(s.mix_tex(s.scale_tex(t_img, s.const_tex((0.9, 0.8, 0.7))), s.const_tex((0.1, 0.6, 0.2)), s.const_tex(0.3))),
        "plastic": lambda: s.plastic((0.3, 0.1, 0.1), (0.5, 0.5, 0.5), 0.15),
        "plastic_noremap": lambda: s.plastic(t_img, (0.4, 0.4, 0.4), 0.2, remap=False),
        "metal": lambda: s.metal(roughness=0.05),
        "metal_aniso": lambda: s.metal(roughness=0.1, urough=0.02, vrough=0.3),
        "mirror": lambda: s.mirror(0.9),
        "glass": lambda: s.glass(index=1.5),
        "glass_rough": lambda: s.glass(kr=0.9, kt=0.8, index=1.33, urough=0.1, vrough=0.2),
        "uber": lambda: s.uber(kd=(0.3, 0.4, 0.2), ks=(0.3, 0.3, 0.3), kr=(0.1, 0.1, 0.1), kt=(0.2, 0.2, 0.2), roughness=0.1, opacity=(0.8, 0.7, 0.9)),
        "uber_opaque": lambda: s.uber(kd=t_img, ks=(0.25, 0.25, 0.25), roughness=0.2),  # opacity 1, Kr = Kt = 0: Lambert + microfacet reflection only
        "substrate": lambda: s.substrate(kd=(0.5, 0.2, 0.2), ks=(0.3, 0.3, 0.3), urough=0.05, vrough=0.2),
        "translucent": lambda: s.translucent(kd=(0.4, 0.4, 0.3), ks=(0.2, 0.2, 0.2), reflect=0.4, transmit=0.6, roughness=0.2),
        "mix": lambda: s.mix(s.plastic((0.1, 0.3, 0.1), (0.4, 0.4, 0.4), 0.1), s.metal(roughness=0.1), 0.35),
        "mix_nested": lambda: s.mix(s.mix(s.matte((0.7, 0.1, 0.1)), s.mirror(0.8), (0.6, 0.5, 0.4)), s.substrate(), t_tri),
    }
    wall, wall2 = s.matte((0.7, 0.7, 0.7)), s.matte((0.2, 0.5, 0.2), sigma=10.0)
    m = mats[material]()
    # room 4 x 3 x 4, open towards the camera when an environment light is present
    s.add_quad((-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2), wall)
    s.add_quad((-2, 0, 2), (-2, 3, 2), (2, 3, 2), (2, 0, 2), wall)
    s.add_quad((-2, 0, -2), (-2, 3, -2), (-2, 3, 2), (-2, 0, 2), wall2)
    s.add_quad((2, 0, -2), (2, 0, 2), (2, 3, 2), (2, 3, -2), wall)
    if light != "infinite":
        s.add_quad((-2, 3, -2), (2, 3, -2), (2, 3, 2), (-2, 3, 2), wall)
    P, F = icosphere(2, (0.2, 0.9, 0.3), 0.8)
    n = (P - np.float32((0.2, 0.9, 0.3))) / np.float32(0.8)
    uv = np.stack([np.arctan2(n[:, 2], n[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(n[:, 1], -1, 1)) / np.pi], -1)
    s.add_mesh(P, F, m, N=n, UV=uv)
    s.add_quad((-1.2, 0.01, -1.0), (-1.2, 1.2, -0.4), (-0.4, 1.2, -0.4), (-0.4, 0.01, -1.0), m, UV=[(0, 0), (0, 1), (1, 1), (1, 0)])  # a flat patch of the material, with uvs
    s.add_quad((-0.5, 2.9, -0.5), (0.5, 2.95, -0.5), (0.5, 2.95, 0.5), (-0.5, 2.9, 0.5), s.matte((0.0, 0.0, 0.0)), emission=(12.0, 11.0, 9.0), two_sided=(light == "area_two_sided"))
    if light == "point":
        s.point_light((1.2, 2.2, -1.0), (3.0, 3.0, 4.0))
    elif light == "distant":
        s.distant_light((0.0, 0.0, 0.0), (-0.3, -1.0, 0.4), (0.8, 0.7, 0.6))
    elif light == "infinite":
        env = s.add_mip(sky_image(64, 32, (0.2, -0.5, 0.8), 40.0, 0.97), trilinear=False, max_aniso=0.0)
        s.infinite_light(env, np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32))
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 1.5, -5.5), (0.0, 1.2, 0.0), 42.0
    s.film.xres, s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = max_depth
    return s


def _check(gpu_host, orc, d):
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render(count_traversal=True)
    assert np.array_equal(fo[..., 3], fh[..., 3])
    ro, rh = orc.film_to_rgb(fo), gpu_host.film_to_rgb(fh)
    assert np.isfinite(rh).all()
    err = rel_l2(rh, ro)
    assert err < L2_GATE, err
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])
    assert int(sh["paths_scrubbed"]) == int(so["scrubbed"]) if "scrubbed" in so else True


MATERIALS = ["matte", "oren_nayar", "matte_image_ewa", "matte_image_trilinear_clamp", "matte_image_black_wrap", "matte_image_npot", "matte_checker_closedform", "matte_checker_none_nested", "matte_uv", "matte_fbm",
             "plastic_fbm_checker_roughness", "matte_bump_fbm", "plastic_bump_image", "mirror_bump_checker", "mix_bump_both", "disney_default", "disney_metal_aniso_clearcoat", "disney_sheen_textured", "disney_spectrans",
             "disney_thin", "disney_scatterdistance", "mix_disney_bump", "matte_scale_mix_tex", "plastic",
             "plastic_noremap", "metal", "metal_aniso", "mirror", "glass", "glass_rough", "uber", "uber_opaque", "substrate", "translucent", "mix", "mix_nested"]


@pytest.mark.parametrize("material", MATERIALS)
def test_material_matches_oracle(gpu_host, orc, material):
    _check(gpu_host, orc, _zoo(material))


@pytest.mark.parametrize("light", ["area_two_sided", "point", "distant", "infinite"])
@pytest.mark.parametrize("material", ["plastic", "glass", "matte_image_ewa"])
def test_light_kinds_match_oracle(gpu_host, orc, material, light):
    _check(gpu_host, orc, _zoo(material, light))


def test_uniform_light_strategy_and_deep_paths(gpu_host, orc):
    d = _zoo("uber", "point", spp=4, max_depth=12)
    d.integrator.light_strategy = "uniform"
    d.integrator.rr_threshold = 0.5
    _check(gpu_host, orc, d)


@pytest.mark.parametrize("lights", ["point", "distant", "infinite", "all"])
def test_constant_matte_scene_under_other_light_kinds(gpu_host, orc, lights):
    """Constant-Kd matte materials with non-area lights: the register-resident front-end with the generic light code (k_shade<3>)."""
    from rustracer_amd.scenes import cornell_box
    from rustracer_amd.scenes.procedural import sky_image
    d = cornell_box(72, 56, 16)
    if lights in ("point", "all"):
        d.point_light((278.0, 400.0, 200.0), (4e4, 3e4, 2e4))
    if lights in ("distant", "all"):
        d.distant_light((0.0, 0.0, 0.0), (0.1, 0.3, 1.0), (0.5, 0.6, 0.8))
    if lights in ("infinite", "all"):
        env = d.add_mip(sky_image(64, 32, (0.0, -0.3, -1.0), 30.0, 0.97), trilinear=False, max_aniso=0.0)
        d.infinite_light(env, np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32))
    _check(gpu_host, orc, d)


@pytest.mark.parametrize("scene", ["cornell", "mis-spheres", "zoo-plastic-infinite", "zoo-matte-image", "mis-plates-small"])
def test_tables_in_lds_change_no_bit_of_the_film(gpu_host, scene, monkeypatch):
    """Round 5: the shade kernels read the small tables of a scene - triangle records, lights, materials, textures, image headers (k_shade's LDSREC forms) - from LDS,
    and the light distribution of a scene with <= 3 lights as one 32-byte record per voxel (ld_rows8 / ld_dense8). Same values through the same arithmetic: the film
    with every one of these switched off (tables in HBM, the three separate distribution tables) is the same film bit for bit."""
    from rustracer_amd.scenes import cornell_box, mis_plates
    if scene == "cornell":
        d = cornell_box(96, 80, 16)                          # k_shade<1, .., LDSREC = 1>, ld_dense8
    elif scene == "mis-spheres":
        d = mis_plates(spp=8, analytic_spheres=True)          # QLIGHTS forms with every table in LDS
        d.film.xres, d.film.yres = 160, 90
    elif scene == "mis-plates-small":
        d = mis_plates(spp=4)                                 # LEAN forms: materials + textures in LDS (1282 lights stay in HBM)
        d.film.xres, d.film.yres = 160, 90
    elif scene == "zoo-plastic-infinite":
        d = _zoo("plastic", "infinite", res=(64, 48))         # plain forms: lights + materials + textures + image headers (LDSREC = 3); rows8 with one area light + environment
    else:
        d = _zoo("matte_image_ewa", "area", res=(64, 48))
    films = []
    for off in (False, True):
        for k in ("RTX_SHADE_LDSREC", "RTX_LD_ROWS8", "RTX_LD_DENSE_MB"):
            if off:
                monkeypatch.setenv(k, "0")
            else:
                monkeypatch.delenv(k, raising=False)
        h = gpu_host.HostScene(d)                             # the knobs are read when the scene and its light distribution are built
        f, st = h.render()
        films.append((f, st))
    (f0, s0), (f1, s1) = films
    assert np.array_equal(f0.view(np.uint32), f1.view(np.uint32))
    for k in ("rays_closest", "rays_shadow", "rays_mis", "camera_rays"):
        assert s0[k] == s1[k], k
    assert np.isfinite(f0).all() and f0[..., :3].max() > 0


@pytest.mark.parametrize("material", ["matte", "oren_nayar", "matte_image_ewa", "plastic", "plastic_noremap", "metal", "metal_aniso", "mirror", "glass", "glass_rough", "substrate", "matte_bump_fbm", "plastic_bump_image", "mirror_bump_checker", "mix", "uber", "uber_opaque"])
def test_register_resident_front_ends_equal_the_generic_one(gpu_host, material, monkeypatch):
    """Class-wise dispatch (k_shade<3> / k_shade<5> / k_shade<6> / k_shade<0>) against every class through the generic lobe array."""
    d = _zoo(material, "infinite")
    h = gpu_host.HostScene(d)
    monkeypatch.setenv("RTX_SHADE_SPLIT", "0")
    f0, s0 = h.render()
    monkeypatch.setenv("RTX_SHADE_SPLIT", "2")
    f2, s2 = h.render()
    assert np.array_equal(f0[..., 3], f2[..., 3])
    assert rel_l2(f2[..., :3], f0[..., :3]) < 1e-6
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(s0[k]) - int(s2[k])) <= 4, k


def test_cropped_film_and_pixel_bounds_with_the_class_wise_dispatch(gpu_host, orc):
    """cropwindow + pixelbounds (samples outside are never traced) on a scene that is binned, class-dispatched and resolved by flag scan."""
    d = _zoo("plastic", "infinite", res=(48, 40), spp=8)
    d.film.crop = (0.1, 0.9, 0.2, 0.85)
    d.integrator.pixel_bounds = (10, 40, 12, 30)
    _check(gpu_host, orc, d)
    d2 = _zoo("mix_nested", "point", res=(48, 40), spp=8)
    d2.integrator.pixel_bounds = (0, 30, 5, 40)
    d2.film.filter_kind, d2.film.filter_params = 1, (1.5, 1.5, 0.0, 0.0)   # triangle filter: samples splat onto neighbouring pixels
    _check(gpu_host, orc, d2)


def test_sharded_render_of_a_class_dispatched_scene_sums_to_the_full_frame(gpu_host):
    """Film shards (one process per GPU in production) of a scene that takes every shade kernel: their sum is the full frame, bit for bit."""
    from rustracer_amd.distributed import owned_pixel_mask
    from util import bits
    d = _zoo("mix_nested", "infinite", res=(40, 70), spp=4)
    h = gpu_host.HostScene(d)
    full, _ = h.render()
    st = h.setup()
    for world in (2, 8):
        acc = np.zeros_like(full)
        for r in range(world):
            part, _ = h.render(rank=r, world_size=world)
            m = owned_pixel_mask(st["cropped"], st["sample_bounds"], r, world)
            assert np.all(part[~m] == 0)
            acc += part
        assert np.array_equal(bits(acc), bits(full))


@pytest.mark.parametrize("material", ["plastic", "metal"])
def test_a_mirror_sharp_lobe_agrees_at_its_first_vertex(gpu_host, orc, material):
    """Found by scripts/fuzz_shading.py: roughness 0.001 taken as alpha itself (remaproughness false) is a lobe 1e-3 rad wide whose D reads 1 - cos^2 of the half vector where
    that is ~1e-6; with the half vector normalised through v_rsq_f32 (one ulp in its length) direct-light frames were 3e-3 ... 1.6e-2 from the oracle's. Below alpha 0.02 the
    lobe takes the correctly rounded normalisation (rtx_dev_bsdf.h sharp_lobe; checked to fail with -DRT_SHARP_ALPHA=0.0f)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import exp_sharp_lobes as E
    d = E.scene(material, "none", 0.001, False, spp=64, depth=1)
    fo, _ = orc.OracleScene(d).render(mode=1)
    fh, _ = gpu_host.HostScene(d).render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-4

