"""Oracle-independent invariants on the ORACLE (CPU): closed-form images (white furnace, Lambert's polygon form factor), the agreement of the
three direct-light estimators (light sampling, BSDF sampling, MIS), and quadrature bounds on every non-specular BSDF. These do not depend on
the reference's test vectors nor on the oracle's author having read the Rust right twice the same way: a wrong pdf, weight or cosine shows."""
import ctypes as C

import numpy as np
import pytest

import invariants as inv


def _rgb(orc, d, **kw):
    film, st = orc.OracleScene(d).render(mode=1, **kw)
    return orc.film_to_rgb(film)


@pytest.mark.parametrize("rho,depth", [(0.5, 1), (0.5, 5), (0.9, 3)])
def test_white_furnace_convex_body_under_constant_environment(orc, rho, depth):
    d = inv.furnace_scene(rho, depth)
    img = _rgb(orc, d)
    body = inv.furnace_body_mask(img, rho, depth)
    assert body.sum() > 300 and (~body).sum() > 100
    got = img[body].mean(axis=0)
    assert np.allclose(got, rho, rtol=0.005), (got, rho)  # L = rho at any depth: a ray leaving a convex body never comes back
    sky = img[img[..., 1] > 0.999]            # pixels no sample of which hit the body see the environment itself: exactly 1 (up to the film's
    exact = np.abs(sky - 1.0).max(axis=-1) < 5e-6  # RGB -> XYZ -> RGB round trip); a silhouette pixel may land above 0.999 by noise
    assert len(sky) > 50 and exact.mean() > 0.9


@pytest.mark.parametrize("kind", ["ball", "box"])
def test_white_furnace_of_a_two_level_instance(orc, kind):
    """The body placed by a rotated, non-uniformly scaled, mirrored instance of an object holding one stretched sphere / a closed box of triangles
    (TransformedPrimitive, rc/primitive.rs:79-118): an affine image of a convex body is convex, L = rho still."""
    rho, depth = 0.8, 4
    img = _rgb(orc, inv.furnace_instances_scene(kind, rho, depth))
    body = inv.furnace_body_mask(img, rho, depth)
    assert body.sum() > 300
    assert np.allclose(img[body].mean(axis=0), rho, rtol=0.005), (img[body].mean(axis=0), rho)
    assert np.abs(img[body] / rho - 1).max() < 0.15


@pytest.mark.parametrize("rho,depth", [(0.5, 0), (0.5, 1), (0.5, 2), (0.5, 5), (0.8, 5), (0.25, 8)])
def test_closed_emissive_box_geometric_series(orc, rho, depth):
    d = inv.furnace_box_scene(rho, depth)
    img = _rgb(orc, d)
    want = inv.furnace_box_expected(rho, depth)
    assert np.allclose(img.mean(axis=(0, 1)), want, rtol=0.005), (img.mean(axis=(0, 1)), want)
    assert np.abs(img / want - 1).max() < 0.5  # every pixel, not just the mean (64 spp of a high-albedo furnace are noisy)


def test_lambert_polygon_form_factor(orc):
    d = inv.form_factor_scene()
    img = _rgb(orc, d)[..., 0]
    want = inv.form_factor_expected(d.film.xres)
    assert abs(img.mean() / want.mean() - 1) < 0.005, (img.mean(), want.mean())
    b = 8  # block means: the shape of the falloff, not just its integral
    gm, wm = img.reshape(4, b, 4, b).mean(axis=(1, 3)), want.reshape(4, b, 4, b).mean(axis=(1, 3))
    assert np.allclose(gm, wm, rtol=0.03), (gm / wm)


@pytest.mark.parametrize("kind,rough", [("plastic", 0.25), ("metal", 0.2), ("substrate", 0.3)])
def test_light_sampling_bsdf_sampling_and_mis_agree(orc, kind, rough):
    d = inv.glossy_scene(kind, rough, res=32, spp=1024)
    o = orc.OracleScene(d)
    mis, light, bsdf = (orc.film_to_rgb(o.render(mode=1, mis_mode=m)[0]) for m in (0, 1, 2))
    # the same direct lighting three ways: image means agree, and so do the means of the bright half and of the dim half of the plate (where
    # BSDF sampling alone rarely finds the emitter: a heavy-tailed estimator, -8 % at 256 spp, -0.1 % at 2048)
    lum = mis.mean(axis=-1)
    for region, tol in ((np.ones_like(lum, bool), 0.01), (lum > np.median(lum), 0.01), (lum <= np.median(lum), 0.06)):
        a, b, c = mis[region].mean(), light[region].mean(), bsdf[region].mean()
        assert abs(b / a - 1) < tol and abs(c / a - 1) < tol, (kind, a, b, c)
    # and MIS is the better estimator: lower per-pixel spread against the (converging) average of the three than the worse of its parts
    ref = (mis + light + bsdf) / 3
    err = [np.abs(x - ref).mean() for x in (mis, light, bsdf)]
    assert err[0] <= max(err[1], err[2])


def _probe(orc, sc, mat, wo, wi):
    f = np.zeros(3, np.float32)
    pdf = C.c_float()
    smp = np.zeros(8, np.float32)
    u = np.float32([0.5, 0.5])
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    orc.lib().orc_bsdf_probe(sc.h, mat, fp(np.float32(wo)), fp(np.float32(wi)), fp(u), fp(f), C.byref(pdf), fp(smp))
    return f, pdf.value


@pytest.fixture(scope="module")
def lobe_scene(orc):
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    mats = dict(matte=s.matte((0.8, 0.8, 0.8)), oren=s.matte((0.8, 0.8, 0.8), sigma=30.0), plastic=s.plastic((0.4,) * 3, (0.5,) * 3, 0.4),
                metal=s.metal(roughness=0.4), substrate=s.substrate((0.5,) * 3, (0.5,) * 3, 0.4, 0.3), rough_glass=s.glass(urough=0.4, vrough=0.4),
                uber=s.uber((0.4,) * 3, (0.4,) * 3, roughness=0.4), translucent=s.translucent((0.5,) * 3, (0.4,) * 3, roughness=0.4),
                disney=s.disney((0.8, 0.8, 0.8), roughness=0.5), disney_metal=s.disney((0.8,) * 3, metallic=1.0, roughness=0.4))
    s.add_quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), mats["matte"])
    return orc.OracleScene(s), mats


@pytest.mark.parametrize("name", ["matte", "oren", "plastic", "metal", "substrate", "rough_glass", "uber", "translucent", "disney", "disney_metal"])
@pytest.mark.parametrize("theta_o", [10.0, 50.0, 75.0])
def test_bsdf_energy_and_pdf_by_quadrature(orc, lobe_scene, name, theta_o):
    """Deterministic quadrature over the whole sphere of directions: the albedo int f |cos| dw never exceeds 1 (passive surface), the sampling
    density integrates to at most 1 (directions a lobe generates below the horizon are rejected, not redistributed) and to a sizeable part of it."""
    sc, mats = lobe_scene
    wo = np.float32([np.sin(np.radians(theta_o)), 0.0, np.cos(np.radians(theta_o))])
    n_mu, n_phi = 96, 192
    mu, w_mu = np.polynomial.legendre.leggauss(n_mu)          # cos(theta) in [-1, 1]
    phi = (np.arange(n_phi) + 0.5) * (2 * np.pi / n_phi)
    albedo, pdf_int, pdf_valid = np.zeros(3), 0.0, 0.0
    for m, wm in zip(mu, w_mu):
        s = np.sqrt(max(0.0, 1 - m * m))
        for p in phi:
            wi = np.array([s * np.cos(p), s * np.sin(p), m])
            f, pdf = _probe(orc, sc, mats[name], wo, wi)
            assert pdf >= 0.0 and np.all(f >= 0.0)
            albedo += f * abs(m) * wm * (2 * np.pi / n_phi)
            pdf_int += pdf * wm * (2 * np.pi / n_phi)
            # a refraction configuration is physical only if wo and wi lie on opposite sides of the half vector (the test pbrt has and the
            # reference lacks, see below); eta = 1.5 for both dielectric materials of this zoo
            wh = wo + wi * 1.5
            if m > 0 or np.dot(wo, wh) * np.dot(wi, wh) < 0:
                pdf_valid += pdf * wm * (2 * np.pi / n_phi)
    # Disney's lobes are "roughly" albedo-preserving by design (disney.rs:270); the reference's microfacet transmission lacks the same-side
    # rejection (quirk 8), so rough glass / translucent are checked on the pdf only
    if name not in ("rough_glass", "translucent"):
        assert np.all(albedo <= (1.25 if name.startswith("disney") else 1.02)), (name, theta_o, albedo)
    assert np.all(albedo > 0.05)
    # MicrofacetTransmission::pdf (microfacet.rs:213-227) lacks pbrt's `wo.wh * wi.wh > 0 => 0` rejection as well and does not flip wh to the
    # upper hemisphere: the reference's density of a rough dielectric integrates to a little more than 1 (1.14 at 50 degrees); kept as it is
    # upper hemisphere: the reference's density of a rough dielectric integrates to more than 1 (1.14 at 50 degrees, 1.43 at 75); kept as it is.
    # Restricted to the physical configurations most of the excess goes (1.43 -> 1.07 at 75 degrees, 1.00 at 10 and 50): it is that missing rejection.
    if name in ("rough_glass", "translucent"):
        assert 0.4 < pdf_valid <= (1.1 if theta_o > 60 else 1.02) and pdf_int < 1.6, (name, theta_o, pdf_int, pdf_valid)
    else:
        assert 0.4 < pdf_int <= 1.02, (name, theta_o, pdf_int)
    if name in ("matte", "oren"):  # cosine-weighted sampling: exactly normalised; Lambert's albedo is Kd
        assert abs(pdf_int - 1) < 2e-3
    if name == "matte":
        assert np.allclose(albedo, 0.8, rtol=2e-3)
