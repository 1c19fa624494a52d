"""The closed-form invariants of tests/invariants.py on the HIP path (through the C ABI): what the GPU renders is checked against
arithmetic that owes nothing to the oracle - the white furnace, the closed emissive box's geometric series, Lambert's polygon form factor -
and against the oracle's light-sampling-only estimator (an unbiased estimator the product does not even contain)."""
import numpy as np
import pytest

import invariants as inv

pytestmark = pytest.mark.gpu


def _rgb(gpu_host, d):
    film, st = gpu_host.HostScene(d).render()
    assert st["paths_scrubbed"] == 0
    return gpu_host.film_to_rgb(film)


@pytest.mark.parametrize("rho,depth", [(0.5, 1), (0.5, 5), (0.9, 3)])
def test_white_furnace_convex_body_under_constant_environment(gpu_host, rho, depth):
    d = inv.furnace_scene(rho, depth, res=96, spp=64)
    img = _rgb(gpu_host, d)
    body = inv.furnace_body_mask(img, rho, depth)
    assert body.sum() > 1000
    assert np.allclose(img[body].mean(axis=0), rho, rtol=0.004), (img[body].mean(axis=0), rho)
    sky = img[img[..., 1] > 0.999]
    assert len(sky) > 200 and (np.abs(sky - 1.0).max(axis=-1) < 5e-6).mean() > 0.9


@pytest.mark.parametrize("kind", ["ball", "box"])
@pytest.mark.parametrize("rho,depth", [(0.5, 1), (0.8, 4)])
def test_white_furnace_of_a_two_level_instance(gpu_host, kind, rho, depth):
    """The furnace's body placed by a rotated, non-uniformly scaled, mirrored INSTANCE of an object that holds one stretched sphere (object_walk_general, the
    quadric's interaction built in object space and transformed, rtx_shade_kernels.h instance_fill_interaction<true>) or a closed box of triangles
    (k_trace_inst): still convex, still L = rho in every body pixel."""
    d = inv.furnace_instances_scene(kind, rho, depth, res=96, spp=64)
    img = _rgb(gpu_host, d)
    body = inv.furnace_body_mask(img, rho, depth)
    assert body.sum() > 800
    assert np.allclose(img[body].mean(axis=0), rho, rtol=0.004), (img[body].mean(axis=0), rho)
    assert np.abs(img[body] / rho - 1).max() < 0.15


@pytest.mark.parametrize("rho,depth", [(0.5, 0), (0.5, 1), (0.5, 2), (0.5, 5), (0.8, 5), (0.25, 8)])
def test_closed_emissive_box_geometric_series(gpu_host, rho, depth):
    d = inv.furnace_box_scene(rho, depth, res=64, spp=64)
    img = _rgb(gpu_host, d)
    want = inv.furnace_box_expected(rho, depth)
    assert np.allclose(img.mean(axis=(0, 1)), want, rtol=0.003), (img.mean(axis=(0, 1)), want)
    assert np.abs(img / want - 1).max() < 0.5


def test_lambert_polygon_form_factor(gpu_host):
    d = inv.form_factor_scene(res=64, spp=256)
    img = _rgb(gpu_host, d)[..., 0]
    want = inv.form_factor_expected(64)
    assert abs(img.mean() / want.mean() - 1) < 0.003, (img.mean(), want.mean())
    gm, wm = img.reshape(8, 8, 8, 8).mean(axis=(1, 3)), want.reshape(8, 8, 8, 8).mean(axis=(1, 3))
    assert np.allclose(gm, wm, rtol=0.03), (gm / wm)


@pytest.mark.parametrize("kind,rough", [("plastic", 0.25), ("metal", 0.2), ("substrate", 0.3)])
def test_mis_agrees_with_light_sampling_alone(gpu_host, orc, kind, rough):
    """The GPU's estimate_direct (MIS, the only mode the product has) against the oracle's light-sampling-only estimator of the same direct
    lighting: unbiased both, so their means agree; a wrong light pdf or BSDF pdf on the device would tilt the MIS weights and show here."""
    d = inv.glossy_scene(kind, rough, res=32, spp=1024)
    mis = _rgb(gpu_host, d)
    light = orc.film_to_rgb(orc.OracleScene(d).render(mode=1, mis_mode=1)[0])
    lum = light.mean(axis=-1)
    for region, tol in ((np.ones_like(lum, bool), 0.01), (lum > np.median(lum), 0.01), (lum <= np.median(lum), 0.03)):
        assert abs(mis[region].mean() / light[region].mean() - 1) < tol, (kind, mis[region].mean(), light[region].mean())
