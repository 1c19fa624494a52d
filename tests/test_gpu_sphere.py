"""Analytic spheres (Shape "sphere", rc/shapes/sphere.rs) on the HIP path: hit records bit-for-bit against the oracle, frames within the image gate,
and closed forms that owe nothing to either (a Lambertian sphere in a furnace; the irradiance of a spherical emitter)."""
import numpy as np
import pytest

from util import bits, random_rays, rel_l2

pytestmark = pytest.mark.gpu


def _rot(axis, deg):
    a = np.radians(deg); c, s = np.cos(a), np.sin(a)
    x, y, z = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    m = np.eye(4)
    m[:3, :3] = [[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                 [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                 [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]]
    return m


def _xf(center, scale=(1, 1, 1), rot=None):
    m = np.eye(4)
    m[:3, 3] = center
    if rot is not None:
        m = m @ rot
    return (m @ np.diag(list(scale) + [1])).astype(np.float32)


def _sphere_zoo(res=48, spp=16):
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(res, res, spp)
    d.add_sphere((150, 100, 200), 70.0, d.plastic((0.2, 0.4, 0.7), (0.5, 0.5, 0.5), 0.1))                              # whole sphere
    d.add_sphere(radius=60.0, material=d.matte((0.8, 0.3, 0.2)), o2w=_xf((400, 380, 300), rot=_rot((1, 0.3, 0), 40)), z_min=-25.0, z_max=45.0)   # a clipped band
    d.add_sphere(radius=50.0, material=d.metal(roughness=0.1), o2w=_xf((300, 120, 120), (1.0, 0.6, 1.4), _rot((0, 1, 0), 30)), phi_max=250.0)  # an ellipsoid wedge
    d.add_sphere(radius=40.0, material=d.matte((0.3, 0.8, 0.3)), o2w=_xf((120, 330, 380), (1, 1, -1)), reverse_orientation=True)              # mirrored + reversed
    d.add_sphere((420, 150, 100), 35.0, d.glass())
    d.add_sphere((278, 450, 250), 30.0, d.matte((0.0,) * 3), emission=(30.0, 25.0, 20.0))                                                    # a sphere light
    # the reference's other quadrics: a tilted ring (disk with a hole, partial sweep), a leaning cylinder, and a disk light under the ceiling
    d.add_disk(_xf((430, 60, 330), rot=_rot((1, 0, 0.2), -70)), 55.0, d.matte((0.7, 0.7, 0.2)), height=5.0, inner_radius=20.0, phi_max=300.0)
    d.add_cylinder(_xf((230, 0, 420), rot=_rot((1, 0, 0), -90) @ _rot((0, 1, 0), 12)), 35.0, d.plastic((0.6, 0.2, 0.5), (0.3, 0.3, 0.3), 0.2), z_min=5.0, z_max=210.0)
    d.add_disk(_xf((120, 520, 150), rot=_rot((1, 0, 0), 90)), 45.0, d.matte((0.0,) * 3), emission=(12.0, 14.0, 18.0))
    return d


def test_sphere_hits_match_the_oracle_bit_for_bit(gpu_host, orc):
    d = _sphere_zoo()
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    bo, bh = o.bvh(), h.bvh()
    assert all(np.array_equal(bo[k], bh[k]) for k in bo)  # the same tree over triangles and sphere boxes
    rays = random_rays(60000, np.float32([0, 0, 0]), np.float32([555, 555, 555]), 9)
    ro, rh = o.trace(rays), h.trace(rays)
    assert np.array_equal(ro["prim"], rh["prim"]) and np.array_equal(bits(ro["t"]), bits(rh["t"]))
    assert (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])
    n_tri = d.n_tris
    on_sphere = bo["ordered"][np.maximum(ro["prim"], 0)] >= n_tri
    assert (on_sphere & (ro["prim"] >= 0)).sum() > 2000
    rr = h.trace(rays, count=False)
    assert np.array_equal(ro["prim"], rr["prim"]) and np.array_equal(bits(ro["t"]), bits(rr["t"]))
    rays[:, 3] = np.random.default_rng(4).uniform(20, 700, len(rays)).astype(np.float32)
    ao, ah = o.trace(rays, True), h.trace(rays, True)
    assert np.array_equal(ao["occluded"], ah["occluded"]) and (ao["nodes"], ao["tris"]) == (ah["nodes"], ah["tris"])
    # rays that start inside spheres, graze them, or are shorter than the far root
    inside = random_rays(20000, np.float32([100, 50, 150]), np.float32([200, 150, 250]), 5)
    io, ih = o.trace(inside), h.trace(inside)
    assert np.array_equal(io["prim"], ih["prim"]) and np.array_equal(bits(io["t"]), bits(ih["t"]))


def test_sphere_zoo_render_matches_oracle(gpu_host, orc):
    d = _sphere_zoo(64, 32)
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render(count_traversal=True)
    assert np.array_equal(fo[..., 3], fh[..., 3])
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis", "nodes_closest", "tris_closest"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])
    fh2, _ = gpu_host.HostScene(d).render()
    assert rel_l2(gpu_host.film_to_rgb(fh2), orc.film_to_rgb(fo)) < 1e-3


def test_veach_style_plates_with_four_sphere_lights(gpu_host, orc):
    from rustracer_amd.scenes import mis_plates
    d = mis_plates(160, 96, 32, analytic_spheres=True)
    assert len(d.lights) == 4 + 2 and len(d.spheres) == 4
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])
    # Sphere::pdf_wi is non-zero for every direction, so the reference casts a BSDF-sampled ray per vertex that picked a sphere light; the ones that miss the
    # sphere's box are counted, not cast - and change nothing: the frame that walks every ray as the reference does (counting kernels) is the same frame
    assert int(sh["rays_mis_not_cast"]) > 0.5 * int(sh["rays_mis"])
    fc, sc = gpu_host.HostScene(d).render(count_traversal=True)
    assert int(sc["rays_mis_not_cast"]) == 0 and abs(int(sc["rays_mis"]) - int(sh["rays_mis"])) <= 2e-3 * int(sh["rays_mis"]) + 16
    assert np.array_equal(fc[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), gpu_host.film_to_rgb(fc)) < 1e-6


def _furnace_sphere(rho, depth, res=64, spp=64):
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    s.add_sphere((0, 0, 0), 0.6, s.matte((rho,) * 3))
    s.add_quad((40, 40, 40), (40.01, 40, 40), (40.01, 40.01, 40), (40, 40.01, 40), s.matte((0.0,) * 3))  # (the soup must hold a triangle: a speck, far away)
    s.infinite_light(s.add_mip(np.ones((4, 8, 3), np.float32)))
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.0, -2.2, 1.2), (0, 0, 0), (0, 0, 1), 30.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = depth
    return s


@pytest.mark.parametrize("who", ["gpu", "oracle"])
@pytest.mark.parametrize("rho,depth", [(0.5, 1), (0.8, 4)])
def test_lambertian_sphere_in_a_furnace(gpu_host, orc, who, rho, depth):
    """A convex body of albedo rho in a constant environment of radiance 1 shows L = rho at any depth (tests/invariants.py)."""
    import invariants as inv
    d = _furnace_sphere(rho, depth)
    img = gpu_host.film_to_rgb(gpu_host.HostScene(d).render()[0]) if who == "gpu" else orc.film_to_rgb(orc.OracleScene(d).render(mode=1)[0])
    body = inv.furnace_body_mask(img, rho, depth)
    assert body.sum() > 400
    assert np.allclose(img[body].mean(axis=0), rho, rtol=0.005), img[body].mean(axis=0)


@pytest.mark.parametrize("who", ["gpu", "oracle"])
def test_irradiance_of_a_spherical_emitter(gpu_host, orc, who):
    """A Lambertian floor under an emitting sphere, direct light only: L = rho * Le * (r / D)^2 * cos(theta) (the whole sphere is above the horizon),
    whether the sphere is sampled inside its cone (Sphere::sample_si), found by the BSDF-sampled ray (Sphere::pdf_wi) or both (MIS)."""
    from rustracer_amd.scene_desc import SceneDesc
    rho, le, r, hz, cam = 0.6, 20.0, 0.5, 3.0, 2.0
    s = SceneDesc()
    s.add_quad((-6, -6, 0), (6, -6, 0), (6, 6, 0), (-6, 6, 0), s.matte((rho,) * 3))
    s.add_sphere((0, 0, hz), r, s.matte((0.0,) * 3), emission=(le,) * 3)
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.0, 0.0, cam), (0, 0, 0), (0, 1, 0), 60.0
    s.film.xres = s.film.yres = 64
    s.sampler.spp = 256
    s.integrator.max_depth = 1
    img = (gpu_host.film_to_rgb(gpu_host.HostScene(s).render()[0]) if who == "gpu" else orc.film_to_rgb(orc.OracleScene(s).render(mode=1)[0]))[..., 0]
    t = np.tan(np.radians(60.0) / 2) * cam
    c = (2 * (np.arange(64) + 0.5) / 64 - 1) * t
    x, y = np.meshgrid(c, c)
    D2 = x * x + y * y + hz * hz
    want = rho * le * (r * r / D2) * (hz / np.sqrt(D2))
    assert abs(img.mean() / want.mean() - 1) < 0.004, (img.mean(), want.mean())
    gm, wm = img.reshape(8, 8, 8, 8).mean(axis=(1, 3)), want.reshape(8, 8, 8, 8).mean(axis=(1, 3))
    assert np.allclose(gm, wm, rtol=0.03), gm / wm


def test_spheres_through_a_pbrt_file(gpu_host, tmp_path):
    """Shape "sphere" in the .pbrt loader: same primitives, same film as the scene built call by call. (In a file a shape's lights follow it in order of
    appearance; the exporter writes spheres after the meshes, so the zoo's light order - mesh lights, then the sphere light - is the file's too.)"""
    from rustracer_amd.pbrt_export import write_pbrt
    d = _sphere_zoo(40, 8)
    path = str(tmp_path / "spheres.pbrt")
    write_pbrt(d, path)
    a, sa = gpu_host.HostScene(d).render()
    p = gpu_host.PbrtScene(path)
    b, sb = p.render()
    assert np.array_equal(a[..., 3], b[..., 3]) and rel_l2(gpu_host.film_to_rgb(b), gpu_host.film_to_rgb(a)) < 2e-4
    assert sa["rays_closest"] == sb["rays_closest"] or abs(int(sa["rays_closest"]) - int(sb["rays_closest"])) < 1e-3 * sa["rays_closest"]


@pytest.mark.parametrize("who", ["gpu", "oracle"])
def test_irradiance_of_a_disk_emitter(gpu_host, orc, who):
    """A Lambertian floor under a parallel emitting disk of radius R at height h, direct light only: at distance a from the axis
    E = pi L / 2 * (1 - (h^2 + a^2 - R^2) / sqrt((h^2 + a^2 + R^2)^2 - 4 R^2 a^2)) (the classic disk-to-element form factor); L_floor = rho / pi * E.
    Disk::sample + the trait's default sample_si / pdf_wi (area measure converted to solid angle) and MIS against the BSDF-sampled ray."""
    from rustracer_amd.scene_desc import SceneDesc
    rho, le, rad, hz, cam = 0.6, 8.0, 0.7, 3.0, 2.0
    s = SceneDesc()
    s.add_quad((-6, -6, 0), (6, -6, 0), (6, 6, 0), (-6, 6, 0), s.matte((rho,) * 3))
    o2w = np.eye(4, dtype=np.float32); o2w[2, 3] = hz
    s.add_disk(o2w, rad, s.matte((0.0,) * 3), emission=(le,) * 3, reverse_orientation=True)  # its normal is +z: reversed, it emits downwards
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.0, 0.0, cam), (0, 0, 0), (0, 1, 0), 60.0
    s.film.xres = s.film.yres = 64
    s.sampler.spp = 256
    s.integrator.max_depth = 1
    img = (gpu_host.film_to_rgb(gpu_host.HostScene(s).render()[0]) if who == "gpu" else orc.film_to_rgb(orc.OracleScene(s).render(mode=1)[0]))[..., 0]
    t = np.tan(np.radians(60.0) / 2) * cam
    c = (2 * (np.arange(64) + 0.5) / 64 - 1) * t
    x, y = np.meshgrid(c, c)
    a2 = x * x + y * y
    want = rho / np.pi * (np.pi * le / 2) * (1 - (hz * hz + a2 - rad * rad) / np.sqrt((hz * hz + a2 + rad * rad) ** 2 - 4 * rad * rad * a2))
    assert abs(img.mean() / want.mean() - 1) < 0.004, (img.mean(), want.mean())
    gm, wm = img.reshape(8, 8, 8, 8).mean(axis=(1, 3)), want.reshape(8, 8, 8, 8).mean(axis=(1, 3))
    assert np.allclose(gm, wm, rtol=0.03), gm / wm


def _quadrics_among_many_triangles(res=40, spp=8):
    """Too many primitives for the LDS-resident kernel: the child-pair / quad-leaf / top-of-tree kernels' GENERAL instantiations take the scene."""
    from rustracer_amd.scenes import cornell_box
    from rustracer_amd.scenes.procedural import icosphere
    d = cornell_box(res, res, spp)
    P, F = icosphere(3, (300, 250, 300), 120.0)  # 1280 triangles
    d.add_mesh(P, F, d.plastic((0.5, 0.5, 0.2), (0.3, 0.3, 0.3), 0.2))
    img = np.zeros((16, 16, 3), np.float32)
    img[(np.add.outer(np.arange(16) // 4, np.arange(16) // 4) % 2) == 0] = 1.0
    mask = d.image_tex(d.add_mip(img, trilinear=True), su=3.0, sv=3.0)
    d.add_mesh([(60, 60, 120), (260, 60, 100), (260, 300, 140), (60, 300, 160)], [[0, 1, 2], [0, 2, 3]], d.matte((0.2, 0.5, 0.9)), UV=[(0, 0), (1, 0), (1, 1), (0, 1)], alpha=mask)
    d.add_sphere((120, 90, 380), 80.0, d.glass())
    d.add_sphere(radius=55.0, material=d.matte((0.8, 0.3, 0.2)), o2w=_xf((450, 400, 200), (1.0, 0.7, 1.3), _rot((1, 0.3, 0), 40)), z_min=-25.0, z_max=45.0)
    d.add_sphere((300, 250, 300), 90.0, d.metal(roughness=0.1))  # inside the mesh: reached only through its gaps - none - and by rays that start inside
    d.add_cylinder(_xf((480, 0, 450), rot=_rot((1, 0, 0), -90)), 30.0, d.matte((0.6, 0.2, 0.5)), z_min=5.0, z_max=300.0)
    d.add_sphere((100, 470, 100), 25.0, d.matte((0.0,) * 3), emission=(40.0, 35.0, 30.0))
    return d


def test_quadrics_and_masks_among_many_triangles_take_the_fast_kernels(gpu_host, orc):
    d = _quadrics_among_many_triangles()
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    assert len(o.bvh()["ordered"]) > 1280  # (the LDS-resident kernel holds 128 primitives)
    rays = random_rays(80000, np.float32([0, 0, 0]), np.float32([555, 555, 555]), 21)
    ro = o.trace(rays)
    for count in (True, False):  # the counting kernel, then the kernels rt_render launches
        rh = h.trace(rays, count=count)
        assert np.array_equal(ro["prim"], rh["prim"]) and all(np.array_equal(bits(ro[k]), bits(rh[k])) for k in ("t", "b0", "b1"))
    kinds = np.asarray(o.bvh()["ordered"])[np.maximum(ro["prim"], 0)]
    assert ((kinds >= d.n_tris) & (ro["prim"] >= 0)).sum() > 3000  # quadrics were hit
    rays[:, 3] = np.random.default_rng(4).uniform(20, 700, len(rays)).astype(np.float32)
    ao = o.trace(rays, True)
    assert np.array_equal(ao["occluded"], h.trace(rays, True)["occluded"]) and np.array_equal(ao["occluded"], h.trace(rays, True, count=False)["occluded"])
    assert 0.2 < ao["occluded"].mean() < 0.95
    fo, so = o.render(mode=1)
    fh, sh = h.render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])


def test_reference_sphere_reintersection_property_holds_on_the_device(gpu_host, orc):
    """rustracer-core/tests/shapes.rs:16-54 through the C ABI (rt_trace_closest / rt_trace_any): for seeded full spheres with radii 10^+-4 and ray origins up to
    1e8 away, the device finds the first hit exactly where the oracle does, and none of the 1000 rays spawned from it into the normal's hemisphere finds the
    sphere again - as occlusion query (Shape::intersect_p) or as closest hit (Shape::intersect). The spawned rays are the oracle's (offset_ray_origin on the
    oracle's interaction: the device returns hit records, not interactions); cases as in tests/test_oracle_kat.py."""
    import ctypes as C
    from rustracer_amd.scene_desc import SceneDesc
    from test_oracle_kat import sphere_reintersect_case
    L = orc.lib()
    L.orc_sphere_reintersect.restype = C.c_int
    L.orc_sphere_reintersect.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    tested = missed_box = 0
    for i in range(0, 1000, 4):  # every fourth of the reference's 1000 spheres: a scene is created per sphere
        radius, ray, u = sphere_reintersect_case(i)
        out = np.zeros((len(u), 8), np.float32)
        r = L.orc_sphere_reintersect(radius, ray.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), len(u), out.ctypes.data_as(C.c_void_p))
        d = SceneDesc()
        d.add_sphere((0.0, 0.0, 0.0), radius, d.matte((0.5, 0.5, 0.5)))
        h = gpu_host.HostScene(d)
        first = h.trace(ray[None, :], count=False)
        ro = orc.OracleScene(d).trace(ray[None, :])   # the same query on the oracle: Scene::intersect, i.e. the sphere behind its box in a one-leaf BVH
        assert np.array_equal(ro["prim"], first["prim"]) and np.array_equal(bits(ro["t"]), bits(first["t"])), (i, radius, ray)
        if r < 0:
            assert first["prim"][0] < 0
            continue
        if first["prim"][0] < 0:  # the shape is hit but its box is not: Bounds3::intersect_p_fast without the 1 + 2 gamma(3) widening (reference quirk 3) on a
            missed_box += 1       # sphere of radius 1e-4 seen from 1e7 away. Oracle and device agree (asserted above); the property below needs a hit.
            continue
        assert r == 0
        assert not h.trace(out, True, count=False)["occluded"].any(), (i, radius)
        assert (h.trace(out, count=False)["prim"] < 0).all(), (i, radius)
        tested += 1
    assert tested > 150 and missed_box < 10


def test_sphere_light_scenes_shade_triangle_vertices_on_the_three_wave_forms(gpu_host, orc):
    """Round 4: with constant textures and sphere lights that no triangle reaches into, vertices on triangles are shaded by the QLIGHTS
    forms of the register-resident front-ends (cone branches of Sphere::sample_si / pdf_wi inline) and only the vertices ON quadrics by the generic GENERAL
    kernel. Observable: the generic front-end's vertex count is exactly the number of path vertices on quadrics - non-zero although no material of the
    scene belongs to a generic class -; the frame is the oracle's. A triangle poking into an emitter sphere switches the scene back to the GENERAL forms
    (a vertex may then lie inside the sphere: the other branch of sample_si), where quadric hits are shaded by their material's own front-end."""
    from rustracer_amd.scenes import mis_plates
    d = mis_plates(96, 54, 16, analytic_spheres=True)
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-4
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 1e-3 * int(so[k]) + 4, (k, sh[k], so[k])
    assert 0 < sh["vertices_generic"] < 0.2 * (sh["vertices_lambert"] + sh["vertices_two_lobe"])
    # the same scene with a sliver of a triangle inside the first sphere light: no QLIGHTS forms, same agreement with the oracle
    d2 = mis_plates(96, 54, 16, analytic_spheres=True)
    sp = d2.spheres[0]
    c = np.float32(sp.o2w)[:3, 3]
    d2.add_mesh(np.float32([c, c + np.float32([0.01, 0, 0]), c + np.float32([0, 0.01, 0])]), [[0, 1, 2]], d2.matte((0.5, 0.5, 0.5)))
    fo2, so2 = orc.OracleScene(d2).render(mode=1)
    fh2, sh2 = gpu_host.HostScene(d2).render()
    assert np.array_equal(fo2[..., 3], fh2[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh2), orc.film_to_rgb(fo2)) < 1e-4
    assert sh2["vertices_generic"] == 0
