"""Object instances (ObjectBegin / ObjectInstance, TransformedPrimitive rc/primitive.rs:79-118) on the HIP path as a two-level traversal: hit records
bit-for-bit against the oracle, frames within the image gate, and the written-out (flattened) scene as a second, independent reference."""
import numpy as np
import pytest

from util import bits, random_rays, rel_l2

pytestmark = pytest.mark.gpu


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float64)
    m[0, 0] = c; m[0, 2] = s; m[2, 0] = -s; m[2, 2] = c
    return m


def _placements():
    out = []
    for k, (x, z, sc) in enumerate([(-2.0, 0.0, 1.0), (0.0, 1.0, 1.5), (2.0, -1.0, 0.7), (0.5, -2.0, 0.9)]):
        m = np.eye(4, dtype=np.float64)
        m[:3, 3] = (x, 0.8, z)
        m = m @ _rot_y(0.7 * k) @ np.diag([sc, sc * (1.3 if k == 3 else 1.0), sc, 1.0])
        if k == 2:
            m = m @ np.diag([-1.0, 1.0, 1.0, 1.0])  # mirrored: swaps handedness
        out.append(m.astype(np.float32))
    return out


def _scene(two_level, res=(48, 32), spp=8, single=True, material="plastic"):
    from rustracer_amd.scene_desc import SceneDesc
    from rustracer_amd.scenes.procedural import checker_fbm_image, icosphere
    s = SceneDesc()
    floor = s.matte((0.6, 0.5, 0.4))
    if material == "textured":
        img = s.add_mip(checker_fbm_image(64, 3, (0.8, 0.3, 0.2), (0.2, 0.3, 0.8), 8), max_aniso=8.0)
        m2 = s.plastic(s.image_tex(img, su=2, sv=2), (0.3, 0.3, 0.3), 0.1)
    elif material == "glass":
        m2 = s.glass(index=1.5)
    else:
        m2 = s.plastic((0.2, 0.3, 0.6), (0.3, 0.3, 0.3), 0.1)
    m3 = s.matte((0.7, 0.2, 0.2))
    s.add_quad((-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5), floor)
    P, F = icosphere(1, (0, 0, 0), 0.5)
    n = (P / np.float32(0.5)).astype(np.float32)
    uv = np.stack([np.arctan2(n[:, 2], n[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(n[:, 1], -1, 1)) / np.pi], -1).astype(np.float32)
    cap = dict(P=np.float32([[-0.2, 0.55, -0.2], [0.2, 0.55, -0.2], [0.2, 0.55, 0.2], [-0.2, 0.55, 0.2]]), idx=[[0, 1, 2], [0, 2, 3]], material=m3)  # a second mesh of the object: no N, no UV
    if two_level:
        o = s.add_object([dict(P=P, idx=F, material=m2, N=n, UV=uv), cap])
        for m in _placements():
            s.add_instance(o, m)
        if single:  # an object of ONE primitive is wrapped without an aggregate (api.rs:1073-1082)
            o1 = s.add_object([dict(P=np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), idx=[[0, 1, 2]], material=m3)])
            m = np.eye(4, dtype=np.float32); m[:3, 3] = (-0.5, 0.2, -2.5)
            s.add_instance(o1, m)
    else:  # the same scene written out: vertices through the instance matrix, normals through the inverse transpose, mirrored instances flipped
        for m in _placements():
            m64 = m.astype(np.float64)
            nit = np.linalg.inv(m64[:3, :3]).T
            flip = np.linalg.det(m64[:3, :3]) < 0
            s.add_mesh((P @ m64[:3, :3].T + m64[:3, 3]).astype(np.float32), F, m2, N=(n @ nit.T).astype(np.float32), UV=uv, reverse_orientation=bool(flip))
            s.add_mesh((cap["P"] @ m64[:3, :3].T + m64[:3, 3]).astype(np.float32), cap["idx"], m3, reverse_orientation=bool(flip))
        if single:
            s.add_mesh(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]) + np.float32((-0.5, 0.2, -2.5)), [[0, 1, 2]], m3)
    s.add_quad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), s.matte((0, 0, 0)), emission=(10, 10, 10))
    s.point_light((3.0, 3.0, -3.0), (4.0, 4.0, 5.0))
    s.camera.pos = (0, 3, -8); s.camera.look = (0, 0.5, 0); s.camera.fov = 40
    s.film.xres, s.film.yres = res
    s.sampler.spp = spp
    return s


def test_instance_hits_match_the_oracle_bit_for_bit(gpu_host, orc):
    d = _scene(True)
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    bo, bh = o.bvh(), h.bvh()
    assert all(np.array_equal(bo[k], bh[k]) for k in bo)  # the same top-level tree over triangles and instance boxes
    rays = random_rays(60000, np.float32([-4, 0, -4]), np.float32([4, 3, 4]), 21)
    ro, rh = o.trace(rays), h.trace(rays)
    assert np.array_equal(ro["prim"], rh["prim"]) and np.array_equal(bits(ro["t"]), bits(rh["t"]))
    assert np.array_equal(bits(ro["b0"]), bits(rh["b0"])) and np.array_equal(bits(ro["b1"]), bits(rh["b1"]))
    assert (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])  # nested visits included
    n_top = len(bo["ordered"])
    assert (ro["prim"] >= n_top).sum() > 3000  # hits inside instances carry ids past the top level's
    rr = h.trace(rays, count=False)
    assert np.array_equal(ro["prim"], rr["prim"]) and np.array_equal(bits(ro["t"]), bits(rr["t"]))
    rays[:, 3] = np.random.default_rng(5).uniform(0.3, 9.0, len(rays)).astype(np.float32)
    ao, ah = o.trace(rays, True), h.trace(rays, True)
    assert np.array_equal(ao["occluded"], ah["occluded"]) and (ao["nodes"], ao["tris"]) == (ah["nodes"], ah["tris"])


@pytest.mark.parametrize("material", ["plastic", "textured", "glass"])
def test_instanced_scene_matches_the_oracle(gpu_host, orc, material):
    d = _scene(True, material=material)
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render(count_traversal=True)
    assert np.array_equal(fo[..., 3], fh[..., 3])
    ro, rh = orc.film_to_rgb(fo), gpu_host.film_to_rgb(fh)
    assert np.isfinite(rh).all()
    assert rel_l2(rh, ro) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])
    assert int(sh["nodes_closest"]) == int(so["nodes_closest"]) or abs(int(sh["nodes_closest"]) - int(so["nodes_closest"])) <= 2e-3 * int(so["nodes_closest"])


def test_two_level_and_written_out_instances_agree(gpu_host):
    """TransformedPrimitive intersects in object space and carries a larger p_error than a mesh written out through the instance matrix: two roundings of
    one scene. The frames agree to the image gate (the flattened scene is what `instantiate` in the loader produces on request)."""
    a, _ = gpu_host.HostScene(_scene(True, spp=32, single=False)).render()
    b, _ = gpu_host.HostScene(_scene(False, spp=32, single=False)).render()
    assert np.array_equal(a[..., 3], b[..., 3])
    assert rel_l2(gpu_host.film_to_rgb(a), gpu_host.film_to_rgb(b)) < 5e-3


def test_instances_render_through_the_multi_device_entry_point(gpu_host):
    d = _scene(True, res=(40, 48), spp=4)
    h = gpu_host.HostScene(d)
    one, _ = h.render()
    two = h.render_multi(devices=[0, 0], chunks_per_device=2)[0]
    assert np.array_equal(bits(one), bits(two))


def test_a_scene_of_instances_only(gpu_host, orc):
    """No top-level triangle at all: one two-triangle object placed twice and a single-triangle object (wrapped without a tree) under a point light."""
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    m = s.matte((0.5, 0.6, 0.4))
    quad = s.add_object([dict(P=np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]]), idx=[[0, 1, 2], [1, 3, 2]], material=m)])
    tri = s.add_object([dict(P=np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), idx=[[0, 1, 2]], material=m)])
    for obj, t, sc in ((quad, (0.0, 0.0, 3.0), 2.0), (quad, (-2.5, -0.5, 4.0), 1.0), (tri, (1.0, -1.0, 2.5), 1.5)):
        mtx = np.diag([sc, sc, sc, 1.0]).astype(np.float32); mtx[:3, 3] = t
        s.add_instance(obj, mtx)
    s.point_light((0.5, 0.5, 0.0), (8.0, 8.0, 8.0))
    s.camera.pos, s.camera.look = (0.5, 0.5, -3.0), (0.5, 0.5, 3.0)
    s.film.xres, s.film.yres = 32, 24
    s.sampler.spp = 4
    fo, _ = orc.OracleScene(s).render(mode=1)
    fh, _ = gpu_host.HostScene(s).render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and orc.film_to_rgb(fo).mean() > 0
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3


def _glowing_objects(res=(64, 48), spp=16):
    """Instances of an object whose cap sits under an AreaLightSource: it glows when a camera ray or a mirror bounce reaches it and lights nothing
    (rc/api.rs:954-964 keeps the primitive's area light and drops the light from the scene's list). A mirror wall shows the specular case."""
    from rustracer_amd.scene_desc import SceneDesc
    from rustracer_amd.scenes.procedural import icosphere
    s = SceneDesc()
    s.add_quad((-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5), s.matte((0.6, 0.5, 0.4)))
    s.add_quad((-5, 0, 4), (-5, 4, 4), (5, 4, 4), (5, 0, 4), s.mirror(0.9))
    P, F = icosphere(1, (0, 0, 0), 0.5)
    body = dict(P=P, idx=F, material=s.plastic((0.2, 0.3, 0.6), (0.3, 0.3, 0.3), 0.1), N=(P / np.float32(0.5)).astype(np.float32))
    cap = dict(P=np.float32([[-0.3, 0.55, -0.3], [0.3, 0.55, -0.3], [0.3, 0.55, 0.3], [-0.3, 0.55, 0.3]]), idx=[[0, 2, 1], [0, 3, 2]], material=s.matte((0.0,) * 3),
               emission=(6.0, 4.0, 2.0), two_sided=True)
    o = s.add_object([body, cap])
    for m in _placements()[:3]:
        s.add_instance(o, m)
    s.point_light((0.0, 5.0, -3.0), (40.0, 40.0, 40.0))
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 3.0, -6.5), (0.0, 0.8, 0.5), 45.0
    s.film.xres, s.film.yres = res
    s.sampler.spp = spp
    return s


def test_emitters_inside_objects_glow_and_light_nothing(gpu_host, orc, tmp_path):
    d = _glowing_objects()
    assert len(d.lights) == 1 and len(d.emitters) == 1
    fo, so = orc.OracleScene(d).render(mode=1)
    h = gpu_host.HostScene(d)
    fh, sh = h.render(count_traversal=True)
    assert np.array_equal(fo[..., 3], fh[..., 3])
    ro, rh = orc.film_to_rgb(fo), gpu_host.film_to_rgb(fh)
    assert rel_l2(rh, ro) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])
    fp, _ = h.render()                                  # the production kernels
    assert rel_l2(gpu_host.film_to_rgb(fp), ro) < 1e-3
    # the caps are visible: the same scene without emission is darker where they are, and ONLY there or in the mirror (they light nothing)
    d0 = _glowing_objects(); d0.emitters[0] = ((0.0, 0.0, 0.0), True)
    r0 = orc.film_to_rgb(orc.OracleScene(d0).render(mode=1)[0])
    glow = (ro - r0).sum(-1)
    assert glow.max() > 1.0 and (glow > 0.05).mean() < 0.08 and np.abs(glow[glow <= 0.05]).max() < 1e-4
    # the same through a scene file, two-level and written out (a written-out instance keeps its emitter out of the light list as well)
    from rustracer_amd.pbrt_export import write_pbrt
    path = str(tmp_path / "glow.pbrt")
    write_pbrt(d, path)
    two = gpu_host.PbrtScene(path)
    flat = gpu_host.PbrtScene(path, flatten_instances=True)
    assert two.n_lights() == 1 and flat.n_lights() == 1
    ft, _ = two.render()
    ff, _ = flat.render()
    assert rel_l2(gpu_host.film_to_rgb(ft), rh) < 1e-5
    assert rel_l2(gpu_host.film_to_rgb(ff), ro) < 5e-3   # (two roundings of one scene, see test_two_level_and_written_out_instances_agree)


@pytest.mark.parametrize("extra", ["quadrics", "masks", "both"])
def test_instances_beside_quadrics_and_masked_meshes(gpu_host, orc, extra):
    """The trace kernels come in three general forms (instances only / + quadrics / + alpha masks, rtx_dev_scene.h RT_GEN_*): an instanced scene that also
    holds top-level quadrics and / or a masked mesh takes the other two; hit records of the production kernels and of the counting kernels equal the oracle's."""
    d = _scene(True)
    if extra in ("quadrics", "both"):
        d.add_sphere((2.5, 0.8, 1.0), 0.8, d.glass())
        d.add_sphere((-3.0, 0.5, 2.0), 0.5, d.matte((0.0,) * 3), emission=(6.0, 5.0, 4.0))
        m = np.eye(4, dtype=np.float32); m[:3, 3] = (0.0, 0.0, 3.0)
        d.add_cylinder(m, 0.3, d.matte((0.2, 0.7, 0.3)), z_min=0.0, z_max=1.5)
    if extra in ("masks", "both"):
        img = np.zeros((8, 8, 3), np.float32); img[::2, ::2] = 1.0; img[1::2, 1::2] = 1.0
        mask = d.image_tex(d.add_mip(img, trilinear=True), su=3.0, sv=3.0)
        d.add_mesh([(-4, 0.1, -1), (-1, 0.1, -1), (-1, 2.5, -0.5), (-4, 2.5, -0.5)], [[0, 1, 2], [0, 2, 3]], d.matte((0.9, 0.8, 0.1)), UV=[(0, 0), (1, 0), (1, 1), (0, 1)], alpha=mask)
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    rays = random_rays(60000, np.float32([-4, 0, -4]), np.float32([4, 3, 4]), 33)
    ro = o.trace(rays)
    for count in (True, False):
        rh = h.trace(rays, count=count)
        assert np.array_equal(ro["prim"], rh["prim"]) and all(np.array_equal(bits(ro[k]), bits(rh[k])) for k in ("t", "b0", "b1"))
    rays[:, 3] = np.random.default_rng(6).uniform(0.3, 9.0, len(rays)).astype(np.float32)
    ao = o.trace(rays, True)
    assert np.array_equal(ao["occluded"], h.trace(rays, True)["occluded"]) and np.array_equal(ao["occluded"], h.trace(rays, True, count=False)["occluded"])
    fo, so = o.render(mode=1)
    fh, sh = h.render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])


def _general_object_scene(kind, res=(48, 32), spp=8):
    """Objects that hold what a TransformedPrimitive wraps besides plain triangles (rc/primitive.rs:79-118, api.rs:1019-1051): quadrics (in object space, under a
    transform of their own inside the definition) and an alpha-masked mesh; placed by rotated, non-uniformly scaled and mirrored instances."""
    from rustracer_amd.scene_desc import SceneDesc
    from rustracer_amd.scenes.procedural import icosphere
    s = SceneDesc()
    floor, blue, red, glass = s.matte((0.6, 0.5, 0.4)), s.plastic((0.2, 0.3, 0.6), (0.3, 0.3, 0.3), 0.1), s.matte((0.7, 0.2, 0.2)), s.glass(index=1.5)
    s.add_quad((-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5), floor)
    s.add_quad((-1, 4.9, -1), (1, 4.9, -1), (1, 4.9, 1), (-1, 4.9, 1), floor, emission=(30.0, 28.0, 25.0))
    P, F = icosphere(1, (0, 0, 0), 0.35)
    q_in = np.eye(4, dtype=np.float32); q_in[:3, 3] = (0.0, 0.6, 0.0); q_in[1, 1] = 1.4            # the CTM inside the definition: a translation and a stretch
    c_in = (np.eye(4) @ _rot_y(0.4)).astype(np.float32); c_in[:3, 3] = (0.5, -0.2, 0.1)
    quadrics = [dict(kind=0, o2w=q_in, radius=0.45, z_min=-0.45, z_max=0.3, phi_max=300.0, material=glass),
                dict(kind=2, o2w=c_in, radius=0.15, z_min=0.0, z_max=0.9, material=red),
                dict(kind=1, o2w=q_in, radius=0.5, z_min=0.75, z_max=0.1, material=blue)]               # a disk: z_min = height, z_max = inner radius
    meshes = [dict(P=P, idx=F, material=blue)]
    if kind in ("masks", "both"):
        img = np.zeros((8, 8, 3), np.float32); img[::2, ::2] = 1.0; img[1::2, 1::2] = 1.0
        mask = s.image_tex(s.add_mip(img, trilinear=True), su=3.0, sv=3.0)
        meshes.append(dict(P=np.float32([[-0.7, -0.4, 0.5], [0.7, -0.4, 0.5], [0.7, 0.9, 0.6], [-0.7, 0.9, 0.6]]), idx=[[0, 1, 2], [0, 2, 3]], material=red,
                           UV=np.float32([(0, 0), (1, 0), (1, 1), (0, 1)]), alpha=mask, shadow_alpha=mask))
    o = s.add_object(meshes, quadrics=quadrics if kind in ("quadrics", "both") else None)
    for m in _placements():
        s.add_instance(o, m)
    if kind in ("quadrics", "both"):  # an object of ONE primitive that is a quadric (wrapped without an aggregate, api.rs:1073-1082), one of them emitting
        o1 = s.add_object([], quadrics=[dict(kind=0, o2w=q_in, radius=0.3, material=red, emission=(4.0, 3.0, 2.0))])
        m = np.eye(4, dtype=np.float32); m[:3, 3] = (-0.5, 0.3, -2.5); m[0, 0] = 1.5
        s.add_instance(o1, m)
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 2.5, -7.0), (0.0, 0.8, 0.0), 50.0
    s.film.xres, s.film.yres = res
    s.sampler.spp = spp
    return s


@pytest.mark.parametrize("kind", ["quadrics", "masks", "both"])
def test_objects_that_hold_quadrics_or_masked_meshes(gpu_host, orc, kind):
    """VERDICT r05 missing #1: rt_scene_create refused an instanced object with anything but plain triangles, and the host wrote such quadrics out under the PRODUCT
    instance_to_world * object_to_world - one rounding where the reference's TransformedPrimitive takes the ray to object space and the quadric takes it on to its own
    (two Transform * Ray, ray.rs:83-93). Now the object holds them: hit records - id, t, barycentrics, visit counts - and occlusion answers bit-equal to the oracle's
    TransformedPrimitive over its Scene of the object (orc_scene.cpp: object_intersect_raw / prim_test), production and counting kernels; the frame inside the gate."""
    d = _general_object_scene(kind)
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    bo, bh = o.bvh(), h.bvh()
    assert all(np.array_equal(bo[k], bh[k]) for k in bo)  # the top-level tree over the instances' world boxes (the object's root box holds the quadrics' boxes)
    rays = random_rays(60000, np.float32([-4, 0, -4]), np.float32([4, 3, 4]), 41)
    ro = o.trace(rays)
    n_top = len(bo["ordered"])
    assert (ro["prim"] >= n_top).sum() > 3000
    for count in (True, False):
        rh = h.trace(rays, count=count)
        assert np.array_equal(ro["prim"], rh["prim"]) and all(np.array_equal(bits(ro[k]), bits(rh[k])) for k in ("t", "b0", "b1")), count
        if count:
            assert (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])
    rays[:, 3] = np.random.default_rng(7).uniform(0.3, 9.0, len(rays)).astype(np.float32)
    ao = o.trace(rays, True)
    assert np.array_equal(ao["occluded"], h.trace(rays, True)["occluded"]) and np.array_equal(ao["occluded"], h.trace(rays, True, count=False)["occluded"])
    fo, so = o.render(mode=1)
    fh, sh = h.render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and np.isfinite(gpu_host.film_to_rgb(fh)).all()
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sh[k]) - int(so[k])) <= 2e-3 * int(so[k]) + 16, (k, sh[k], so[k])


def test_objects_first_instanced_in_another_order_than_defined(gpu_host, orc):
    """Found by scripts/fuzz_objects.py: the scene's quadric table held the objects' quadrics in the order of their DEFINITIONS while the object primitives named them in the order
    of the objects' FIRST INSTANCES (rtx_host.cpp commit: obj_sphere_base) - every scene that instanced its second quadric-holding object first met the wrong quadrics."""
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    m = s.matte((0.5, 0.5, 0.5))
    s.add_quad((-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5), m)
    a_in = np.eye(4, dtype=np.float32); a_in[:3, 3] = (0.0, 0.5, 0.0)
    b_in = (np.eye(4) @ _rot_y(0.7)).astype(np.float32); b_in[:3, 3] = (0.2, 0.3, 0.0)
    tri = dict(P=np.float32([[-0.4, 0, 0], [0.4, 0, 0], [0, 0.6, 0.2]]), idx=[[0, 1, 2]], material=m)
    a = s.add_object([tri], quadrics=[dict(kind=0, o2w=a_in, radius=0.4, material=m), dict(kind=2, o2w=a_in, radius=0.15, z_min=-0.5, z_max=0.5, material=m)])
    b = s.add_object([tri], quadrics=[dict(kind=1, o2w=b_in, radius=0.6, z_min=0.1, z_max=0.2, material=m), dict(kind=0, o2w=b_in, radius=0.25, phi_max=200.0, material=m)])
    c = s.add_object([], quadrics=[dict(kind=0, o2w=a_in, radius=0.3, material=m)])
    for k, (o, at) in enumerate([(c, (2.0, 0.5, 1.0)), (b, (-1.5, 0.4, 0.0)), (a, (0.5, 0.3, -1.0)), (b, (1.0, 1.0, 2.0)), (c, (-2.0, 0.8, -2.0))]):  # first instances: c, b, a
        mtx = (np.eye(4) @ _rot_y(0.5 * k)).astype(np.float32); mtx[:3, 3] = at; mtx[1, 1] = 1.0 + 0.2 * k
        s.add_instance(o, mtx)
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 2.5, -7.0), (0.0, 0.8, 0.0), 50.0
    s.film.xres, s.film.yres = 32, 24
    s.sampler.spp = 4
    o, h = orc.OracleScene(s), gpu_host.HostScene(s)
    rays = random_rays(40000, np.float32([-3, 0, -3]), np.float32([3, 2.5, 3]), 5)
    ro = o.trace(rays)
    assert (ro["prim"] >= len(o.bvh()["ordered"])).sum() > 2000
    for count in (True, False):
        rh = h.trace(rays, count=count)
        assert np.array_equal(ro["prim"], rh["prim"]) and all(np.array_equal(bits(ro[k]), bits(rh[k])) for k in ("t", "b0", "b1")), count
    rays[:, 3] = np.random.default_rng(8).uniform(0.3, 6.0, len(rays)).astype(np.float32)
    assert np.array_equal(o.trace(rays, True)["occluded"], h.trace(rays, True, count=False)["occluded"])
    fo, fh = o.render(mode=1)[0], h.render()[0]
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3


def test_a_rotated_disk_inside_an_object_has_its_light_distribution_voxels(gpu_host, orc):
    """Found by scripts/fuzz_objects.py: Disk::world_bounds ignores rotations (disk.rs:127-134; kept, it shapes the BVH), so a tilted disk's surface leaves the box the tree holds for
    it - and an instance's world box is the image of such boxes. The eager light-distribution build marked voxels from those boxes and a frame then met a voxel without a table
    (rt_render: "voxel marking bug"). k_lightdist_mark now widens an instance's box by the true boxes of its object's quadrics."""
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    m = s.matte((0.6, 0.6, 0.6))
    s.add_quad((-1, 3.0, -1), (1, 3.0, -1), (1, 3.0, 1), (-1, 3.0, 1), m, emission=(30.0, 30.0, 30.0))
    s.add_quad((-0.2, -2.0, -0.2), (-0.2, -2.0, 0.2), (0.2, -2.0, 0.2), (0.2, -2.0, -0.2), m)
    c, sn, cz, sz = np.cos(1.0), np.sin(1.0), np.cos(0.6), np.sin(0.6)
    # turned in its own plane (the two mapped corners no longer span the disk) and tilted (the box has a thickness: a ray goes through it and meets the disk outside it)
    tilt = (np.float64([[1, 0, 0, 0], [0, c, -sn, 0], [0, sn, c, 0], [0, 0, 0, 1]]) @ np.float64([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])).astype(np.float32)
    o = s.add_object([], quadrics=[dict(kind=1, o2w=tilt, radius=1.5, z_min=0.0, z_max=0.0, material=m)])
    for k in range(3):
        mtx = (np.eye(4) @ _rot_y(1.1 * k)).astype(np.float32); mtx[:3, 3] = (2.5 * (k - 1), 0.0, 0.5 * k)
        s.add_instance(o, mtx)
    s.camera.pos, s.camera.look, s.camera.fov = (0.0, 1.5, -8.0), (0.0, 0.0, 0.0), 50.0
    s.film.xres, s.film.yres = 64, 40
    s.sampler.spp = 16
    fo, so = orc.OracleScene(s).render(mode=1)
    fh, sh = gpu_host.HostScene(s).render()
    assert np.array_equal(fo[..., 3], fh[..., 3]) and rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3
    assert abs(int(sh["rays_shadow"]) - int(so["rays_shadow"])) <= 2e-3 * int(so["rays_shadow"]) + 16 and int(so["rays_shadow"]) > 2000


def _chain_mesh(n, material, base=13.0):
    """n triangles at x = 13^k, each as large as its x: of the SAH's 12 buckets the last holds the largest triangle alone and the first all the others
    (13^(n-1) / 13^n < 1 / 12), so every split peels one triangle off and the tree is a chain n - 1 deep."""
    P, idx = [], []
    for k in range(n):
        x = float(base ** k)
        P += [[x, 0.0, 0.0], [x, x, 0.0], [x, 0.0, x]]
        idx.append([3 * k, 3 * k + 1, 3 * k + 2])
    return dict(P=np.float32(P), idx=idx, material=material)


def _depth(b, first, n_nodes):
    d = np.zeros(n_nodes, int)
    for i in range(n_nodes):
        if b["n_prims"][first + i] == 0:
            d[i + 1] = d[i] + 1; d[int(b["offset"][first + i])] = d[i] + 1
    return int(d.max())


def test_top_level_plus_object_deeper_than_one_64_entry_stack(gpu_host, orc):
    """the reference gives each BVH its own 64-entry stack (bvh/mod.rs:374), so a deep top level over a deep object is a valid scene even when the
    two depths add up to more than 64 (round 3 refused it). It is traced by the one-node-per-step kernel with a 128-entry column; hit records equal the oracle's."""
    from rustracer_amd.scene_desc import SceneDesc
    s = SceneDesc()
    m = s.matte((0.5, 0.5, 0.5))
    top = _chain_mesh(35, m)
    s.add_mesh(top["P"], top["idx"], m)
    o = s.add_object([_chain_mesh(35, m)])
    xf = np.eye(4, dtype=np.float32); xf[:3, 3] = (0.0, 0.0, 0.0); xf[0, 0] = 0.5  # the object's triangles at x = 13^k / 2
    s.add_instance(o, xf)
    oc = orc.OracleScene(s)
    alone = SceneDesc(); alone.add_mesh(top["P"], top["idx"], alone.matte((0.5, 0.5, 0.5)))
    ba = orc.OracleScene(alone).bvh()
    assert 2 * (_depth(ba, 0, len(ba["offset"])) + 1) > 64   # each of the two trees needs 35 entries
    h = gpu_host.HostScene(s)
    bo, bh = oc.bvh(), h.bvh()
    assert all(np.array_equal(bo[k], bh[k]) for k in bo)
    rng = np.random.default_rng(21)
    n = 20000
    rays = np.zeros((n, 8), np.float32)
    k = rng.integers(0, 35, n)
    x = 13.0 ** k * np.where(rng.random(n) < 0.5, 1.0, 0.5)
    a = rng.uniform(0.05, 0.45, n); b = rng.uniform(0.05, 0.45, n)
    target = np.stack([x, a * 13.0 ** k, b * 13.0 ** k], -1)
    origin = target + np.stack([rng.choice([-1.0, 1.0], n) * 13.0 ** np.minimum(rng.uniform(k - 2, k + 2), 33.5), rng.normal(0, 0.3, n) * 13.0 ** k, rng.normal(0, 0.3, n) * 13.0 ** k], -1)
    rays[:, 0:3] = origin; rays[:, 3] = np.inf; rays[:, 4:7] = target - origin
    ro = oc.trace(rays)
    assert (ro["prim"] >= 0).mean() > 0.5 and len(np.unique(ro["prim"])) > 40
    for count in (True, False):
        rh = h.trace(rays, count=count)
        assert np.array_equal(ro["prim"], rh["prim"]) and np.array_equal(bits(ro["t"]), bits(rh["t"]))
    rh = h.trace(rays)
    assert (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])
    ao, ah = oc.trace(rays, True), h.trace(rays, True, count=False)
    assert np.array_equal(ao["occluded"], ah["occluded"])
