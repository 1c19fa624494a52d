"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (librtx_host.so -> librtx_hip.so),
against the oracle on the same seeded inputs, against the committed golden vectors, and - at
BASELINE.json's full size - through size-independent properties.

Tolerances: integer / index / bit-level work (BVH, hit records, sampler tables, light tables, film
weights) is bit-exact. Radiance is f32 with transcendental functions (sin/cos/log... differ by
<= 2 ulp between glibc and the ROCm device library), so images are compared by relative L2 of the
linear-RGB film, gate 1e-3 (BASELINE.json north_star); observed values are ~1e-6.
"""
import os

import numpy as np
import pytest

from util import bits, random_rays, rel_l2, tables_from_perm

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
L2_GATE = 1e-3


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "cornell_32x32_16spp.npz"))


def _cornell(*a, **k):
    from rustracer_amd.scenes import cornell_box
    return cornell_box(*a, **k)


# ---------------------------------------------------------------- kernels: bit-exact
def test_trace_kernels_match_golden(gpu_host, gold):
    h = gpu_host.HostScene(_cornell(32, 32, 16))
    hc = h.trace(gold["rays"])
    assert np.array_equal(hc["prim"], gold["hit_prim"])
    assert np.array_equal(bits(hc["t"]), bits(gold["hit_t"]))
    assert np.array_equal(bits(hc["b0"]), bits(gold["hit_b0"])) and np.array_equal(bits(hc["b1"]), bits(gold["hit_b1"]))
    assert (hc["nodes"], hc["tris"]) == (int(gold["hit_nodes"]), int(gold["hit_tris"]))
    ha = h.trace(gold["rays_any"], any_hit=True)
    assert np.array_equal(ha["occluded"], gold["occluded"])
    assert (ha["nodes"], ha["tris"]) == (int(gold["any_nodes"]), int(gold["any_tris"]))


@pytest.mark.parametrize("n_tris,max_prims", [(3000, 4), (257, 1), (100000, 4), (64, 4)])
def test_trace_kernels_match_oracle_on_soups(gpu_host, orc, n_tris, max_prims):
    from rustracer_amd.scenes import random_soup
    d = random_soup(n_tris, seed=n_tris, max_prims=max_prims, degenerate=(n_tris == 64))
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    rays = random_rays(60000, [-20, -20, -20], [120, 140, 120], seed=3)
    ro, rh = o.trace(rays), h.trace(rays)
    assert np.array_equal(ro["prim"], rh["prim"])
    for k in ("t", "b0", "b1"):
        assert np.array_equal(bits(ro[k]), bits(rh[k])), k
    assert (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])
    assert (ro["prim"] >= 0).mean() > 0.02
    rr = h.trace(rays, count=False)  # the kernels rt_render launches
    assert np.array_equal(ro["prim"], rr["prim"]) and all(np.array_equal(bits(ro[k]), bits(rr[k])) for k in ("t", "b0", "b1"))
    rays[:, 3] = np.random.default_rng(4).uniform(1, 150, len(rays)).astype(np.float32)
    ao, ah = o.trace(rays, True), h.trace(rays, True)
    assert np.array_equal(ao["occluded"], ah["occluded"]) and (ao["nodes"], ao["tris"]) == (ah["nodes"], ah["tris"])
    assert np.array_equal(ao["occluded"], h.trace(rays, True, count=False)["occluded"])


def test_trace_edge_cases(gpu_host, orc):
    d = _cornell(8, 8, 1)
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    rays = np.array([
        [278, 273, -800, np.inf, 0, 0, 1, 0],      # straight down the axis
        [278, 273, -800, np.inf, 0, 0, -1, 0],     # away from everything (miss)
        [278, 273, 100, 1e-3, 0, 1, 0, 0],         # t_max shorter than any hit
        [0, 0, 0, np.inf, 1, 0, 0, 0],             # along an edge: exercises e == 0 -> f64 fallback
        [0, 0, 0, np.inf, 0, 1e-30, 1, 0],         # denormal-ish component, inv_dir overflow
        [100, 548.8, 100, np.inf, 0, -1, 0, 0],    # origin exactly on the ceiling plane
    ], np.float32)
    ro, rh = o.trace(rays), h.trace(rays)
    assert np.array_equal(ro["prim"], rh["prim"]) and np.array_equal(bits(ro["t"]), bits(rh["t"]))
    assert rh["prim"][1] == -1 and rh["prim"][2] == -1
    ao, ah = o.trace(rays, True), h.trace(rays, True)
    assert np.array_equal(ao["occluded"], ah["occluded"])


@pytest.mark.parametrize("scene", ["cornell", "soup3000", "soup100000"])
def test_rays_with_zero_direction_components_take_the_reference_selects(gpu_host, orc, scene):
    """The kernels a frame launches test a node with minima / maxima when a ray's reciprocal direction is finite and with the reference's selects when it is
    not (0 * inf = NaN on a box bound). Waves of finite rays, waves of axis-parallel rays whose origins sit EXACTLY on box bounds (vertex coordinates), denormal
    components whose reciprocal overflows, and waves that mix the kinds: hit records and occlusion bit-equal to the oracle."""
    from rustracer_amd.scenes import random_soup
    d = _cornell(8, 8, 1) if scene == "cornell" else random_soup(int(scene[4:]), seed=11, max_prims=4)
    o, h = orc.OracleScene(d), gpu_host.HostScene(d)
    rng = np.random.default_rng(21)
    verts = d.arrays()[0]
    lo, hi = ([0, 0, -800], [556, 549, 560]) if scene == "cornell" else ([-20, -20, -20], [120, 140, 120])
    n = 64 * 96
    rays = random_rays(n, lo, hi, seed=5)
    k = np.arange(n)
    special = (k // 64) % 3 == 1            # every third wave: all lanes special
    special |= ((k // 64) % 3 == 2) & (k % 7 == 0)  # the wave after it: a few special lanes among finite rays
    idx = np.nonzero(special)[0]
    pick = verts[rng.integers(0, len(verts), len(idx))]
    rays[idx, 0:3] = pick                                       # origin exactly on box bounds (a vertex lies on the bounds of every node above it)
    kind = rng.integers(0, 4, len(idx))
    dirs = rays[idx, 4:7].copy()
    ax = rng.integers(0, 3, len(idx))
    dirs[np.arange(len(idx)), ax] = np.where(kind == 0, 0.0, np.where(kind == 1, -0.0, np.where(kind == 2, 1e-39, -1e-42))).astype(np.float32)
    two = rng.random(len(idx)) < 0.3                            # two zero components: an axis-parallel ray
    dirs[np.arange(len(idx))[two], (ax[two] + 1) % 3] = 0.0
    rays[idx, 4:7] = dirs
    ro, rh = o.trace(rays), h.trace(rays, count=False)
    assert np.array_equal(ro["prim"], rh["prim"])
    for f in ("t", "b0", "b1"):
        assert np.array_equal(bits(ro[f]), bits(rh[f])), f
    assert (ro["prim"] >= 0).mean() > 0.02
    rays[:, 3] = rng.uniform(1, 600, n).astype(np.float32)
    assert np.array_equal(o.trace(rays, True)["occluded"], h.trace(rays, True, count=False)["occluded"])


@pytest.mark.parametrize("scene", ["cornell", "soup-128", "soup-40-degenerate", "sphere-zoo", "cutout", "mis-plates"])
def test_untested_interior_nodes_change_no_hit_record(gpu_host, orc, scene, monkeypatch):
    """Round 5: the stackless walks of an LDS-resident scene pass over interior nodes whose box test rarely fails (rt_scene_create picks them on synthetic rays; a box contains
    its children's boxes, so an interior node's test decides nothing). Hit records and occlusion answers must be the oracle's bit for bit with and without that - on rays
    inside and outside the scene, axis-parallel rays and rays with zero components (which walk every node) among them."""
    from rustracer_amd.scenes import cornell_box, random_soup
    if scene == "cornell":
        d = cornell_box(32, 32, 1)
        lo, hi = np.float32([-50, -50, -850]), np.float32([600, 600, 600])
    elif scene == "soup-128":
        d = random_soup(126, seed=5, max_prims=4)            # (+ the emitter quad: 128 primitives, the LDS kernels' limit)
        lo, hi = np.float32([-20] * 3), np.float32([120] * 3)
    elif scene == "mis-plates":                              # S3: 2461 nodes - too many for the 256-node LDS kernels, few enough for ONE workgroup's LDS per CU (k_trace<.., MID>: occlusion rays over an LDS copy of the scene, closest hit over bounds + link rows)
        from rustracer_amd.scenes import mis_plates
        d = mis_plates(spp=1)
        bb = np.asarray(d.arrays()[0], np.float32)
        lo, hi = bb.min(0) - np.float32(0.5), bb.max(0) + np.float32(0.5)
    elif scene in ("sphere-zoo", "cutout"):                   # quadrics of every kind / alpha and shadow-alpha masks: the GENERAL kernels walk the same tables
        from test_gpu_sphere import _sphere_zoo
        from test_gpu_alpha import _cutout_scene
        d = _sphere_zoo(8, 1) if scene == "sphere-zoo" else _cutout_scene(8, 1)
        lo, hi = np.float32([-50, -50, -850]), np.float32([600, 600, 600])
    else:
        d = random_soup(40, seed=9, max_prims=1, degenerate=True)   # coincident centroids, a flat half: the leaf fall-backs of the build
        lo, hi = np.float32([-20] * 3), np.float32([120] * 3)
    n = 60000
    rays = random_rays(n, lo, hi, seed=17)
    k = np.arange(n)
    rays[k % 11 == 0, 4] = 0.0                     # zero direction components: those waves take the full tables
    rays[k % 13 == 0, 5] = -0.0
    rays[k % 17 == 0, 4:7] = np.float32([0, 0, 1])  # axis-parallel
    o = orc.OracleScene(d)
    ro = o.trace(rays)
    rays_any = rays.copy(); rays_any[:, 3] = np.random.default_rng(3).uniform(0.1, 150 if scene.startswith("soup") else (float(np.linalg.norm(hi - lo)) if scene == "mis-plates" else 700), n).astype(np.float32)
    ra = o.trace(rays_any, True)
    tested = {}
    for prune in ("1", "0"):
        monkeypatch.setenv("RTX_LDS_PRUNE", prune)
        h = gpu_host.HostScene(d); h.upload(0)
        assert h.lds_resident() if scene != "mis-plates" else (not h.lds_resident() and gpu_host.lib().rtxh_scene_query(h.h, 2) == 1)
        tested[prune] = gpu_host.lib().rtxh_scene_query(h.h, 1)
        rh = h.trace(rays, count=False)
        assert np.array_equal(ro["prim"], rh["prim"]), prune
        for f in ("t",) if scene == "sphere-zoo" else ("t", "b0", "b1"):   # (a quadric's hit record carries t and the primitive, test_gpu_sphere.py)
            assert np.array_equal(bits(ro[f]), bits(rh[f])), (prune, f)
        assert np.array_equal(ra["occluded"], h.trace(rays_any, True, count=False)["occluded"]), prune
    nn = gpu_host.HostScene(d).bvh_sizes()[0]
    assert tested["0"] == nn and tested["1"] <= nn
    if scene in ("cornell", "sphere-zoo", "cutout", "mis-plates"):
        assert tested["1"] < nn   # the Cornell box's walls fill their parents' boxes: some interior tests never pay
    assert (ro["prim"] >= 0).mean() > 0.02


def test_offset_ray_origin_steps_ulps_like_the_reference(gpu_host, orc):
    """offset_ray_origin (geometry/mod.rs:203-220): the device steps a coordinate up or down with ONE fused sequence (next_float_toward, round 5); the reference has
    next_float_up / next_float_down (lib.rs:227-262). Bit-equal on ordinary points and on every special value: zeros of both signs, denormals, the largest floats,
    infinities, NaN, zero offsets (no step), normals of both orientations."""
    rng = np.random.default_rng(11)
    n = 20000
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.1754944e-38, -1.1754944e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan, 1.0, -1.0, 555.0], np.float32)
    p = rng.uniform(-600, 600, (n, 3)).astype(np.float32)
    k = rng.integers(0, n, 3000)
    p[k, rng.integers(0, 3, 3000)] = specials[rng.integers(0, len(specials), 3000)]
    pe = (np.abs(p) * np.float32(4e-7)).astype(np.float32)
    pe[~np.isfinite(pe)] = 0.0
    nn = rng.normal(size=(n, 3)).astype(np.float32)
    nn[rng.integers(0, n, 2000), rng.integers(0, 3, 2000)] = 0.0      # zero offset components: that coordinate is not stepped
    nn[rng.integers(0, n, 500)] = 0.0                                # a zero normal: nothing moves
    nn = (nn / np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), np.float32(1e-20))).astype(np.float32)
    w = rng.normal(size=(n, 3)).astype(np.float32)
    got = gpu_host.offset_ray_origin(p, pe, nn, w)
    # the reference's arithmetic in float32, its two stepping functions from the oracle
    f = np.float32
    an = np.abs(nn)
    d = ((an[:, 0] * pe[:, 0]).astype(f) + (an[:, 1] * pe[:, 1]).astype(f)).astype(f) + (an[:, 2] * pe[:, 2]).astype(f)
    off = (d[:, None] * nn).astype(f)
    dwn = ((w[:, 0] * nn[:, 0]).astype(f) + (w[:, 1] * nn[:, 1]).astype(f)).astype(f) + (w[:, 2] * nn[:, 2]).astype(f)
    off = np.where((dwn < 0)[:, None], -off, off).astype(f)
    with np.errstate(invalid="ignore"):
        po = (p + off).astype(f)
    L = orc.lib()
    want = po.copy()
    for i, c in zip(*np.nonzero(off > 0)):
        want[i, c] = L.orc_next_float_up(float(po[i, c])) if not np.isnan(po[i, c]) else np.float32(np.uint32(po[i, c].view(np.uint32) - np.uint32(1)).view(np.float32))
    for i, c in zip(*np.nonzero(off < 0)):
        want[i, c] = L.orc_next_float_down(float(po[i, c])) if not np.isnan(po[i, c]) else np.float32(np.uint32(po[i, c].view(np.uint32) + np.uint32(1)).view(np.float32))
    fin = ~np.isnan(want)
    assert np.array_equal(bits(got)[fin], bits(want)[fin])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert (off > 0).sum() > 10000 and (off < 0).sum() > 10000 and (off == 0).sum() > 1000


@pytest.mark.parametrize("spp", [1, 16, 64, 1024])
def test_sampler_tables_match_oracle(gpu_host, orc, spp):
    n = 130 if spp <= 64 else 66  # more than one 64-lane block, ragged tail
    sc, pm = gpu_host.sampler_tables(spp, 4, 5000, n)
    for i in (0, 1, 63, 64, n - 1):
        t1, t2 = tables_from_perm(sc[i], pm[i], 4)
        o1, o2, _ = orc.sampler_tables(spp, 4, 1, 5000 + i)
        assert np.array_equal(bits(t1), bits(o1)) and np.array_equal(bits(t2), bits(o2)), (spp, i)
    assert sorted(pm[0, 0].tolist()) == list(range(pm.shape[2]))  # a permutation


def test_sampler_tables_match_golden(gpu_host):
    g = np.load(os.path.join(GOLD, "sampler_tables_keyed.npz"))
    for spp in (16, 64):
        for px in (0, 1, 777):
            sc, pm = gpu_host.sampler_tables(spp, 4, px, 1)
            t1, t2 = tables_from_perm(sc[0], pm[0], 4)
            assert np.array_equal(bits(t1), bits(g[f"t1_{spp}_{px}"])) and np.array_equal(bits(t2), bits(g[f"t2_{spp}_{px}"]))


@pytest.mark.parametrize("spp", [1024, 16384])
def test_sampler_tables_retry_pixels(gpu_host, orc, spp):
    """Pixels whose RNG stream holds a bounded-draw retry (found by the oracle, stored as a fixture): the segmented
    jump-ahead sampler must detect them and fall back to the sequential stream."""
    px = np.load(os.path.join(GOLD, "sampler_retry_pixels.npz"))[f"spp{spp}"]
    assert len(px) >= 3
    for p in px[:3]:
        p = int(p)
        assert len(orc.sampler_retry_scan(spp, 4, p, 1)) == 1  # the fixture still names a retry pixel
        sc, pm = gpu_host.sampler_tables(spp, 4, p - 2, 5)     # two clean neighbours either side
        for i in range(5):
            t1, t2 = tables_from_perm(sc[i], pm[i], 4)
            o1, o2, _ = orc.sampler_tables(spp, 4, 1, p - 2 + i)
            assert np.array_equal(bits(t1), bits(o1)) and np.array_equal(bits(t2), bits(o2)), (spp, p, i)


def test_sampler_segmented_equals_plain_over_a_large_range(gpu_host):
    """The sampler rt_render uses (jump-ahead segments + pipelined shuffle + retry redo) against the single-kernel
    in-order walk of the same stream, over a pixel range that holds retry pixels 8933 (spp 1024)."""
    # (64 ... 1024 spp: the parallel replay of the shuffle, k_sampler_shuffle_par - odd pixel counts and ragged last workgroups included; others: the chain kernel)
    for spp, p0, n in ((1024, 8000, 2000), (256, 0, 5000), (8, 123456, 4097), (2, 7, 300), (512, 100, 1000), (128, 5, 777), (64, 0, 1031), (2048, 3, 40)):
        a = gpu_host.sampler_tables(spp, 4, p0, n)
        b = gpu_host.sampler_tables(spp, 4, p0, n, plain=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), spp


@pytest.mark.parametrize("dims", [2, 5, 8])
def test_sampler_tables_of_other_dimension_counts(gpu_host, orc, dims):
    """ "dimensions" other than 4 (zerotwosequence.rs:58-63; the C ABI takes 2 ... 8): 2 * dims tables per pixel, up to 16 - ADVICE r04: the launch's table ids were
    packed four bits each into 32 bits, so for dims >= 5 some tables were never shuffled and others twice. Segmented == plain == the oracle, both shuffle kernels."""
    for spp, n in ((64, 130), (1024, 40), (16, 300)):   # (64, 1024: the parallel replay; 16: the chain kernel)
        a = gpu_host.sampler_tables(spp, dims, 777, n)
        b = gpu_host.sampler_tables(spp, dims, 777, n, plain=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (dims, spp)
        for t in range(2 * dims):
            assert sorted(a[1][0, t].tolist()) == list(range(a[1].shape[2])), (dims, spp, t)   # every table is a permutation: none skipped, none shuffled twice
        for i in (0, n - 1):
            t1, t2 = tables_from_perm(a[0][i], a[1][i], dims)
            o1, o2, _ = orc.sampler_tables(spp, dims, 1, 777 + i)
            assert np.array_equal(bits(t1), bits(o1)) and np.array_equal(bits(t2), bits(o2)), (dims, spp, i)


@pytest.mark.parametrize("dims", [2, 8])
def test_frame_with_other_dimension_counts_matches_oracle(gpu_host, orc, dims):
    """A whole frame at "dimensions" 2 / 8: the frame's table groups (which tables bounce 0 / 1 / later read) at the extremes."""
    d = _cornell(24, 24, 16)
    d.sampler.dims = dims
    film, st = gpu_host.HostScene(d).render()
    ref, ost = orc.OracleScene(d).render(mode=1)
    assert np.array_equal(film[..., 3], ref[..., 3])
    assert rel_l2(gpu_host.film_to_rgb(film), orc.film_to_rgb(ref)) < L2_GATE
    assert st["camera_rays"] == 24 * 24 * 16


def test_parallel_replay_survives_an_atomic_unit_that_serves_lanes_in_another_order(gpu_host, capfd):
    """k_sampler_shuffle_par reads a writer's rank inside its group off an LDS atomic's return value and CHECKS that the groups came out ascending - lane order
    within one atomic instruction is not architected. RTX_K0_FORCE_RESORT=1 hands every group its ranks reversed: every wave must notice, re-sort, and still produce
    the sequential replay's tables (the fallback was 'never seen', so never run)."""
    import re
    os.environ["RTX_K0_FORCE_RESORT"] = "1"
    os.environ["RTX_K0_REPORT"] = "1"
    try:
        capfd.readouterr()
        a = gpu_host.sampler_tables(1024, 4, 8900, 70)   # (1024 spp uses the parallel replay by default; pixel 8933 holds a retry)
        err = capfd.readouterr().err
        m = re.search(r"k_sampler_shuffle_par: (\d+) wave", err)
        assert m and int(m.group(1)) >= 70 * 8 // 16, err[-300:]   # every workgroup's waves took the re-sort branch
    finally:
        del os.environ["RTX_K0_FORCE_RESORT"]
        del os.environ["RTX_K0_REPORT"]
    b = gpu_host.sampler_tables(1024, 4, 8900, 70, plain=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_light_distribution_matches_oracle_and_golden(gpu_host, orc, gold):
    d = _cornell(8, 8, 1)
    ldh = gpu_host.HostScene(d).light_distribution()
    assert ldh["n_voxels"].tolist() == gold["ld_n_voxels"].tolist()
    m = gold["ld_func"].shape[0]
    assert np.array_equal(bits(ldh["func"][:m]), bits(gold["ld_func"])) and np.array_equal(bits(ldh["cdf"][:m]), bits(gold["ld_cdf"]))
    ldo = orc.OracleScene(d).light_distrib(max_voxels=40000)
    k = ldo["func"].shape[0]
    assert np.array_equal(bits(ldh["func"][:k]), bits(ldo["func"])) and np.array_equal(bits(ldh["cdf"][:k]), bits(ldo["cdf"]))
    assert np.array_equal(bits(ldh["func_int"][:k]), bits(ldo["func_int"]))


# ---------------------------------------------------------------- frames
def test_render_matches_golden_film(gpu_host, gold):
    h = gpu_host.HostScene(_cornell(32, 32, 16))
    film, st = h.render()
    assert np.array_equal(film[..., 3], gold["film_xyzw"][..., 3])  # filter weight sums: exact
    assert rel_l2(gpu_host.film_to_rgb(film), gpu_host.film_to_rgb(gold["film_xyzw"])) < L2_GATE
    assert st["camera_rays"] == int(gold["stats"][0])
    assert abs(int(st["rays_closest"]) - int(gold["stats"][1])) <= 8  # a handful of paths may flip at grazing hits


@pytest.mark.parametrize("res,spp", [((64, 64), 16), ((100, 60), 8), ((33, 17), 4)])
def test_render_matches_oracle(gpu_host, orc, res, spp):
    d = _cornell(res[0], res[1], spp)
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = gpu_host.HostScene(d).render(count_traversal=True)
    assert np.array_equal(fo[..., 3], fh[..., 3])
    err = rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo))
    assert err < L2_GATE, err
    assert np.array_equal(bits(gpu_host.film_to_rgb(fo)), bits(orc.film_to_rgb(fo)))  # the two write_image restatements agree
    # ray and traversal counts follow the oracle's (identical up to a few flipped paths)
    for k in ("rays_closest", "rays_shadow", "nodes_closest", "tris_closest"):
        assert abs(int(sh[k]) - int(so[k])) <= 1e-3 * int(so[k]) + 8, k


def test_rays_nothing_reads_are_not_cast_and_the_film_is_the_same(gpu_host, orc):
    """PathIntegrator::li traces the next ray before it tests the depth (path.rs:100-139) and reads the hit at bounces == max_depth only after a specular
    bounce. A production frame does not cast the others (rt_stats::rays_tail_not_cast, part of rays_closest); a frame that counts the reference's walk casts
    every ray the reference casts. Same film bit for bit, same ray counts - also with a mirror in the scene, where the rays behind a specular bounce at the
    limit are read (emitted light) and therefore cast."""
    for mirror in (False, True):
        d = _cornell(40, 36, 16)
        if mirror:
            d.materials[1] = d.materials[d.mirror(0.9)]  # one wall becomes a mirror
        h = gpu_host.HostScene(d)
        plain, sp = h.render()
        ref, sr = h.render(count_traversal=True)
        assert np.array_equal(bits(plain), bits(ref))
        assert sp["rays_tail_not_cast"] > 0 and sr["rays_tail_not_cast"] == 0
        for k in ("camera_rays", "rays_closest", "rays_shadow", "rays_mis"):
            assert sp[k] == sr[k], k
        fo, so = orc.OracleScene(d).render(mode=1, n_threads=1)
        assert abs(int(sp["rays_closest"]) - int(so["rays_closest"])) <= 8 and rel_l2(gpu_host.film_to_rgb(plain), orc.film_to_rgb(fo)) < L2_GATE


def test_render_is_deterministic_and_reentrant(gpu_host):
    h = gpu_host.HostScene(_cornell(48, 48, 16))
    a, _ = h.render()
    b, _ = h.render()
    assert np.array_equal(bits(a), bits(b))
    h2 = gpu_host.HostScene(_cornell(48, 48, 16))
    c, _ = h2.render()
    assert np.array_equal(bits(a), bits(c))


def test_sharded_render_sums_to_the_full_frame(gpu_host):
    from rustracer_amd.distributed import owned_pixel_mask
    d = _cornell(40, 70, 8)  # 70 rows: 5 tile rows, the last one partial
    h = gpu_host.HostScene(d)
    full, _ = h.render()
    st = h.setup()
    for world in (2, 3, 8):
        acc = np.zeros_like(full)
        for r in range(world):
            part, s = h.render(rank=r, world_size=world)
            m = owned_pixel_mask(st["cropped"], st["sample_bounds"], r, world)
            assert np.all(part[~m] == 0)  # box filter: a rank only touches the pixels it owns
            acc += part
        assert np.array_equal(bits(acc), bits(full))


@pytest.mark.parametrize("kind,params", [(2, (2.0, 2.0, 2.0, 0.0)), (1, (1.5, 1.5, 0, 0)), (3, (2.0, 2.0, 1 / 3, 1 / 3))])
def test_wide_filters(gpu_host, orc, kind, params):
    d = _cornell(40, 40, 8)
    d.film.filter_kind, d.film.filter_params = kind, params
    fo, _ = orc.OracleScene(d).render(mode=1, n_threads=1)
    fh, _ = gpu_host.HostScene(d).render()
    assert np.allclose(fo[..., 3], fh[..., 3], rtol=1e-5, atol=1e-5)  # splat order differs -> not bitwise
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < L2_GATE


def test_crop_window_and_pixel_bounds(gpu_host, orc):
    d = _cornell(64, 48, 4)
    d.film.crop = (0.25, 0.75, 0.5, 1.0)
    d.integrator.pixel_bounds = (20, 40, 26, 44)
    fo, _ = orc.OracleScene(d).render(mode=1)
    fh, _ = gpu_host.HostScene(d).render()
    assert fo.shape == fh.shape == (24, 32, 4)
    assert np.array_equal(fo[..., 3], fh[..., 3]) and fh[..., 3].max() == 4 and fh[..., 3].min() == 0
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < L2_GATE


def test_depth_and_strategy_variants(gpu_host, orc):
    for max_depth, strategy, rr in ((0, "spatial", 1.0), (1, "uniform", 1.0), (8, "spatial", 0.2)):
        d = _cornell(32, 32, 8, max_depth=max_depth)
        d.integrator.light_strategy = strategy
        d.integrator.rr_threshold = rr
        fo, _ = orc.OracleScene(d).render(mode=1)
        fh, _ = gpu_host.HostScene(d).render()
        assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < L2_GATE, (max_depth, strategy)


# ---------------------------------------------------------------- full-size properties (BASELINE configs[1] geometry)
def test_full_size_properties(gpu_host):
    """1024x1024 Cornell at reduced spp: properties that do not need the oracle."""
    d = _cornell(1024, 1024, 16)
    h = gpu_host.HostScene(d)
    film, st = h.render()
    assert film.shape == (1024, 1024, 4)
    # box filter, radius 0.5: every pixel receives its own spp unit weights; a sample whose x + u rounds up to the next
    # integer in f32 (u > 1 - 2^-14 at x ~ 1000) also splats onto the neighbour, exactly as film.rs:313-321 does
    assert np.all(film[..., 3] >= 16.0) and np.all(film[..., 3] == np.round(film[..., 3])) and film[..., 3].mean() < 16.01
    assert np.isfinite(film).all() and st["paths_scrubbed"] == 0
    assert st["camera_rays"] == 1024 * 1024 * 16
    rgb = gpu_host.film_to_rgb(film)
    assert rgb.min() >= 0 and 0.05 < rgb.mean() < 1.0
    # linearity in the emitted radiance: scaling L by 2 scales the film by exactly 2 (power-of-two scaling is exact in f32)
    for l in d.lights:
        l.rgb = tuple(2 * c for c in l.rgb)
    film2, _ = gpu_host.HostScene(d).render()
    assert np.array_equal(bits(film2[..., :3]), bits(film[..., :3] * np.float32(2)))
    # 4-way sharding reproduces the frame bit for bit
    acc = np.zeros_like(film2)
    h2 = gpu_host.HostScene(d)
    for r in range(4):
        acc += h2.render(rank=r, world_size=4)[0]
    # pixels that received only their own samples are bit-identical; the few that also caught a neighbour's
    # edge splat are summed per rank in XYZ (as the reference sums per film tile) and agree to rounding
    plain = film2[..., 3] == 16.0
    assert plain.mean() > 0.99
    assert np.array_equal(bits(acc[plain]), bits(film2[plain]))
    assert np.allclose(acc, film2, rtol=1e-5, atol=1e-6)


def test_screen_window_frame_matches_oracle(gpu_host, orc):
    d = _cornell(48, 36, 8)
    d.camera.screen_window = (-0.8, 1.1, -0.9, 0.6)   # off-centre window (camera.rs:98-107)
    fo, _ = orc.OracleScene(d).render(mode=1, n_threads=1)
    fh, _ = gpu_host.HostScene(d).render()
    assert np.array_equal(fo[..., 3], fh[..., 3])
    assert rel_l2(gpu_host.film_to_rgb(fh), orc.film_to_rgb(fo)) < 1e-3


@pytest.mark.parametrize("chunks_per_device", [1, 3])
def test_multi_device_entry_point_on_one_gpu(gpu_host, chunks_per_device):
    """rt_multi_render with two workers that both sit on device 0 (each with its own scene replica, host thread and stream): chunks of tile rows
    from the shared queue, rows gathered by peer copies and summed on the first device - the same film as one rt_render call."""
    d = _cornell(72, 70, 8)  # 70 rows: 5 tile rows, the last one partial
    h = gpu_host.HostScene(d)
    full, sf = h.render()
    film, total, per = h.render_multi([0, 0], chunks_per_device=chunks_per_device)
    assert np.array_equal(film[..., 3], full[..., 3])
    # a pixel two chunks touched (a sample exactly on a row edge) sums in chunk order instead of sample order
    assert np.allclose(film, full, rtol=1e-6, atol=0) and (film != full).mean() < 1e-3
    assert total["camera_rays"] == sf["camera_rays"] == sum(p["camera_rays"] for p in per)
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert total[k] == sf[k]
    assert all(p["camera_rays"] > 0 for p in per) or chunks_per_device > 1  # both workers rendered (a dynamic queue may starve one on a tiny frame)
    film2, _, _ = h.render_multi([0, 0], chunks_per_device=chunks_per_device)  # replicas and light distributions are reused
    assert np.array_equal(film2, film)


def test_multi_device_wide_filter_rows_are_summed(gpu_host):
    d = _cornell(48, 64, 4)
    d.film.filter_kind, d.film.filter_params = 2, (2.0, 2.0, 2.0, 0.0)  # gaussian radius 2: every tile row splats into its neighbours
    h = gpu_host.HostScene(d)
    full, _ = h.render()
    film, _, _ = h.render_multi([0, 0, 0], chunks_per_device=1)
    assert np.allclose(film, full, rtol=2e-5, atol=1e-6)



def test_parallel_replay_of_the_shuffle_is_exact():
    """k_sampler_shuffle_par replays a Fisher-Yates chain with one wave instead of one lane. By default it serves 1024-spp frames only (rtx_hip.hip,
    launch_sampler_tables); RTX_K0_PARALLEL=1 turns it on for 64 <= spp <= 1024. The knob is read once per process, so the sampler parity tests of this file run
    again in a child process with it set: tables bit-equal to the oracle's (spp 64 and 1024), to the single-kernel in-order walk (64 ... 1024 spp, odd pixel
    counts, ragged workgroups) and on the retry pixels - and every wave's groups came out sorted (the re-sort branch has its own test above)."""
    import subprocess, sys
    env = dict(os.environ, RTX_K0_PARALLEL="1", RTX_K0_REPORT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-s", "-k",
                        "sampler_tables_match_oracle or sampler_tables_retry_pixels or sampler_segmented_equals_plain"], capture_output=True, text=True, env=env, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout and "k_sampler_shuffle_par: 0 wave(s) re-sorted" in r.stderr + r.stdout  # the parallel kernel ran, its groups came out sorted


def test_random_scenes_of_every_size_class_walk_like_the_oracle(gpu_host, orc, monkeypatch):
    """scripts/fuzz_lds_walks.py, sixteen scenes of it: soups of 3 ... 1400 triangles with random leaf sizes (LDS-resident, with quadrics, mid-size; some degenerate), 40 000
    rays each with zero / denormal components, axis-parallel rays and origins on vertices among them - closest hit bit for bit, occlusion answers equal."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_lds_walks", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_lds_walks.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    monkeypatch.setattr("sys.argv", ["fuzz_lds_walks.py", "16", "5"])
    assert m.main() == 0
