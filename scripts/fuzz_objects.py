"""Randomised parity of two-level scenes whose objects hold quadrics and alpha-masked meshes (rc/primitive.rs:79-118 TransformedPrimitive over an object's aggregate;
rtx_kernels.h object_walk_general / instance_intersect) against the oracle: random objects (a soup of triangles, some of it masked by a checker texture; partial spheres,
cylinders, disks under transforms of their own; objects of one quadric only), random instances (rotation x non-uniform scale x sometimes a mirror x translation), sometimes
top-level quadrics and a masked top-level mesh beside them. Closest hit (primitive id and the bits of t, b0, b1; node / primitive visit counts of the counting kernels) and
occlusion, production and counting kernels; a small frame of every scene (40 x 30 x 4 spp under an emitting quad, half of them inside a constant environment): filter weights
equal, film inside 1e-3, ray counts inside 2e-3. GPU box, repo root:
    python scripts/fuzz_objects.py [n_scenes=40] [seed=1]          (FUZZ_RICH=1: the objects wear the material zoo of scripts/fuzz_shading.py)
Prints one line per scene and the number of mismatching values in all (expected: 0). The oracle is the checker here, as in tests/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from rustracer_amd import host  # noqa: E402
from rustracer_amd.scene_desc import SceneDesc  # noqa: E402
from oracle import orc  # noqa: E402  (the checker, as in tests/conftest.py)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


OFF = set(os.environ.get("FUZZ_OFF", "").split(","))   # bisecting switches: inst_mirror, inst_scale, q_mirror, q_xform, partial, kind0, kind1, kind2, masks


def rotation(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def affine(rng, spread, scale=(0.4, 1.8), mirror=0.25, who="inst"):
    m = np.eye(4)
    s = rng.uniform(scale[0], scale[1], 3)
    if rng.random() < 0.3 or who + "_scale" in OFF:
        s[:] = s[0]                                        # a similarity now and then
    if rng.random() < mirror and who + "_mirror" not in OFF:
        s[int(rng.integers(0, 3))] *= -1.0
    m[:3, :3] = rotation(rng) @ np.diag(s)
    m[:3, 3] = rng.uniform(-spread, spread, 3)
    return m.astype(np.float32)


def soup(rng, n, extent):
    c = rng.uniform(-extent, extent, (n, 1, 3))
    P = (c + rng.normal(scale=0.25 * extent, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    return P, np.arange(3 * n, dtype=np.int32).reshape(n, 3)


def quadric(rng, mats, extent):
    kind = int(rng.integers(0, 3))
    while f"kind{kind}" in OFF:
        kind = (kind + 1) % 3
    o2w = affine(rng, extent, scale=(0.6, 1.5), mirror=0.15, who="q")
    if rng.random() < 0.25 or "q_xform" in OFF:
        o2w = np.eye(4, dtype=np.float32); o2w[:3, 3] = rng.uniform(-extent, extent, 3)
    r = float(rng.uniform(0.1, 0.5) * extent)
    q = dict(kind=kind, o2w=o2w, radius=r, material=mats[int(rng.integers(0, len(mats)))])
    if kind == 0:
        if rng.random() < 0.5:
            a, b = sorted(rng.uniform(-r, r, 2)); q.update(z_min=float(a), z_max=float(b))
        if rng.random() < 0.5:
            q.update(phi_max=float(rng.uniform(60.0, 330.0)))
    elif kind == 2:
        a, b = sorted(rng.uniform(-extent, extent, 2)); q.update(z_min=float(a), z_max=float(b) + 0.05)
        if rng.random() < 0.4:
            q.update(phi_max=float(rng.uniform(60.0, 330.0)))
    else:                                                  # a disk: z_min = height, z_max = inner radius
        q.update(z_min=float(rng.uniform(-0.5, 0.5) * extent), z_max=float(r * rng.uniform(0.0, 0.6)) if rng.random() < 0.5 else 0.0)
        if rng.random() < 0.4:
            q.update(phi_max=float(rng.uniform(60.0, 330.0)))
    if "partial" in OFF:
        for k in ("phi_max",) + (("z_min", "z_max") if kind == 0 else ()):
            q.pop(k, None)
        if kind == 1:
            q["z_max"] = 0.0
    return q


def make_scene(rng):
    s = SceneDesc()
    mats = [s.matte((0.6, 0.5, 0.4)), s.plastic((0.2, 0.3, 0.6), (0.3, 0.3, 0.3), 0.1), s.glass(index=1.5)]
    if os.environ.get("FUZZ_RICH"):   # the material zoo of scripts/fuzz_shading.py (every material, texture class, bump maps) on the objects' surfaces: three of them beside the matte one
        import fuzz_shading
        mips = [s.add_mip(rng.uniform(0.02, 1.0, (16, 16, 3)).astype(np.float32) ** 2, trilinear=bool(rng.random() < 0.5), wrap=int(rng.integers(0, 3)))]
        mats = [mats[0]] + [fuzz_shading.material(s, rng, mips) for _ in range(2)]
    img = np.zeros((8, 8, 3), np.float32); img[::2, ::2] = 1.0; img[1::2, 1::2] = 1.0
    mask = s.image_tex(s.add_mip(img, trilinear=True), su=float(rng.uniform(1.0, 5.0)), sv=float(rng.uniform(1.0, 5.0)))
    what = []
    n_obj = int(rng.integers(1, 5))
    objects = []
    for _ in range(n_obj):
        form = int(rng.integers(0, 5))
        if "masks" in OFF and form in (2, 3):
            form -= 2
        # 0 plain triangles, 1 + quadrics, 2 + masked mesh, 3 both, 4 one quadric alone
        meshes, quads = [], None
        if form != 4:
            P, F = soup(rng, int(rng.choice([1, 2, 7, 40, 300])), 1.0)
            meshes.append(dict(P=P, idx=F, material=mats[int(rng.integers(0, 3))]))
        if form in (2, 3):
            P, F = soup(rng, int(rng.integers(1, 30)), 1.0)
            UV = rng.uniform(0.0, 1.0, (len(P), 2)).astype(np.float32)
            meshes.append(dict(P=P, idx=F, material=mats[0], UV=UV, alpha=mask, shadow_alpha=mask if rng.random() < 0.7 else None))
        if form in (1, 3):
            quads = [quadric(rng, mats, 1.0) for _ in range(int(rng.integers(1, 5)))]
        if form == 4:
            quads = [quadric(rng, mats, 1.0)]
        if meshes and meshes[-1].get("shadow_alpha", 0) is None:
            del meshes[-1]["shadow_alpha"]
        objects.append(s.add_object(meshes, quadrics=quads))
        what.append("tqmbQ"[form])
    n_inst = int(rng.choice([1, 3, 12, 60, 400]))
    spread = 2.0 + 0.6 * n_inst ** (1.0 / 3.0)
    for _ in range(n_inst):
        s.add_instance(objects[int(rng.integers(0, n_obj))], affine(rng, spread))
    top = int(rng.integers(0, 4))        # 0 nothing beside the instances, 1 a floor, 2 + top-level quadrics, 3 + a masked top-level mesh
    if top >= 1:
        e = spread + 2.0
        s.add_quad((-e, -e, -e), (-e, -e, e), (e, -e, e), (e, -e, -e), mats[0])
    if top >= 2:
        s.add_sphere(tuple(rng.uniform(-spread, spread, 3)), float(rng.uniform(0.3, 1.2)), mats[2])
        s.add_cylinder(affine(rng, spread, mirror=0.0), float(rng.uniform(0.2, 0.6)), mats[1], z_min=0.0, z_max=float(rng.uniform(0.5, 2.0)))
    if top >= 3:
        P, F = soup(rng, int(rng.integers(1, 20)), spread)
        s.add_mesh(P, F, mats[0], UV=rng.uniform(0.0, 1.0, (len(P), 2)).astype(np.float32), alpha=mask)
    # what a frame needs: an emitting quad above the scene (listed light), sometimes a constant environment, a camera outside
    e = spread + 1.0
    s.add_quad((-1, e, -1), (1, e, -1), (1, e, 1), (-1, e, 1), mats[0], emission=(40.0, 38.0, 35.0))
    if rng.random() < 0.5:
        s.infinite_light(s.add_mip(np.full((4, 8, 3), 0.3, np.float32)))
    s.camera.pos, s.camera.look, s.camera.fov = (0.3 * spread, 0.5 * spread, -2.2 * spread - 3.0), (0.0, 0.0, 0.0), 45.0
    s.film.xres, s.film.yres = 40, 30
    s.sampler.spp = 4
    s.integrator.max_depth = 4
    return s, "".join(what), n_inst, top, spread


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def rays_for(rng, spread, n):
    org = rng.uniform(-spread - 2.0, spread + 2.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = np.arange(n)
    half = k % 2 == 0                                      # half of the rays aim at the middle of the scene
    d[half] = (rng.uniform(-0.6 * spread, 0.6 * spread, (int(half.sum()), 3)) - org[half]).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[k % 11 == 0, rng.integers(0, 3)] = 0.0
    d[k % 13 == 0, rng.integers(0, 3)] = -0.0
    d[k % 17 == 0] = np.float32([0, 0, 1])
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = org; rays[:, 3] = np.inf; rays[:, 4:7] = d
    return rays


def main():
    orc.build()
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad_total, t0 = 0, time.time()
    for k in range(n_scenes):
        d, what, n_inst, top, spread = make_scene(rng)
        o, h = orc.OracleScene(d), host.HostScene(d)
        bo, bh = o.bvh(), h.bvh()
        bad = sum(int(not np.array_equal(bo[f], bh[f])) for f in bo)
        rays = rays_for(rng, spread, 40000)
        ro = o.trace(rays)
        for count in (True, False):
            rh = h.trace(rays, count=count)
            bad += int((ro["prim"] != rh["prim"]).sum())
            for f in ("t", "b0", "b1"):
                bad += int((bits(ro[f]) != bits(rh[f])).sum())
            if count:
                bad += int(ro["nodes"] != rh["nodes"]) + int(ro["tris"] != rh["tris"])
        rays[:, 3] = rng.uniform(0.05, 1.5, len(rays)).astype(np.float32) * np.float32(2.0 * spread)
        rays_o = rays.copy()
        ao = o.trace(rays, True)["occluded"]
        for count in (True, False):
            bad += int((ao != h.trace(rays, True, count=count)["occluded"]).sum())
        fo, so = o.render(mode=1)
        try:
            fh, sh = h.render()
            err = rel_l2(host.film_to_rgb(fh), orc.film_to_rgb(fo))
            if err > 1e-3:  # at 4 spp one path that ends a bounce earlier (the radiance-only reciprocals) can be 1e-3 of a 40 x 30 frame: judged at 64 spp
                d.sampler.spp = 64
                o2, h2 = orc.OracleScene(d), host.HostScene(d)
                fo, so = o2.render(mode=1); fh, sh = h2.render()
                print(f"    film rel-L2 {err:.1e} at 4 spp; at 64 spp {rel_l2(host.film_to_rgb(fh), orc.film_to_rgb(fo)):.1e}")
                err = rel_l2(host.film_to_rgb(fh), orc.film_to_rgb(fo))
            film_bad = int(not np.array_equal(fo[..., 3], fh[..., 3])) + int(not np.isfinite(host.film_to_rgb(fh)).all()) + int(err > 1e-3)
            film_bad += sum(int(abs(int(sh[f]) - int(so[f])) > 2e-3 * int(so[f]) + 16) for f in ("rays_closest", "rays_shadow", "rays_mis"))
        except host.BackendError as e:
            print(f"    render refused: {e}")
            err, film_bad = float("nan"), 1
        bad += film_bad
        bad_total += bad
        n_top = len(bo["ordered"])
        if bad and os.environ.get("FUZZ_VERBOSE"):
            rays[:, 3] = np.inf
            rh = h.trace(rays, count=False)
            dp = ro["prim"] != rh["prim"]; dt = (bits(ro["t"]) != bits(rh["t"])) & ~dp; db = ((bits(ro["b0"]) != bits(rh["b0"])) | (bits(ro["b1"]) != bits(rh["b1"]))) & ~dp & ~dt
            print(f"    closest: prim differs {int(dp.sum())}, t only {int(dt.sum())}, b only {int(db.sum())}; occlusion differs {int((ao != h.trace(rays_o, True, count=False)['occluded']).sum())}")
            for i in np.flatnonzero(dp | dt | db)[:4]:
                print(f"    ray {i}: o {rays[i, :3]} d {rays[i, 4:7]} oracle prim {ro['prim'][i]} t {ro['t'][i]:.9g} b {ro['b0'][i]:.6g} {ro['b1'][i]:.6g} | device prim {rh['prim'][i]} t {rh['t'][i]:.9g} b {rh['b0'][i]:.6g} {rh['b1'][i]:.6g}")
        print(f"scene {k:3d}: objects {what:4s} x {n_inst:3d} instances, top level {top}, {n_top:4d} top-level primitives, hits {float((ro['prim'] >= 0).mean()):.2f} "
              f"(inside objects {float((ro['prim'] >= n_top).mean()):.2f}), occluded {float(ao.mean()):.2f}, film rel-L2 {err:.1e}: {bad} mismatches", flush=True)
    print(f"{n_scenes} scenes, {bad_total} mismatching values, {time.time() - t0:.0f} s")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
