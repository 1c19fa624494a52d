"""Randomised FILM parity against the oracle: random scenes over every size class of the trace kernels (a few dozen triangles in LDS, a mid-size tree, a mesh that lives in
HBM), every material of rc/material/* with constant, image (trilinear and EWA, three wrap modes), checkerboard, uv, fbm, scale and mix textures in its slots, bump maps, mix
materials, alpha masks, per-vertex normals / uv, analytic quadrics, and light sets drawn from emitting quads (one- and two-sided), emitting spheres and disks, point, distant
and environment lights (constant or image); random depth, light strategy, pixel filter, lens, crop window / pixel bounds / screen window. One small frame (48 x 36 x 8 spp) per scene: filter weights equal, film inside
1e-3 (a scene above it is judged again at 64, 512 and 4096 spp: one path that ends a bounce early - the radiance-only reciprocals, DESIGN §2 - or one firefly of a mirror-sharp lobe
can be 1e-3 of so small a frame, and weighs 1 / spp; one still above it is compared with what the ORACLE's frame does when the camera moves by one ulp: a scene whose own frame moves as
much is chaotic, not wrong), ray
counts inside 2e-3; every fourth scene is also rendered as three film shards whose sum has to be the whole frame (every eighth by two threads of rt_multi_render too), and of
every fourth 20 000 random rays' hit records, occlusion answers and the light-distribution tables are compared bit for bit. GPU box, repo root:
    python scripts/fuzz_shading.py [n_scenes=60] [seed=1]
The oracle is the checker here, as in tests/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rustracer_amd import host  # noqa: E402
from rustracer_amd import scene_desc as sd  # noqa: E402
from rustracer_amd.scenes.procedural import displaced_sphere, icosphere, box_mesh  # noqa: E402
from oracle import orc  # noqa: E402  (the checker, as in tests/conftest.py)


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def rgb(rng, lo=0.05, hi=0.95):
    return tuple(float(x) for x in rng.uniform(lo, hi, 3))


def spectrum_tex(s, rng, mips, depth=0):
    """A random spectrum texture id (or a constant tuple)."""
    k = int(rng.integers(0, 9 if depth < 1 else 5))   # (the device evaluates texture graphs two levels deep, rtx_dev_shading.h; deeper ones are refused by name)
    m = (float(rng.uniform(0.5, 6.0)), float(rng.uniform(0.5, 6.0)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)))
    if k <= 1:
        return rgb(rng)
    if k == 2:
        return s.image_tex(int(rng.choice(mips)), *m)
    if k == 3:
        return s.checker_tex(rgb(rng), rgb(rng), *m, aa="closedform" if rng.random() < 0.6 else "none")
    if k == 4:
        return s.uv_tex(*m)
    if k == 5:
        return s.scale_tex(s._t(spectrum_tex(s, rng, mips, depth + 1)), s._t(spectrum_tex(s, rng, mips, depth + 1)))
    if k == 6:
        return s.mix_tex(s._t(spectrum_tex(s, rng, mips, depth + 1)), s._t(spectrum_tex(s, rng, mips, depth + 1)), s._t(float(rng.uniform(0.1, 0.9))))
    if k == 7:
        return s.checker_tex(s._t(spectrum_tex(s, rng, mips, depth + 1)), rgb(rng), *m)
    return s.image_tex(int(rng.choice(mips)), *m)


def rough(rng, remap=True):
    """0.001 only where the material remaps it (roughness_to_alpha(0.001) = 0.054). Taken as alpha itself it is a lobe 1e-3 rad wide, whose D lives where sin^2 = 1 - cos^2 of
    the half vector is ~1e-6: ONE ulp in a direction that reaches the vertex - the cosf / sinf of a cosine-sampled bounce, glibc's on the host and ocml's on the device - moves a
    sample's D by ~10 %. The first vertex still agrees (rtx_dev_bsdf.h sharp_lobe); from the second on the frames differ like noise (2e-3 at 64 spp, 9e-4 at 512:
    scripts/exp_sharp_lobes.py) and the reference itself would differ from itself on another libm."""
    return float(rng.choice([0.0, 0.001, 0.02, 0.1, 0.3, 0.8] if remap else [0.0, 0.02, 0.05, 0.1, 0.3, 0.8]))


def material(s, rng, mips):
    k = int(rng.integers(0, 10))
    T = lambda: spectrum_tex(s, rng, mips)
    remap = bool(rng.random() < 0.7)
    if k == 0:
        m = s.matte(T(), sigma=float(rng.choice([0.0, 0.0, 20.0, 60.0])))
    elif k == 1:
        m = s.plastic(T(), T(), max(rough(rng, remap), 0.001 if remap else 0.02), remap)
    elif k == 2:
        m = s.metal(roughness=max(rough(rng, remap), 0.001 if remap else 0.02), remap=remap) if rng.random() < 0.5 else s.metal(eta=rgb(rng, 0.2, 3.0), k=rgb(rng, 1.0, 4.0), roughness=0.05, urough=max(rough(rng, remap), 0.001 if remap else 0.02), vrough=max(rough(rng, remap), 0.001 if remap else 0.02), remap=remap)
    elif k == 3:
        m = s.mirror(T())
    elif k == 4:
        # (never ONE zero roughness: the lobe is then a microfacet one with alpha = 0 on an axis, every value of it is NaN, and the reference's assert!(ld.y() >= 0.0) /
        # assert!(beta.y() >= 0.0) (path.rs:166, 179) end its render - there is nothing to be equal to; oracle and device scrub such samples, not the same ones)
        r = rough(rng, remap); r2 = r if rng.random() < 0.5 else rough(rng, remap)
        if (r == 0.0) != (r2 == 0.0):
            r2 = r
        m = s.glass(T(), T(), float(rng.uniform(1.1, 2.2)), r, r2, remap)
    elif k == 5:
        m = s.uber(T(), T(), rgb(rng, 0.0, 0.5) if rng.random() < 0.5 else 0.0, rgb(rng, 0.0, 0.5) if rng.random() < 0.5 else 0.0, max(rough(rng, remap), 0.001 if remap else 0.02), index=float(rng.uniform(1.1, 2.0)),
                   opacity=rgb(rng, 0.3, 1.0) if rng.random() < 0.4 else 1.0, remap=remap)
    elif k == 6:
        m = s.substrate(T(), T(), max(rough(rng, remap), 0.001 if remap else 0.02), max(rough(rng, remap), 0.001 if remap else 0.02), remap)
    elif k == 7:
        m = s.translucent(T(), T(), rgb(rng), rgb(rng), max(rough(rng, remap), 0.001 if remap else 0.02), remap)
    elif k == 8:
        m = s.disney(color=T(), metallic=float(rng.uniform(0, 1)), eta=float(rng.uniform(1.2, 2.0)), roughness=float(rng.uniform(0.05, 1.0)), speculartint=float(rng.uniform(0, 1)),
                     anisotropic=float(rng.uniform(0, 0.9)), sheen=float(rng.uniform(0, 1)), sheentint=float(rng.uniform(0, 1)), clearcoat=float(rng.choice([0.0, 0.5, 1.0])),
                     clearcoatgloss=float(rng.uniform(0, 1)), spectrans=float(rng.choice([0.0, 0.0, 0.6])), thin=bool(rng.random() < 0.25), flatness=float(rng.uniform(0, 1)),
                     difftrans=float(rng.uniform(0, 2)))
    else:
        m = s.matte(T())
    if rng.random() < 0.2:
        bump = s.fbm_tex(0.5, int(rng.integers(2, 6))) if rng.random() < 0.5 else s.image_tex(int(rng.choice(mips)), 3.0, 3.0)
        s.set_bump(m, s.scale_tex(bump, s.const_tex(float(rng.uniform(0.01, 0.1)))))
    return m


def make_scene(rng):
    s = sd.SceneDesc()
    mips = []
    for _ in range(int(rng.integers(1, 4))):
        h, w = [(16, 16), (8, 32), (5, 7), (32, 32)][int(rng.integers(0, 4))]
        img = rng.uniform(0.02, 1.0, (h, w, 3)).astype(np.float32) ** 2
        mips.append(s.add_mip(img, trilinear=bool(rng.random() < 0.5), max_aniso=float(rng.choice([2.0, 8.0])), wrap=int(rng.integers(0, 3))))
    mats = [material(s, rng, mips) for _ in range(int(rng.integers(2, 9)))]
    if rng.random() < 0.4:
        mats.append(s.mix(mats[0], mats[1], spectrum_tex(s, rng, mips) if rng.random() < 0.5 else float(rng.uniform(0.2, 0.8))))
    pick = lambda: mats[int(rng.integers(0, len(mats)))]
    size = int(rng.integers(0, 3))   # 0: tens of triangles, 1: about a thousand, 2: tens of thousands
    R = 4.0
    # a room (floor, back wall, sometimes all six sides) and things in it
    closed = rng.random() < 0.35
    quv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    def quad(p, mat, **kw):
        s.add_mesh(np.float32(p), [[0, 1, 2], [0, 2, 3]], mat, UV=np.float32(quv), **kw)
    quad([(-R, 0, -R), (-R, 0, R), (R, 0, R), (R, 0, -R)], pick())
    quad([(-R, 0, R), (-R, 2 * R, R), (R, 2 * R, R), (R, 0, R)], pick())
    if closed:
        quad([(-R, 0, -R), (-R, 2 * R, -R), (-R, 2 * R, R), (-R, 0, R)], pick())
        quad([(R, 0, R), (R, 2 * R, R), (R, 2 * R, -R), (R, 0, -R)], pick())
        quad([(-R, 2 * R, R), (-R, 2 * R, -R), (R, 2 * R, -R), (R, 2 * R, R)], pick())
    n_things = int(rng.integers(1, 4))
    for k in range(n_things):
        c = (float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.8, 2.5)), float(rng.uniform(-1.5, 2.5)))
        if size == 0:
            P, F = box_mesh(np.float32(c) - rng.uniform(0.3, 0.8, 3).astype(np.float32), np.float32(c) + rng.uniform(0.3, 0.8, 3).astype(np.float32))[:2]
            s.add_mesh(P, F, pick())
        elif size == 1:
            P, F, N, UV = displaced_sphere(16, 12, c, float(rng.uniform(0.5, 1.0)), 0.2, int(rng.integers(1 << 20)))
            s.add_mesh(P, F, pick(), N=N if rng.random() < 0.6 else None, UV=UV, reverse_orientation=bool(rng.random() < 0.15))
        else:
            P, F, N, UV = displaced_sphere(96, 64, c, float(rng.uniform(0.5, 1.0)), 0.25, int(rng.integers(1 << 20)))
            s.add_mesh(P, F, pick(), N=N if rng.random() < 0.6 else None, UV=UV)
    if rng.random() < 0.3:      # a masked card
        if rng.random() < 0.5:
            tex = s.checker_tex(1.0, 0.0, 4.0, 4.0, aa="none")   # (a float checkerboard: the reference's file format has none, api.rs:1201-1216; its Texture trait has)
        else:
            holes = (rng.random((8, 8, 1)) < 0.6).astype(np.float32).repeat(3, axis=2)
            tex = s.image_tex(s.add_mip(holes, trilinear=True, wrap=int(rng.integers(0, 3))), float(rng.uniform(1.0, 4.0)), float(rng.uniform(1.0, 4.0)))
        c = rng.uniform(-2, 2, 3); c[1] = abs(c[1]) + 0.5
        s.add_mesh(np.float32([c + (-0.8, -0.5, 0), c + (0.8, -0.5, 0), c + (0.8, 0.5, 0.3), c + (-0.8, 0.5, 0.3)]), [[0, 1, 2], [0, 2, 3]], pick(), UV=np.float32(quv), alpha=tex,
                   shadow_alpha=tex if rng.random() < 0.6 else None)
    if rng.random() < 0.4:      # analytic quadrics among the triangles
        s.add_sphere((float(rng.uniform(-2, 2)), float(rng.uniform(0.5, 2)), float(rng.uniform(-1, 2))), float(rng.uniform(0.3, 0.8)), pick(),
                     phi_max=float(rng.choice([360.0, 360.0, 250.0])))
        if rng.random() < 0.5:
            m = np.eye(4, dtype=np.float32); m[:3, 3] = (float(rng.uniform(-2, 2)), 0.0, float(rng.uniform(-1, 2))); m[:3, :3] = np.float32([[1, 0, 0], [0, 0, 1], [0, -1, 0]])
            s.add_cylinder(m, float(rng.uniform(0.2, 0.5)), pick(), z_min=0.0, z_max=float(rng.uniform(0.5, 2.0)))
    # lights: at least one
    n_l = 0
    if rng.random() < 0.7:
        for _ in range(int(rng.integers(1, 4))):
            c = np.float32([rng.uniform(-2.5, 2.5), rng.uniform(3.0, 2 * R - 0.3), rng.uniform(-2.5, 2.5)]); e = float(rng.uniform(0.3, 1.2))
            quad([c + (-e, 0, -e), c + (e, 0, -e), c + (e, 0, e), c + (-e, 0, e)], mats[0], emission=rgb(rng, 5.0, 40.0), two_sided=bool(rng.random() < 0.3))
            n_l += 1
    if rng.random() < 0.3:
        s.add_sphere((float(rng.uniform(-2, 2)), float(rng.uniform(3.0, 5.0)), float(rng.uniform(-2, 2))), float(rng.uniform(0.15, 0.5)), mats[0], emission=rgb(rng, 10.0, 60.0)); n_l += 1
    if rng.random() < 0.2:
        m = np.eye(4, dtype=np.float32); m[:3, :3] = np.float32([[1, 0, 0], [0, 0, 1], [0, -1, 0]]); m[:3, 3] = (float(rng.uniform(-2, 2)), float(rng.uniform(4.0, 6.0)), float(rng.uniform(-2, 2)))
        s.add_disk(m, float(rng.uniform(0.3, 0.9)), mats[0], emission=rgb(rng, 10.0, 40.0), two_sided=True); n_l += 1
    if rng.random() < 0.3:
        s.point_light((float(rng.uniform(-3, 3)), float(rng.uniform(2, 6)), float(rng.uniform(-3, 3))), rgb(rng, 10.0, 80.0)); n_l += 1
    if rng.random() < 0.25 and not closed:
        s.distant_light((float(rng.uniform(-1, 1)), 3.0, float(rng.uniform(-2, 0))), (0, 0, 0), rgb(rng, 0.5, 3.0)); n_l += 1
    if (rng.random() < 0.4 or n_l == 0) and not closed:
        env = np.full((4, 8, 3), 0.5, np.float32) if rng.random() < 0.4 else (rng.uniform(0.0, 1.0, (8, 16, 3)).astype(np.float32) ** 3 * 3.0)
        s.infinite_light(s.add_mip(env, trilinear=False, max_aniso=8.0)); n_l += 1
    if n_l == 0:
        s.point_light((0.0, 5.0, 0.0), (60.0, 60.0, 60.0))
    s.camera.pos, s.camera.look, s.camera.fov = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(1.5, 4.0)), -R + 0.3), (0.0, 1.5, 1.0), float(rng.uniform(40.0, 75.0))
    if rng.random() < 0.2:
        s.camera.lens_radius, s.camera.focal_distance = 0.1, 4.0
    s.film.xres, s.film.yres = 48, 36
    if rng.random() < 0.5:
        s.film.filter_kind = int(rng.integers(0, 4))
        s.film.filter_params = {0: (0.5, 0.5, 0, 0), 1: (1.5, 1.5, 0, 0), 2: (1.5, 1.5, 2.0, 0), 3: (2.0, 2.0, 1 / 3, 1 / 3)}[s.film.filter_kind]
    if rng.random() < 0.15:     # a crop window (film.rs:48-64) / the integrator's pixel bounds (path.rs:53, renderer.rs:103) / a screen window
        x0, y0 = float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.0, 0.4))
        s.film.crop = (x0, x0 + float(rng.uniform(0.3, 0.6)), y0, y0 + float(rng.uniform(0.3, 0.6)))
    elif rng.random() < 0.1:
        s.integrator.pixel_bounds = (int(rng.integers(0, 20)), int(rng.integers(24, 48)), int(rng.integers(0, 12)), int(rng.integers(16, 36)))
    elif rng.random() < 0.1:
        s.camera.screen_window = (-1.2, 0.9, -0.7, 0.8)
    s.sampler.spp = 8
    s.integrator.max_depth = int(rng.choice([1, 2, 3, 5, 5, 8]))
    s.integrator.light_strategy = str(rng.choice(["spatial", "spatial", "uniform", "power"]))
    s.max_prims_per_node = int(rng.choice([1, 2, 4, 4, 8]))
    return s, size, closed, n_l


def main():
    orc.build()
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rng_rays = np.random.default_rng(10 ** 6 + (int(sys.argv[2]) if len(sys.argv) > 2 else 1))   # (the scenes are a function of the seed alone: `bisect` replays them)
    bad_total, worst, t0 = 0, 0.0, time.time()
    for k in range(n_scenes):
        d, size, closed, n_l = make_scene(rng)
        note = ""
        try:
            o, h = orc.OracleScene(d), host.HostScene(d)
            fo, so = o.render(mode=1)
            fh, sh = h.render()
            err = rel_l2(host.film_to_rgb(fh), orc.film_to_rgb(fo))
            for spp in (64, 512, 4096):   # a frame above the gate is judged again with more samples: what one early-ended path or one firefly weighs falls as 1 / spp, what a wrong value weighs does not
                if err > 1e-3:
                    d.sampler.spp = spp
                    fo, so = orc.OracleScene(d).render(mode=1); fh, sh = host.HostScene(d).render()
                    note += f" (at {spp // 8} spp {err:.1e})"
                    err = rel_l2(host.film_to_rgb(fh), orc.film_to_rgb(fo))
            if err > 1e-3:
                # Is the scene itself that sensitive? The ORACLE against the oracle with the camera moved by 1e-7 of its position (about one ulp): sharp lobes over
                # high-frequency bump maps and many bounces turn one ulp in a direction - the device's sincos polynomial against glibc's (rtx_dev_bsdf.h:11-15) - into
                # another path within two bounces. A frame that differs from the oracle's by no more than the oracle's own frame moves is not judged.
                import copy
                dd = copy.deepcopy(d); pc = np.float32(dd.camera.pos); dd.camera.pos = tuple(float(x) for x in (pc * np.float32(1.0 + 1e-7)))
                self_err = rel_l2(orc.film_to_rgb(orc.OracleScene(dd).render(mode=1)[0]), orc.film_to_rgb(fo))
                note += f" (the oracle's own frame moves {self_err:.1e} under a 1e-7 step of the camera)"
                if self_err > 0.5 * err:
                    note += " ILL-CONDITIONED, not judged"; err = 0.0
            box = d.film.filter_kind == sd.FILTER_BOX and max(d.film.filter_params[:2]) <= 0.5   # (a wider filter splats into neighbouring pixels, in another order: not bitwise)
            a, b = host.film_to_rgb(fh).astype(np.float64), orc.film_to_rgb(fo).astype(np.float64)
            off = float((np.abs(a - b).max(axis=-1) > 1e-3 * np.maximum(np.abs(b).max(axis=-1), b.mean())).mean())
            checks = {"weights": not (np.array_equal(fo[..., 3], fh[..., 3]) if box else np.allclose(fo[..., 3], fh[..., 3], rtol=1e-4, atol=1e-3)), "finite": not np.isfinite(host.film_to_rgb(fh)).all(), "film": err > 1e-3,
                      "scrubbed": abs(int(sh["paths_scrubbed"]) - int(so["scrubbed"])) > 2}
            for f in ("rays_closest", "rays_shadow", "rays_mis"):
                checks[f] = abs(int(sh[f]) - int(so[f])) > 2e-3 * int(so[f]) + 16
            # the frame in film shards (one process per GPU in production: rt_shard's 4-row bands) sums to the whole frame - bit for bit under the box filter, in another
            # order of the splats under a wider one
            if k % 4 == 1:   # hit records of random rays (primitive, the bits of t / b0 / b1, visit counts) and occlusion answers; the light-distribution tables bit for bit
                rays = np.zeros((20000, 8), np.float32)
                rays[:, :3] = rng_rays.uniform((-4, 0, -4), (4, 8, 4), (20000, 3)); dd_ = rng_rays.normal(size=(20000, 3)); rays[:, 4:7] = dd_ / np.linalg.norm(dd_, axis=1, keepdims=True); rays[:, 3] = np.inf
                ro = o.trace(rays)
                for count in (True, False):
                    rh = h.trace(rays, count=count)
                    checks[f"hits{int(count)}"] = not (np.array_equal(ro["prim"], rh["prim"]) and all(np.array_equal(ro[f].view(np.uint32), rh[f].view(np.uint32)) for f in (("t",) if d.spheres else ("t", "b0", "b1")))
                                                       and (not count or (ro["nodes"], ro["tris"]) == (rh["nodes"], rh["tris"])))
                rays[:, 3] = rng_rays.uniform(0.2, 9.0, 20000).astype(np.float32)
                ao = o.trace(rays, True)["occluded"]
                checks["occlusion"] = not (np.array_equal(ao, h.trace(rays, True)["occluded"]) and np.array_equal(ao, h.trace(rays, True, count=False)["occluded"]))
                if d.integrator.light_strategy == "spatial" and len(d.lights) > 1:
                    lo, lh = o.light_distrib(max_voxels=400), h.light_distribution()
                    kk = lo["func"].shape[0]
                    # Bit for bit where the contribution of a light is arithmetic alone (triangle emitters, point and distant lights, a constant environment); a light that is
                    # sampled through sin / cos - an emitting sphere or disk (cone / concentric-disk sampling), an image-mapped environment (sin(theta) of its pdf) - carries the last-bit
                    # difference of the device's sine and cosine against glibc's (rtx_dev_bsdf.h:11-15) into its column: those within 4 ulp
                    if lo["n_voxels"].tolist() != lh["n_voxels"].tolist():
                        note += f" [voxels {lo['n_voxels'].tolist()} / {lh['n_voxels'].tolist()}]"; checks["light tables"] = True
                    else:
                        trig = np.array([(l.kind == sd.LIGHT_INFINITE and d.mipmaps[l.mip].data.std() > 0) or getattr(l, "sphere", -1) >= 0 for l in d.lights])
                        fa, fb = lo["func"], lh["func"][:kk]
                        exact = np.array_equal(fa[:, ~trig].view(np.uint32), fb[:, ~trig].view(np.uint32))
                        near = np.allclose(fa[:, trig], fb[:, trig], rtol=5e-7, atol=0.0)
                        cdf_ok = np.array_equal(lo["cdf"].view(np.uint32), lh["cdf"][:kk].view(np.uint32)) if not trig.any() else np.allclose(lo["cdf"], lh["cdf"][:kk], rtol=0.0, atol=5e-7)
                        checks["light tables"] = not (exact and near and cdf_ok)
                        if checks["light tables"]:
                            cols = np.flatnonzero((fa.view(np.uint32) != fb.view(np.uint32)).any(axis=0))
                            note += f" [light columns that differ {cols.tolist()}: kinds {[d.lights[int(c)].kind for c in cols]}, through sin / cos {trig[cols].tolist()}; exact part {exact}, near part {near}, cdf {cdf_ok}]"
            if k % 4 == 0:
                d.sampler.spp = 8
                hs = host.HostScene(d); full = hs.render()[0]; acc = np.zeros_like(full)
                for r in range(3):
                    acc += hs.render(rank=r, world_size=3)[0]
                checks["shards"] = not (np.array_equal(acc, full) if box else np.allclose(acc, full, rtol=2e-5, atol=1e-5))
                if k % 8 == 0:   # ... and by two host threads of this process on chunks of tile rows (rt_multi_render; both on device 0 here)
                    multi = hs.render_multi([0, 0], chunks_per_device=int(rng_rays.integers(1, 4)))[0]
                    checks["multi"] = not (np.array_equal(multi, full) if box else np.allclose(multi, full, rtol=2e-5, atol=1e-5))
            bad = sum(int(v) for v in checks.values())
            if bad:
                kinds = sorted({m.kind for m in d.materials})
                note += " FAILED: " + ", ".join(k_ for k_, v in checks.items() if v) + f" [pixels off {off:.3f}; material kinds {kinds}; bumps {sum(1 for m in d.materials if getattr(m, 'bump', -1) is not None and getattr(m, 'bump', -1) >= 0)}; scrubbed {sh['paths_scrubbed']} / {so['scrubbed']}; rays " + " ".join(f"{sh[f]}/{so[f]}" for f in ("rays_closest", "rays_shadow", "rays_mis")) + "]"
            worst = max(worst, err)
        except (host.BackendError, RuntimeError) as e:
            print(f"    refused: {e}")
            err, bad, so = float("nan"), 1, {"rays_closest": 0}
        bad_total += bad
        print(f"scene {k:3d}: size class {size}, {d.n_tris:6d} triangles, {len(d.spheres)} quadrics, {len(d.materials):2d} materials, {len(d.lights):2d} lights{' (closed room)' if closed else ''}, depth {d.integrator.max_depth}, "
              f"{d.integrator.light_strategy:7s}: film rel-L2 {err:.1e}{note}, {so['rays_closest']} rays: {bad} mismatches", flush=True)
    print(f"{n_scenes} scenes, {bad_total} mismatches, worst film rel-L2 {worst:.1e}, {time.time() - t0:.0f} s")
    return 1 if bad_total else 0


def bisect(seed, index, spp=64):
    """python scripts/fuzz_shading.py bisect <seed> <scene>: the scene again, then with each material in turn replaced by a grey matte one (and each bump map removed): which
    material carries the difference."""
    import copy
    orc.build()
    rng = np.random.default_rng(seed)
    for _ in range(index + 1):
        d, size, closed, n_l = make_scene(rng)
    d.sampler.spp = spp

    def err_of(dd):
        fo, so = orc.OracleScene(dd).render(mode=1); fh, sh = host.HostScene(dd).render()
        a, b = host.film_to_rgb(fh).astype(np.float64), orc.film_to_rgb(fo).astype(np.float64)
        off_mask = np.abs(a - b).max(axis=-1) > 1e-3 * np.maximum(np.abs(b).max(axis=-1), b.mean()); off = float(off_mask.mean())
        worst = [(int(y), int(x), float(a[y, x].max()), float(b[y, x].max())) for y, x in zip(*np.unravel_index(np.argsort(-np.abs(a - b).max(axis=-1).ravel())[:3], off_mask.shape))]
        return rel_l2(a, b), off, int(sh["paths_scrubbed"]), int(so["scrubbed"]), worst
    def tex_str(i, depth=0):
        t = d.textures[i]
        name = {sd.TEX_CONST: "const", sd.TEX_SCALE: "scale", sd.TEX_MIX: "mix", sd.TEX_IMAGE: "image", sd.TEX_CHECKER: "checker", sd.TEX_UV: "uv", sd.TEX_FBM: "fbm"}[t.kind]
        if t.kind == sd.TEX_CONST:
            return f"const{tuple(round(x, 3) for x in t.value)}"
        if t.kind == sd.TEX_IMAGE:
            mi = d.mipmaps[t.mip]
            return f"image(mip {t.mip}: {mi.data.shape[1]}x{mi.data.shape[0]} trilinear={mi.trilinear} aniso={mi.max_aniso} wrap={mi.wrap}, map {tuple(round(x, 2) for x in t.mapping)})"
        if t.kind in (sd.TEX_SCALE, sd.TEX_MIX, sd.TEX_CHECKER):
            return f"{name}({tex_str(t.tex1, depth + 1)}, {tex_str(t.tex2, depth + 1)}" + (f", amount {tex_str(t.amount, depth + 1)}" if t.kind == sd.TEX_MIX else (f", map {tuple(round(x, 2) for x in t.mapping)} aa {t.amount}" if t.kind == sd.TEX_CHECKER else "")) + ")"
        if t.kind == sd.TEX_FBM:
            return f"fbm(omega {t.value[0]}, octaves {t.amount})"
        return f"{name}(map {tuple(round(x, 2) for x in t.mapping)})"
    for i, m in enumerate(d.materials):
        if m.bump >= 0:
            print(f"material {i} bump: {tex_str(m.bump)}")
    print(f"film filter {d.film.filter_kind} {d.film.filter_params}, lens {d.camera.lens_radius}, depth {d.integrator.max_depth}, strategy {d.integrator.light_strategy}, max prims {d.max_prims_per_node}")
    print("scene as it is:", err_of(d))
    kd, zero = d.const_tex(0.5), d.const_tex(0.0)
    for i, m in enumerate(d.materials):
        dd = copy.deepcopy(d)
        dd.materials[i] = sd.Material(sd.MAT_MATTE, {"kd": kd, "sigma": zero})
        print(f"material {i} (kind {m.kind}, bump {m.bump}, remap {m.remap_roughness}, " + ", ".join(f"{k}={d.textures[v].kind if k not in ('m1', 'm2') else v}:{tuple(round(x, 3) for x in d.textures[v].value) if k not in ('m1', 'm2') and d.textures[v].kind == sd.TEX_CONST else ''}" for k, v in m.params.items()) + ") as grey matte:", err_of(dd))
        if m.bump >= 0:
            dd = copy.deepcopy(d); dd.materials[i].bump = -1
            print(f"material {i} without its bump map:", err_of(dd))


if __name__ == "__main__":
    sys.exit(bisect(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 64) if len(sys.argv) > 1 and sys.argv[1] == "bisect" else main())
