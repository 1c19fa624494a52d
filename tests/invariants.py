"""Scenes whose expected image has a CLOSED FORM that does not come from the oracle or the reference (the invariants
that widen the oracle's pin beyond the reference's three known-answer vectors). Used on the oracle by tests/test_invariants_cpu.py and on
the HIP path by tests/test_gpu_invariants.py.

  furnace_scene      a convex Lambertian body of albedo rho inside a constant environment of radiance 1: a ray leaving the body never
                     returns, so the one vertex of every path collects exactly rho (the cosine-weighted integral of the environment) and a
                     camera ray that hits the body sees L = rho at ANY max_depth >= 1, whatever the sampling and the MIS weights.
  furnace_instances_scene  the same furnace with a convex body placed by a TWO-LEVEL instance (rc/primitive.rs:79-118): an object that holds one sphere under a stretch
                     of its own, or an object that holds a closed box of triangles, placed by a rotated, non-uniformly scaled, mirrored instance - affine images of
                     convex bodies are convex, so every body pixel is rho still. A wrong normal transform (inverse transpose, the mirror's flip), a wrong
                     hit point after the two Transform * Ray or a leak at an object's boundary shows as L != rho.
  furnace_box_scene  the camera inside a closed box whose walls emit L_e and reflect rho: every point sees emitters over its whole hemisphere,
                     so each further bounce adds rho times the previous order and L = L_e (1 + rho + ... + rho^max_depth) in every pixel
                     (emission at the first hit, direct light at every vertex with bounces < max_depth, rc/integrator/path.rs:127-165; Russian
                     roulette is unbiased). Exercises twelve one-sided area lights, the emitter-identity test of the MIS ray and the spatial
                     light distribution.
  form_factor_scene  a Lambertian floor under a one-sided square emitter, direct light only (maxdepth 1): the floor's radiance is
                     rho / pi * E(x) with E the irradiance of a polygonal Lambertian emitter, Lambert's contour formula
                     E = L/2 * sum_edges angle(v_i, v_i+1) * dot(n, normalize(v_i x v_i+1)).
  glossy_scene       a rough plastic / metal plate under a square emitter, direct light only: no closed form, but light sampling alone, BSDF
                     sampling alone and their MIS combination are three unbiased estimators of the same image (the oracle's mis_mode hook).
"""
import numpy as np

from rustracer_amd.scene_desc import SceneDesc


def _box(s, lo, hi, material):
    lo, hi = np.float32(lo), np.float32(hi)
    c = [(lo[0] if i & 1 == 0 else hi[0], lo[1] if i & 2 == 0 else hi[1], lo[2] if i & 4 == 0 else hi[2]) for i in range(8)]
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]  # outward-facing
    for q in quads:
        s.add_quad(*(c[i] for i in q), material)


def furnace_scene(rho=0.5, max_depth=5, res=48, spp=64):
    s = SceneDesc()
    s.name = f"furnace rho={rho} depth={max_depth}"
    _box(s, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), s.matte((rho, rho, rho)))
    env = s.add_mip(np.ones((4, 8, 3), np.float32), trilinear=False, max_aniso=8.0)
    s.infinite_light(env)
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (1.9, 1.4, 1.1), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 40.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = max_depth
    return s


def furnace_instances_scene(kind="ball", rho=0.5, max_depth=5, res=64, spp=64):
    s = SceneDesc()
    s.name = f"furnace of an instanced {kind} rho={rho} depth={max_depth}"
    m = s.matte((rho, rho, rho))

    def place(angle, axis, scale, at):
        a = np.float64(axis) / np.linalg.norm(axis); K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
        M = np.eye(4); M[:3, :3] = R @ np.diag(scale); M[:3, 3] = at
        return M.astype(np.float32)
    if kind == "ball":
        stretch = np.eye(4, dtype=np.float32); stretch[1, 1] = 1.5; stretch[:3, 3] = (0.1, -0.05, 0.2)  # the transform inside the object's definition
        o = s.add_object([], quadrics=[dict(kind=0, o2w=stretch, radius=0.3, material=m)])
        s.add_instance(o, place(0.7, (1, 2, 0.5), (1.6, 1.0, -2.0), (0.2, 0.0, 0.1)))                # rotated, non-uniformly scaled, mirrored
    else:
        lo, hi = np.float32((-0.3, -0.25, -0.2)), np.float32((0.3, 0.25, 0.2))
        c = np.float32([(lo[0] if i & 1 == 0 else hi[0], lo[1] if i & 2 == 0 else hi[1], lo[2] if i & 4 == 0 else hi[2]) for i in range(8)])
        quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
        o = s.add_object([dict(P=c, idx=[t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], material=m)])
        s.add_instance(o, place(-1.1, (0.3, 1, 1), (-1.7, 1.2, 1.5), (0.0, 0.1, 0.0)))
    env = s.add_mip(np.ones((4, 8, 3), np.float32), trilinear=False, max_aniso=8.0)
    s.infinite_light(env)
    s.add_quad((40, 40, 40), (40.01, 40, 40), (40.01, 40.01, 40), (40, 40.01, 40), s.matte((0.0,) * 3))  # (the top level must hold a triangle: a speck, far away)
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (1.9, -2.4, 1.1), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 36.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = max_depth
    return s


def furnace_box_scene(rho=0.5, max_depth=5, le=1.0, res=32, spp=64):
    s = SceneDesc()
    s.name = f"furnace box rho={rho} depth={max_depth}"
    m = s.matte((rho, rho, rho))
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    c = [(lo[0] if i & 1 == 0 else hi[0], lo[1] if i & 2 == 0 else hi[1], lo[2] if i & 4 == 0 else hi[2]) for i in range(8)]
    for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:  # wound so that the normals point inward
        s.add_quad(*(c[i] for i in q), m, emission=(le, le, le))
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.1, -0.2, 0.05), (1.0, 0.7, 0.4), (0.0, 0.0, 1.0), 70.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = max_depth
    return s


def furnace_box_expected(rho, max_depth, le=1.0):
    return le * sum(rho ** k for k in range(0, max_depth + 1))


def furnace_body_mask(film_rgb, rho, max_depth):
    """Pixels all of whose samples hit the body. A pixel the environment (radiance 1) touches at all is brighter than the body; silhouette
    pixels are mixtures, so the mask is the set of pixels at least three pixels away from anything as bright as the sky."""
    sky = film_rgb[..., 1] > 0.999
    near = sky.copy()
    for _ in range(3):
        grown = near.copy()
        grown[1:, :] |= near[:-1, :]; grown[:-1, :] |= near[1:, :]; grown[:, 1:] |= near[:, :-1]; grown[:, :-1] |= near[:, 1:]
        near = grown
    return ~near


LIGHT_Z, LIGHT_HALF, LIGHT_L, CAM_Z, FLOOR_RHO = 3.0, 0.5, 10.0, 2.0, 0.6


def form_factor_scene(res=32, spp=256):
    s = SceneDesc()
    s.name = "form factor"
    s.add_quad((-6, -6, 0), (6, -6, 0), (6, 6, 0), (-6, 6, 0), s.matte((FLOOR_RHO,) * 3))
    h = LIGHT_HALF  # emitter above the camera, facing down (normal -z): the camera never sees it
    s.add_quad((-h, -h, LIGHT_Z), (-h, h, LIGHT_Z), (h, h, LIGHT_Z), (h, -h, LIGHT_Z), s.matte((0.0,) * 3), emission=(LIGHT_L,) * 3)
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.0, 0.0, CAM_Z), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = 1
    return s


def _polygon_irradiance(x, n, verts, radiance):
    v = [np.asarray(p, np.float64) - x for p in verts]
    v = [p / np.linalg.norm(p) for p in v]
    e = 0.0
    for a, b in zip(v, v[1:] + v[:1]):
        c = np.cross(a, b)
        e += np.arccos(np.clip(np.dot(a, b), -1, 1)) * np.dot(n, c / np.linalg.norm(c))
    return abs(0.5 * radiance * e)


def form_factor_expected(res):
    """Radiance of the floor seen through each pixel centre. The set-up is symmetric under x -> -x and y -> -y, so the image does not depend on
    the camera's handedness conventions; only |offset from the axis| enters."""
    t = np.tan(np.radians(60.0) / 2)
    img = np.zeros((res, res))
    h = LIGHT_HALF
    verts = [(-h, -h, LIGHT_Z), (-h, h, LIGHT_Z), (h, h, LIGHT_Z), (h, -h, LIGHT_Z)]
    for j in range(res):
        for i in range(res):
            sx, sy = (2 * (i + 0.5) / res - 1) * t * CAM_Z, (2 * (j + 0.5) / res - 1) * t * CAM_Z  # floor point under the pixel centre
            img[j, i] = FLOOR_RHO / np.pi * _polygon_irradiance(np.array([sx, sy, 0.0]), np.array([0.0, 0.0, 1.0]), verts, LIGHT_L)
    return img


def glossy_scene(kind="plastic", roughness=0.25, res=40, spp=256):
    s = SceneDesc()
    s.name = f"glossy {kind}"
    m = s.plastic((0.3, 0.25, 0.2), (0.5, 0.5, 0.5), roughness) if kind == "plastic" else (
        s.metal(roughness=roughness) if kind == "metal" else s.substrate((0.4, 0.3, 0.2), (0.3, 0.3, 0.3), roughness, roughness))
    s.add_quad((-3, -3, 0), (3, -3, 0), (3, 3, 0), (-3, 3, 0), m)
    # a tilted emitter well off the mirror direction for part of the plate and on it for the rest
    s.add_quad((-0.6, 1.4, 0.7), (0.6, 1.4, 0.7), (0.6, 2.0, 1.6), (-0.6, 2.0, 1.6), s.matte((0.0,) * 3), emission=(8.0, 7.0, 6.0), two_sided=True)
    s.camera.pos, s.camera.look, s.camera.up, s.camera.fov = (0.0, -2.2, 1.0), (0.0, 0.3, 0.0), (0.0, 0.0, 1.0), 50.0
    s.film.xres = s.film.yres = res
    s.sampler.spp = spp
    s.integrator.max_depth = 1
    return s
