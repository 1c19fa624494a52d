// rtx_dev_sphere.h — the analytic sphere on the device: rc/shapes/sphere.rs (intersect :71-203, sample :227-244, sample_si :246-308, pdf_wi
// :310-334, area :336-338) over the running-error arithmetic of rc/efloat.rs, Ray::transform (rc/ray.rs:46-71) and the error-carrying transforms of
// rc/transform.rs:175-253. Same operations in the same order as the reference; the CPU restatement the tests compare with is oracle/orc_sphere.h.
// Only the kernels of scenes that hold a sphere (k_trace_big<.., GENERAL>, k_shade<0, true>, k_resolve<true>) reach this code.
#pragma once
#include "rtx_dev_bsdf.h"
#include "rtx_dev_scene.h"

namespace rtx {

struct DSphere {  // Sphere (sphere.rs:15-27); o2w / w2o: object-to-world matrix and its inverse, row-major 4x4
  float o2w[16], w2o[16];
  float radius, z_min, z_max, theta_min, theta_max, phi_max;
  int reverse_orientation, swaps_handedness;
  int kind;                    // 0 sphere, 1 disk (rc/shapes/disk.rs), 2 cylinder (rc/shapes/cylinder.rs)
  float height, inner_radius;  // disk
};

// ---------------------------------------------------------------- EFloat (rc/efloat.rs)
struct EFloat { float v, low, high; };
RT_DEV EFloat ef_new(float v, float err) {  // :15-27
  EFloat r; r.v = v;
  if (err == 0.0f) { r.low = v; r.high = v; } else { r.low = next_float_down(v - err); r.high = next_float_up(v + err); }
  return r;
}
RT_DEV EFloat ef(float v) { EFloat r; r.v = r.low = r.high = v; return r; }
RT_DEV EFloat operator+(EFloat a, EFloat f) { EFloat r; r.v = a.v + f.v; r.low = next_float_down(a.low + f.low); r.high = next_float_up(a.high + f.high); return r; }  // :134-146
RT_DEV EFloat operator-(EFloat a, EFloat f) { EFloat r; r.v = a.v - f.v; r.low = next_float_down(a.low - f.high); r.high = next_float_up(a.high - f.low); return r; }  // :148-160
RT_DEV EFloat operator*(EFloat a, EFloat f) {  // :162-188
  const float p0 = a.low * f.low, p1 = a.high * f.low, p2 = a.low * f.high, p3 = a.high * f.high;
  EFloat r; r.v = a.v * f.v;
  r.low = next_float_down(fminf(fminf(p0, p1), fminf(p2, p3))); r.high = next_float_up(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)));
  return r;
}
RT_DEV EFloat operator/(EFloat a, EFloat f) {  // :190-220
  EFloat r; r.v = a.v / f.v;
  if (f.low < 0.0f && f.high > 0.0f) { r.low = -kInf; r.high = kInf; }
  else {
    const float d0 = a.low / f.low, d1 = a.high / f.low, d2 = a.low / f.high, d3 = a.high / f.high;
    r.low = next_float_down(fminf(fminf(d0, d1), fminf(d2, d3))); r.high = next_float_up(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)));
  }
  return r;
}
RT_DEV bool ef_solve_quadratic(EFloat a, EFloat b, EFloat c, EFloat& t0, EFloat& t1) {  // solve_quadratic, :96-120
  const double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
  if (discrim < 0.0) return false;
  const double root_discrim = sqrt(discrim);
  const EFloat frd = ef_new((float)root_discrim, kMachineEpsilon * (float)root_discrim);
  const EFloat q = b.v < 0.0f ? ef(-0.5f) * (b - frd) : ef(-0.5f) * (b + frd);
  EFloat r0 = q / a, r1 = c / q;
  if (r0.v > r1.v) { const EFloat t = r0; r0 = r1; r1 = t; }
  t0 = r0; t1 = r1;
  return true;
}

// ---------------------------------------------------------------- affine transforms, with the reference's error bounds (transform.rs:175-318)
RT_DEV f3 xf34_point(const float* m, f3 p) {  // Transform * Point3f, transform.rs:264-286: divided by w unless w == 1
  const float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
  const float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
  if (wp == 1.0f) return mk3(xp, yp, zp);
  return mk3(xp, yp, zp) / wp;
}
RT_DEV f3 xf34_vector(const float* m, f3 v) { return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }
RT_DEV f3 xf34_normal(const float* mi, f3 n) {  // transform_normal: the transpose of the inverse
  return mk3(mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z);
}
RT_DEV f3 xf34_abs_sum(const float* m, f3 p) {  // transform_point / transform_vector error term (both add |m[r][3]|: a reference quirk for vectors, kept)
  return mk3(fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]), fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]),
             fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]));
}
RT_DEV f3 xf34_point_with_error(const float* m, f3 p, f3 pe, f3& err) {  // transform_point_with_error, :190-219
  float e[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    e[r] = (gamma_n(3) + 1.0f) * (fabsf(m[4 * r] * pe.x) + fabsf(m[4 * r + 1] * pe.y) + fabsf(m[4 * r + 2] * pe.z)) +
           gamma_n(3) * (fabsf(m[4 * r] * p.x) + fabsf(m[4 * r + 1] * p.y) + fabsf(m[4 * r + 2] * p.z) + fabsf(m[4 * r + 3]));
  err = mk3(e[0], e[1], e[2]);
  return xf34_point(m, p);
}

RT_DEV float sphere_area(const DSphere& s) {
  if (s.kind == 1) return s.phi_max * 0.5f * (s.radius * s.radius - s.inner_radius * s.inner_radius);  // Disk::area, disk.rs:153-155
  if (s.kind == 2) return (s.z_max - s.z_min) * s.radius * s.phi_max;                                 // Cylinder::area, cylinder.rs:252-254
  return s.phi_max * s.radius * (s.z_max - s.z_min);                                                  // Sphere::area, sphere.rs:336-338
}
// the tail the three intersect() share: SurfaceInteraction::new(p_hit, p_error, uv, -ray.d, dpdu, dpdv, ..) then .transform(object_to_world)
// (interaction.rs:107-190; dndu / dndv are zeroed by that transform, so they are not formed)
RT_DEV void quadric_finish(const DSphere& s, f3 p_hit, f3 p_error, f2 uv, f3 d_obj, f3 dpdu, f3 dpdv, SurfaceInteraction* si) {
  f3 n = normalize(cross(dpdu, dpdv));
  if ((s.reverse_orientation != 0) != (s.swaps_handedness != 0)) n = n * -1.0f;
  const f3 wo = normalize(normalize(-d_obj));
  f3 pe_w;
  si->hit.p = xf34_point_with_error(s.o2w, p_hit, p_error, pe_w);
  si->hit.p_error = pe_w;
  si->hit.wo = normalize(normalize(xf34_vector(s.o2w, wo)));
  si->hit.n = normalize(xf34_normal(s.w2o, n));
  si->uv = uv;
  si->dpdu = xf34_vector(s.o2w, dpdu); si->dpdv = xf34_vector(s.o2w, dpdv);
  const f3 sn = normalize(xf34_normal(s.w2o, n));
  si->sh_n = dot(sn, si->hit.n) < 0.0f ? -sn : sn;  // face_forward_n(shading.n, hit.n)
  si->sh_dpdu = si->dpdu; si->sh_dpdv = si->dpdv;
  si->dudx = si->dvdx = si->dudy = si->dvdy = 0.0f; si->dpdx = si->dpdy = mk3(0, 0, 0);
}
// Ray::transform(world_to_object), ray.rs:46-71
RT_DEV void ray_to_object(const DSphere& s, f3 ray_o, f3 ray_d, f3& o, f3& d, f3& o_err, f3& d_err) {
  o = xf34_point(s.w2o, ray_o);
  o_err = gamma_n(3) * xf34_abs_sum(s.w2o, ray_o);
  d = xf34_vector(s.w2o, ray_d);
  d_err = gamma_n(3) * xf34_abs_sum(s.w2o, ray_d);
  const float l2 = len2(d);
  if (l2 > 0.0f) { const float dt = dot(abs3(d), o_err) / l2; o = o + d * dt; }
}
template <bool FILL>
RT_DEV bool disk_intersect(const DSphere& s, f3 ray_o, f3 ray_d, float t_max, float& t_hit, SurfaceInteraction* si) {  // Disk::intersect, disk.rs:65-118
  f3 o, d, o_err, d_err; ray_to_object(s, ray_o, ray_d, o, d, o_err, d_err);
  if (d.z == 0.0f) return false;
  const float ts = (s.height - o.z) / d.z;
  if (ts <= 0.0f || ts > t_max) return false;
  f3 p_hit = o + ts * d;
  const float dist2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
  if (dist2 > s.radius * s.radius || dist2 < s.inner_radius * s.inner_radius) return false;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if (phi > s.phi_max) return false;
  t_hit = ts;
  if (!FILL) return true;
  const float u = phi / s.phi_max;
  const float r_hit = sqrtf(dist2);
  const float one_minus_v = (r_hit - s.inner_radius) / (s.radius - s.inner_radius);
  const f3 dpdu = mk3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  const f3 dpdv = mk3(p_hit.x, p_hit.y, 0.0f) * (s.radius - s.inner_radius) / r_hit;
  p_hit.z = s.height;
  quadric_finish(s, p_hit, mk3(0, 0, 0), mk2(u, 1.0f - one_minus_v), d, dpdu, dpdv, si);
  return true;
}
template <bool FILL>
RT_DEV bool cylinder_intersect(const DSphere& s, f3 ray_o, f3 ray_d, float t_max, float& t_hit, SurfaceInteraction* si) {  // Cylinder::intersect / intersect_p, cylinder.rs:62-250
  f3 o, d, o_err, d_err; ray_to_object(s, ray_o, ray_d, o, d, o_err, d_err);
  const EFloat ox = ef_new(o.x, o_err.x), oy = ef_new(o.y, o_err.y);
  const EFloat dx = ef_new(d.x, d_err.x), dy = ef_new(d.y, d_err.y);
  const EFloat a = dx * dx + dy * dy;
  const EFloat b = ef(2.0f) * (dx * ox + dy * oy);
  const EFloat c = ox * ox + oy * oy - ef(s.radius) * ef(s.radius);
  EFloat t0, t1;
  if (!ef_solve_quadratic(a, b, c, t0, t1)) return false;
  if (t0.high > t_max || t1.low <= 0.0f) return false;
  EFloat ts = t0; bool is_t1 = false;
  if (ts.low <= 0.0f) { ts = t1; is_t1 = true; if (ts.high > t_max) return false; }
  f3 p_hit = o + ts.v * d;
  float hit_rad = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
  p_hit.x *= s.radius / hit_rad; p_hit.y *= s.radius / hit_rad;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if (p_hit.z < s.z_min || p_hit.z > s.z_max || phi > s.phi_max) {
    if (is_t1 || ts.v == t1.v) return false;
    ts = t1;
    if (t1.high > t_max) return false;
    p_hit = o + ts.v * d;
    hit_rad = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
    p_hit.x *= s.radius / hit_rad; p_hit.y *= s.radius / hit_rad;
    phi = atan2f(p_hit.y, p_hit.x);
    if (phi < 0.0f) phi += 2.0f * kPi;
    if (p_hit.z < s.z_min || p_hit.z > s.z_max || phi > s.phi_max) return false;
  }
  t_hit = ts.v;
  if (!FILL) return true;
  const float u = phi / s.phi_max;
  const float v = (p_hit.z - s.z_min) / (s.z_max / s.z_min);  // a division where pbrt subtracts (cylinder.rs:121): kept
  const f3 dpdu = mk3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  const f3 dpdv = mk3(0.0f, 0.0f, s.z_max - s.z_min);
  const f3 p_error = gamma_n(3) * mk3(fabsf(p_hit.x), fabsf(p_hit.y), 0.0f);
  quadric_finish(s, p_hit, p_error, mk2(u, v), d, dpdu, dpdv, si);
  return true;
}

// Sphere::intersect (:71-203). FILL: also the world-space SurfaceInteraction that SurfaceInteraction::new + .transform(object_to_world) leave
// (interaction.rs:107-190; dndu / dndv are zeroed by that transform, so they are not formed).
template <bool FILL>
RT_DEV bool sphere_only_intersect(const DSphere& s, f3 ray_o, f3 ray_d, float t_max, float& t_hit, SurfaceInteraction* si) {
  // Ray::transform(world_to_object), ray.rs:46-71
  f3 o = xf34_point(s.w2o, ray_o);
  const f3 o_err = gamma_n(3) * xf34_abs_sum(s.w2o, ray_o);
  const f3 d = xf34_vector(s.w2o, ray_d);
  const f3 d_err = gamma_n(3) * xf34_abs_sum(s.w2o, ray_d);
  const float l2 = len2(d);
  if (l2 > 0.0f) { const float dt = dot(abs3(d), o_err) / l2; o = o + d * dt; }
  const EFloat ox = ef_new(o.x, o_err.x), oy = ef_new(o.y, o_err.y), oz = ef_new(o.z, o_err.z);
  const EFloat dx = ef_new(d.x, d_err.x), dy = ef_new(d.y, d_err.y), dz = ef_new(d.z, d_err.z);
  const EFloat a = dx * dx + dy * dy + dz * dz;
  const EFloat b = ef(2.0f) * (dx * ox + dy * oy + dz * oz);
  const EFloat c = (ox * ox + oy * oy + oz * oz) - ef(s.radius) * ef(s.radius);
  EFloat t0, t1;
  if (!ef_solve_quadratic(a, b, c, t0, t1)) return false;
  if (t0.high > t_max || t1.low <= 0.0f) return false;
  EFloat ts = t0; bool is_t1 = false;
  if (ts.low <= 0.0f) { ts = t1; is_t1 = true; if (ts.high > t_max) return false; }
  f3 p_hit = o + ts.v * d;
  p_hit = p_hit * (s.radius / len(p_hit));
  if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || phi > s.phi_max) {
    if (is_t1 || ts.v == t1.v) return false;  // `t_shape_hit == t1` compares the values (efloat.rs:122-126)
    if (t1.high > t_max) return false;
    ts = t1;
    p_hit = o + ts.v * d;
    p_hit = p_hit * (s.radius / len(p_hit));
    if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
    phi = atan2f(p_hit.x, p_hit.y);  // arguments swapped on the retry path in the reference (sphere.rs:134) - kept
    if (phi < 0.0f) phi += 2.0f * kPi;
    if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || phi > s.phi_max) return false;
  }
  t_hit = ts.v;
  if (!FILL) return true;
  const float u = phi / s.phi_max;
  const float theta = acosf(clampf(p_hit.z / s.radius, -1.0f, 1.0f));
  const float v = (theta - s.theta_min) / (s.theta_max - s.theta_min);
  const float z_radius = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
  const float inv_z_radius = 1.0f / z_radius;
  const float cos_phi_ = p_hit.x * inv_z_radius, sin_phi_ = p_hit.y * inv_z_radius;
  const f3 dpdu = mk3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  const f3 dpdv = (s.theta_max - s.theta_min) * mk3(p_hit.z * cos_phi_, p_hit.z * sin_phi_, -s.radius * sinf(theta));
  const f3 p_error = gamma_n(5) * abs3(p_hit);
  f3 n = normalize(cross(dpdu, dpdv));  // SurfaceInteraction::new, interaction.rs:107-147
  if ((s.reverse_orientation != 0) != (s.swaps_handedness != 0)) n = n * -1.0f;
  const f3 wo = normalize(normalize(-d));
  f3 pe_w;  // .transform(object_to_world), interaction.rs:156-190
  si->hit.p = xf34_point_with_error(s.o2w, p_hit, p_error, pe_w);
  si->hit.p_error = pe_w;
  si->hit.wo = normalize(normalize(xf34_vector(s.o2w, wo)));
  si->hit.n = normalize(xf34_normal(s.w2o, n));
  si->uv = mk2(u, v);
  si->dpdu = xf34_vector(s.o2w, dpdu); si->dpdv = xf34_vector(s.o2w, dpdv);
  const f3 sn = normalize(xf34_normal(s.w2o, n));
  si->sh_n = dot(sn, si->hit.n) < 0.0f ? -sn : sn;  // face_forward_n(shading.n, hit.n)
  si->sh_dpdu = si->dpdu; si->sh_dpdv = si->dpdv;
  si->dudx = si->dvdx = si->dudy = si->dvdy = 0.0f; si->dpdx = si->dpdy = mk3(0, 0, 0);
  return true;
}
template <bool FILL>
RT_DEV bool sphere_intersect(const DSphere& s, f3 ray_o, f3 ray_d, float t_max, float& t_hit, SurfaceInteraction* si) {  // Shape::intersect of a quadric
  if (s.kind == 1) return disk_intersect<FILL>(s, ray_o, ray_d, t_max, t_hit, si);
  if (s.kind == 2) return cylinder_intersect<FILL>(s, ray_o, ray_d, t_max, t_hit, si);
  return sphere_only_intersect<FILL>(s, ray_o, ray_d, t_max, t_hit, si);
}
// out-of-line entry points: the hit test of the traversal leaf loop, and the interaction of an accepted hit (t_max = infinity: the roots, the clipping
// retry and p_hit do not depend on t_max, which only rejects - the accepted test's decisions are reproduced)
RT_DEVN bool sphere_test(const DSphere& s, f3 o, f3 d, float t_max, float& t_hit) { return sphere_intersect<false>(s, o, d, t_max, t_hit, nullptr); }
// the same inline: for the kernels whose only large callee it would be (RT_GEN_NO_MASKS) - inside the kernel it falls under the kernel's register bound
RT_DEV bool sphere_test_inl(const DSphere& s, f3 o, f3 d, float t_max, float& t_hit) { return sphere_intersect<false>(s, o, d, t_max, t_hit, nullptr); }
RT_DEVN bool sphere_fill_interaction(const DSphere& s, f3 o, f3 d, SurfaceInteraction& si) { float t; return sphere_intersect<true>(s, o, d, kInf, t, &si); }

struct SpherePoint { f3 p, p_error, n; };
RT_DEV SpherePoint sphere_sample(const DSphere& s, f2 u, float& pdf) {  // Sphere::sample, :227-244
  const float z = 1.0f - 2.0f * u.x;  // uniform_sample_sphere, sampling/mod.rs:14-20
  const float r = sqrtf(fmaxf(1.0f - z * z, 0.0f));
  const float phi = 2.0f * kPi * u.y;
  float sn_phi, cs_phi; sincosf(phi, &sn_phi, &cs_phi);
  f3 p_obj = mk3(0, 0, 0) + s.radius * mk3(r * cs_phi, r * sn_phi, z);
  SpherePoint it;
  it.n = normalize(xf34_normal(s.w2o, p_obj));
  p_obj = p_obj * s.radius / sqrtf(distance_squared(p_obj, mk3(0, 0, 0)));
  const f3 p_obj_error = gamma_n(5) * abs3(p_obj);
  it.p = xf34_point_with_error(s.o2w, p_obj, p_obj_error, it.p_error);
  pdf = 1.0f / sphere_area(s);
  return it;
}
// Shape::sample of a disk / cylinder (disk.rs:136-151, cylinder.rs:256-278) and the trait's default sample_si over it (shapes/mod.rs:39-53)
RT_DEV SpherePoint quadric_default_sample_si(const DSphere& s, const Interaction& ref, f2 u, float& pdf_out) {
  SpherePoint it;
  if (s.kind == 1) {
    const f2 pd = concentric_sample_disk(u);
    const f3 p_obj = mk3(pd.x * s.radius, pd.y * s.radius, s.height);
    it.n = normalize(xf34_normal(s.w2o, mk3(0.0f, 0.0f, 1.0f)));
    if (s.reverse_orientation) it.n = -it.n;
    it.p = xf34_point_with_error(s.o2w, p_obj, mk3(0, 0, 0), it.p_error);
  } else {
    const float z = lerpf(u.x, s.z_min, s.z_max);
    const float phi = u.y * s.phi_max;
    float sn_phi, cs_phi; sincosf(phi, &sn_phi, &cs_phi);
    f3 p_obj = mk3(s.radius * cs_phi, s.radius * sn_phi, z);
    f3 n = normalize(xf34_normal(s.w2o, mk3(p_obj.x, p_obj.y, 0.0f)));
    if (s.reverse_orientation) n = n * -1.0f;
    const float hit_rad = sqrtf(p_obj.x * p_obj.x + p_obj.y * p_obj.y);
    p_obj.x *= s.radius / hit_rad; p_obj.y *= s.radius / hit_rad;
    const f3 p_obj_error = gamma_n(3) * mk3(fabsf(p_obj.x), fabsf(p_obj.y), 0.0f);
    it.p = xf34_point_with_error(s.o2w, p_obj, p_obj_error, it.p_error);
    it.n = n;
  }
  float pdf = 1.0f / sphere_area(s);
  f3 wi = it.p - ref.p;
  if (len2(wi) == 0.0f) pdf = 0.0f;
  else { wi = normalize(wi); pdf *= distance_squared(ref.p, it.p) / fabsf(dot(it.n, -wi)); if (isinf(pdf)) pdf = 0.0f; }
  pdf_out = pdf;
  return it;
}
// The branch of Sphere::sample_si for a reference point OUTSIDE the sphere (sphere.rs:264-308): uniform sampling of the cone the sphere subtends.
RT_DEV SpherePoint sphere_cone_sample_si(const DSphere& s, f3 p_center, const Interaction& ref, f2 u, float& pdf_out) {
  const f3 wc = normalize(p_center - ref.p); f3 wc_x, wc_y;
  coordinate_system(wc, wc_x, wc_y);
  const float sin_theta_max_2 = s.radius * s.radius / distance_squared(ref.p, p_center);
  const float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_max_2));
  const float cos_theta_ = (1.0f - u.x) + u.x * cos_theta_max;
  const float sin_theta_ = sqrtf(fmaxf(0.0f, 1.0f - cos_theta_ * cos_theta_));
  const float phi = u.y * 2.0f * kPi;
  const float dc = sqrtf(distance_squared(ref.p, p_center));
  const float ds = dc * cos_theta_ - sqrtf(fmaxf(0.0f, s.radius * s.radius - dc * dc * sin_theta_ * sin_theta_));
  const float cos_alpha = (dc * dc + s.radius * s.radius - ds * ds) / (2.0f * dc * s.radius);
  const float sin_alpha = sqrtf(fmaxf(0.0f, 1.0f - cos_alpha * cos_alpha));
  float sn_phi, cs_phi; sincosf(phi, &sn_phi, &cs_phi);
  const f3 n_world = sin_alpha * cs_phi * (-wc_x) + sin_alpha * sn_phi * (-wc_y) + cos_alpha * (-wc);  // spherical_direction_vec, geometry/mod.rs:117-126
  SpherePoint it;
  it.p = p_center + s.radius * mk3(n_world.x, n_world.y, n_world.z);
  it.p_error = gamma_n(5) * abs3(it.p);
  it.n = n_world;
  if (s.reverse_orientation) it.n = it.n * -1.0f;
  pdf_out = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));  // uniform cone pdf
  return it;
}
// ... and of Sphere::pdf_wi (sphere.rs:325-333): the cone's uniform density, whatever wi is
RT_DEV float sphere_cone_pdf_wi(const DSphere& s, f3 p_center, const Interaction& ref) {
  const float sin_theta_max_2 = s.radius * s.radius / distance_squared(ref.p, p_center);
  const float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_max_2));
  return 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));  // uniform_cone_pdf, sampling/mod.rs:54-56
}
RT_DEVN SpherePoint sphere_sample_si(const DSphere& s, const Interaction& ref, f2 u, float& pdf_out) {  // Sphere::sample_si, :246-308
  if (s.kind != 0) return quadric_default_sample_si(s, ref, u, pdf_out);
  const f3 p_center = xf34_point(s.o2w, mk3(0, 0, 0));
  const f3 p_origin = offset_ray_origin(ref.p, ref.p_error, ref.n, p_center - ref.p);
  if (distance_squared(p_origin, p_center) <= s.radius * s.radius) {
    float pdf; SpherePoint intr = sphere_sample(s, u, pdf);
    f3 wi = intr.p - ref.p;
    if (len2(wi) == 0.0f) pdf = 0.0f;
    else { wi = normalize(wi); pdf *= distance_squared(ref.p, intr.p) / fabsf(dot(intr.n, -wi)); }
    if (isinf(pdf)) pdf = 0.0f;
    pdf_out = pdf;
    return intr;
  }
  return sphere_cone_sample_si(s, p_center, ref, u, pdf_out);
}
RT_DEVN float sphere_pdf_wi(const DSphere& s, const Interaction& ref, f3 wi) {  // Sphere::pdf_wi, :310-334; disk / cylinder: the trait default, shapes/mod.rs:59-68
  const f3 p_center = xf34_point(s.o2w, mk3(0, 0, 0));
  const f3 p_origin = offset_ray_origin(ref.p, ref.p_error, ref.n, p_center - ref.p);
  if (s.kind != 0 || distance_squared(p_origin, p_center) <= s.radius * s.radius) {
    const f3 ro = offset_ray_origin(ref.p, ref.p_error, ref.n, wi);  // Interaction::spawn_ray
    SurfaceInteraction li; float t;
    if (!sphere_intersect<true>(s, ro, wi, kInf, t, &li)) return 0.0f;
    return distance_squared(ref.p, li.hit.p) / (fabsf(dot(li.hit.n, -wi)) * sphere_area(s));
  }
  return sphere_cone_pdf_wi(s, p_center, ref);
}

}  // namespace rtx
