// ORACLE — test infrastructure only (see orc_math.h). The analytic sphere of the reference: rc/shapes/sphere.rs (intersect :71-203,
// object / world bounds :205-225, sample :227-244, sample_si :246-308, pdf_wi :310-334, area :336-338) over the running-error arithmetic of
// rc/efloat.rs, Ray::transform (rc/ray.rs:46-71) and the error-carrying transforms of rc/transform.rs:175-253.
#pragma once
#include "orc_math.h"

namespace orc {

// ---------------------------------------------------------------- EFloat (rc/efloat.rs): a value with a conservative interval
struct EFloat {
  float v, low, high;
  float lower_bound() const { return low; }
  float upper_bound() const { return high; }
};
inline EFloat ef_new(float v, float err) {  // :15-27
  if (err == 0.0f) return {v, v, v};
  return {v, next_float_down(v - err), next_float_up(v + err)};
}
inline EFloat ef(float v) { return {v, v, v}; }  // From<f32>, :222-226
inline EFloat operator+(EFloat a, EFloat f) { return {a.v + f.v, next_float_down(a.low + f.low), next_float_up(a.high + f.high)}; }  // :134-146
inline EFloat operator-(EFloat a, EFloat f) { return {a.v - f.v, next_float_down(a.low - f.high), next_float_up(a.high - f.low)}; }  // :148-160
inline EFloat operator*(EFloat a, EFloat f) {  // :162-188
  float p0 = a.low * f.low, p1 = a.high * f.low, p2 = a.low * f.high, p3 = a.high * f.high;
  return {a.v * f.v, next_float_down(fminf(fminf(p0, p1), fminf(p2, p3))), next_float_up(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)))};
}
inline EFloat operator/(EFloat a, EFloat f) {  // :190-220
  float lo, hi;
  if (f.low < 0.0f && f.high > 0.0f) { lo = -kInf; hi = kInf; }
  else {
    float d0 = a.low / f.low, d1 = a.high / f.low, d2 = a.low / f.high, d3 = a.high / f.high;
    lo = next_float_down(fminf(fminf(d0, d1), fminf(d2, d3)));
    hi = next_float_up(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)));
  }
  return {a.v / f.v, lo, hi};
}
inline EFloat operator*(float s, EFloat f) { return ef(s) * f; }
inline EFloat ef_sqrt(EFloat a) { return {sqrtf(a.v), next_float_down(sqrtf(a.low)), next_float_up(sqrtf(a.high))}; }  // EFloat::sqrt, :39-47 (not on the sphere's path; replayed by the reference's property tests)
inline EFloat ef_abs(EFloat a) {  // EFloat::abs, :49-70
  if (a.low >= 0.0f) return a;
  if (a.high <= 0.0f) return {-a.v, -a.high, -a.low};
  return {fabsf(a.v), 0.0f, fmaxf(-a.low, a.high)};
}
inline bool ef_solve_quadratic(EFloat a, EFloat b, EFloat c, EFloat* t0, EFloat* t1) {  // solve_quadratic, :96-120
  double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
  if (discrim < 0.0) return false;
  double root_discrim = sqrt(discrim);
  const float kMachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5f;  // lib.rs MACHINE_EPSILON
  EFloat float_root_discrim = ef_new((float)root_discrim, kMachineEpsilon * (float)root_discrim);
  EFloat q = b.v < 0.0f ? -0.5f * (b - float_root_discrim) : -0.5f * (b + float_root_discrim);
  EFloat r0 = q / a, r1 = c / q;
  if (r0.v > r1.v) { EFloat t = r0; r0 = r1; r1 = t; }
  *t0 = r0; *t1 = r1;
  return true;
}

// ---------------------------------------------------------------- transforms with error bounds (rc/transform.rs:175-253)
inline V3 xf_point_err(const M44& m, V3 p, V3* err) {  // transform_point, :175-188
  float x = p.x, y = p.y, z = p.z;
  float xs = fabsf(m.m[0][0] * x) + fabsf(m.m[0][1] * y) + fabsf(m.m[0][2] * z) + fabsf(m.m[0][3]);
  float ys = fabsf(m.m[1][0] * x) + fabsf(m.m[1][1] * y) + fabsf(m.m[1][2] * z) + fabsf(m.m[1][3]);
  float zs = fabsf(m.m[2][0] * x) + fabsf(m.m[2][1] * y) + fabsf(m.m[2][2] * z) + fabsf(m.m[2][3]);
  *err = gamma_n(3) * v3(xs, ys, zs);
  return xf_point(m, p);
}
inline V3 xf_vector_err(const M44& m, V3 v, V3* err) {  // transform_vector, :223-242 (the translation column enters the bound: a reference quirk, kept)
  float x = v.x, y = v.y, z = v.z;
  float xs = fabsf(m.m[0][0] * x) + fabsf(m.m[0][1] * y) + fabsf(m.m[0][2] * z) + fabsf(m.m[0][3]);
  float ys = fabsf(m.m[1][0] * x) + fabsf(m.m[1][1] * y) + fabsf(m.m[1][2] * z) + fabsf(m.m[1][3]);
  float zs = fabsf(m.m[2][0] * x) + fabsf(m.m[2][1] * y) + fabsf(m.m[2][2] * z) + fabsf(m.m[2][3]);
  *err = gamma_n(3) * v3(xs, ys, zs);
  return xf_vector(m, v);
}
inline V3 xf_point_with_error(const M44& m, V3 p, V3 pe, V3* err) {  // transform_point_with_error, :190-219
  float x = p.x, y = p.y, z = p.z;
  float e[3];
  for (int r = 0; r < 3; ++r)
    e[r] = (gamma_n(3) + 1.0f) * (fabsf(m.m[r][0] * pe.x) + fabsf(m.m[r][1] * pe.y) + fabsf(m.m[r][2] * pe.z)) +
           gamma_n(3) * (fabsf(m.m[r][0] * x) + fabsf(m.m[r][1] * y) + fabsf(m.m[r][2] * z) + fabsf(m.m[r][3]));
  *err = v3(e[0], e[1], e[2]);
  return xf_point(m, p);
}

// ---------------------------------------------------------------- Sphere (and, by `kind`, the two other quadrics of the reference: Disk, Cylinder)
enum { QUADRIC_SPHERE = 0, QUADRIC_DISK = 1, QUADRIC_CYLINDER = 2 };
struct Sphere {
  Transform o2w;  // object_to_world (m) / world_to_object (m_inv)
  float radius, z_min, z_max, theta_min, theta_max, phi_max;
  bool reverse_orientation, swaps_handedness;
  int kind = QUADRIC_SPHERE;
  float height = 0, inner_radius = 0;  // Disk (rc/shapes/disk.rs:13-22)
};
inline Sphere sphere_new(const Transform& o2w, float radius, float z_min, float z_max, float phi_max, bool reverse_orientation) {  // Sphere::new, :29-51
  Sphere s;
  s.o2w = o2w; s.radius = radius;
  s.z_min = clamp_t(fminf(z_min, z_max), -radius, radius);
  s.z_max = clamp_t(fmaxf(z_min, z_max), -radius, radius);
  s.theta_min = acosf(clamp_t(fminf(z_min, z_max) / radius, -1.0f, 1.0f));
  s.theta_max = acosf(clamp_t(fmaxf(z_min, z_max) / radius, -1.0f, 1.0f));
  s.phi_max = to_radians(clamp_t(phi_max, 0.0f, 360.0f));
  s.reverse_orientation = reverse_orientation; s.swaps_handedness = xf_swaps_handedness(o2w.m);
  return s;
}
inline float sphere_area(const Sphere& s) { return s.phi_max * s.radius * (s.z_max - s.z_min); }  // :336-338
inline B3 sphere_world_bounds(const Sphere& s) {  // :205-225: the 8 corners of the object box, in this order
  const V3 lo = v3(-s.radius, -s.radius, s.z_min), hi = v3(s.radius, s.radius, s.z_max);
  B3 b = b3_empty();
  const V3 c[8] = {v3(lo.x, lo.y, lo.z), v3(hi.x, lo.y, lo.z), v3(lo.x, hi.y, lo.z), v3(lo.x, lo.y, hi.z),
                   v3(hi.x, hi.y, lo.z), v3(hi.x, lo.y, hi.z), v3(lo.x, hi.y, hi.z), v3(hi.x, hi.y, hi.z)};
  for (int i = 0; i < 8; ++i) b = b3_union_p(b, xf_point(s.o2w.m, c[i]));
  return b;
}

struct SphereHit {  // what SurfaceInteraction::new + .transform(object_to_world) leave (interaction.rs:107-190); dndu = dndv = 0 after the transform
  float t; V3 p, p_error, wo, n; P2 uv; V3 dpdu, dpdv, sh_n;
};
// Sphere::intersect (:71-203). `fill`: also build the world-space interaction (the hit test alone decides with the same operations).
inline bool sphere_intersect(const Sphere& s, const Ray& ray, bool fill, SphereHit* out) {
  // Ray::transform(world_to_object), ray.rs:46-71
  V3 o_err, d_err;
  V3 o = xf_point_err(s.o2w.m_inv, ray.o, &o_err);
  V3 d = xf_vector_err(s.o2w.m_inv, ray.d, &d_err);
  float l2 = length_squared(d);
  if (l2 > 0.0f) { float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
  const float t_max = ray.t_max;
  EFloat ox = ef_new(o.x, o_err.x), oy = ef_new(o.y, o_err.y), oz = ef_new(o.z, o_err.z);
  EFloat dx = ef_new(d.x, d_err.x), dy = ef_new(d.y, d_err.y), dz = ef_new(d.z, d_err.z);
  EFloat a = dx * dx + dy * dy + dz * dz;
  EFloat b = 2.0f * (dx * ox + dy * oy + dz * oz);
  EFloat c = (ox * ox + oy * oy + oz * oz) - ef(s.radius) * ef(s.radius);
  EFloat t0, t1;
  if (!ef_solve_quadratic(a, b, c, &t0, &t1)) return false;
  if (t0.upper_bound() > t_max || t1.lower_bound() <= 0.0f) return false;
  EFloat t_shape_hit = t0;
  bool is_t1 = false;
  if (t_shape_hit.lower_bound() <= 0.0f) {
    t_shape_hit = t1; is_t1 = true;
    if (t_shape_hit.upper_bound() > t_max) return false;
  }
  V3 p_hit = o + t_shape_hit.v * d;
  p_hit = p_hit * (s.radius / length(p_hit));
  if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || phi > s.phi_max) {
    // `t_shape_hit == t1` compares the VALUES (PartialEq for EFloat, efloat.rs:122-126)
    if (is_t1 || t_shape_hit.v == t1.v) return false;
    if (t1.upper_bound() > t_max) return false;
    t_shape_hit = t1;
    p_hit = o + t_shape_hit.v * d;
    p_hit = p_hit * (s.radius / length(p_hit));
    if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
    phi = atan2f(p_hit.x, p_hit.y);  // arguments swapped on the retry path in the reference (:134) - kept
    if (phi < 0.0f) phi += 2.0f * kPi;
    if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || phi > s.phi_max) return false;
  }
  out->t = t_shape_hit.v;
  if (!fill) return true;
  float u = phi / s.phi_max;
  float theta = acosf(clamp_t(p_hit.z / s.radius, -1.0f, 1.0f));
  float v = (theta - s.theta_min) / (s.theta_max - s.theta_min);
  float z_radius = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
  float inv_z_radius = 1.0f / z_radius;
  float cos_phi = p_hit.x * inv_z_radius, sin_phi = p_hit.y * inv_z_radius;
  V3 dpdu = v3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  V3 dpdv = (s.theta_max - s.theta_min) * v3(p_hit.z * cos_phi, p_hit.z * sin_phi, -s.radius * sinf(theta));
  // dndu / dndv (:166-184) are computed by the reference and then zeroed by SurfaceInteraction::transform (interaction.rs:169-170): not formed
  V3 p_error = gamma_n(5) * vabs(p_hit);
  // SurfaceInteraction::new (interaction.rs:107-147)
  V3 n = normalize(cross(dpdu, dpdv));
  if (s.reverse_orientation != s.swaps_handedness) n = n * -1.0f;
  V3 wo = normalize(normalize(-d));  // new() normalises, Interaction::new normalises again
  // .transform(object_to_world) (interaction.rs:156-190)
  V3 pe_w;
  out->p = xf_point_with_error(s.o2w.m, p_hit, p_error, &pe_w);
  out->p_error = pe_w;
  out->wo = normalize(normalize(xf_vector(s.o2w.m, wo)));
  out->n = normalize(xf_normal(s.o2w.m_inv, n));
  out->uv = P2{u, v};
  out->dpdu = xf_vector(s.o2w.m, dpdu); out->dpdv = xf_vector(s.o2w.m, dpdv);
  V3 sn = normalize(xf_normal(s.o2w.m_inv, n));
  out->sh_n = dot(sn, out->n) < 0.0f ? -sn : sn;  // face_forward_n(shading.n, hit.n)
  return true;
}

inline V3 uniform_sample_sphere(P2 u) {  // sampling/mod.rs:14-20
  float z = 1.0f - 2.0f * u.x;
  float r = sqrtf(fmaxf(1.0f - z * z, 0.0f));
  float phi = 2.0f * kPi * u.y;
  return v3(r * cosf(phi), r * sinf(phi), z);
}
struct SpherePoint { V3 p, p_error, n; };
inline SpherePoint sphere_sample(const Sphere& s, P2 u, float* pdf) {  // Sphere::sample, :227-244
  V3 p_obj = v3(0, 0, 0) + s.radius * uniform_sample_sphere(u);
  SpherePoint it;
  it.n = normalize(xf_normal(s.o2w.m_inv, p_obj));  // (reverse_orientation is not applied here in the reference)
  p_obj = p_obj * s.radius / sqrtf(distance_squared(p_obj, v3(0, 0, 0)));
  V3 p_obj_error = gamma_n(5) * vabs(p_obj);
  it.p = xf_point_with_error(s.o2w.m, p_obj, p_obj_error, &it.p_error);
  *pdf = 1.0f / sphere_area(s);
  return it;
}
inline SpherePoint sphere_sample_si(const Sphere& s, const Interaction& si, P2 u, float* pdf_out) {  // Sphere::sample_si, :246-308
  V3 p_center = xf_point(s.o2w.m, v3(0, 0, 0));
  V3 p_origin = offset_ray_origin(si.p, si.p_error, si.n, p_center - si.p);
  if (distance_squared(p_origin, p_center) <= s.radius * s.radius) {
    float pdf; SpherePoint intr = sphere_sample(s, u, &pdf);
    V3 wi = intr.p - si.p;
    if (length_squared(wi) == 0.0f) pdf = 0.0f;
    else { wi = normalize(wi); pdf *= distance_squared(si.p, intr.p) / fabsf(dot(intr.n, -wi)); }
    if (std::isinf(pdf)) pdf = 0.0f;
    *pdf_out = pdf;
    return intr;
  }
  V3 wc = normalize(p_center - si.p), wc_x, wc_y;
  coordinate_system(wc, &wc_x, &wc_y);
  float sin_theta_max_2 = s.radius * s.radius / distance_squared(si.p, p_center);
  float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_max_2));
  float cos_theta = (1.0f - u.x) + u.x * cos_theta_max;
  float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
  float phi = u.y * 2.0f * kPi;
  float dc = sqrtf(distance_squared(si.p, p_center));
  float ds = dc * cos_theta - sqrtf(fmaxf(0.0f, s.radius * s.radius - dc * dc * sin_theta * sin_theta));
  float cos_alpha = (dc * dc + s.radius * s.radius - ds * ds) / (2.0f * dc * s.radius);
  float sin_alpha = sqrtf(fmaxf(0.0f, 1.0f - cos_alpha * cos_alpha));
  // spherical_direction_vec(sin_alpha, cos_alpha, phi, -wc_x, -wc_y, -wc), geometry/mod.rs:117-126
  V3 n_world = sin_alpha * cosf(phi) * (-wc_x) + sin_alpha * sinf(phi) * (-wc_y) + cos_alpha * (-wc);
  V3 p_world = p_center + s.radius * v3(n_world.x, n_world.y, n_world.z);
  SpherePoint it;
  it.p = p_world; it.p_error = gamma_n(5) * vabs(p_world); it.n = n_world;
  if (s.reverse_orientation) it.n = it.n * -1.0f;
  *pdf_out = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));
  return it;
}
inline float sphere_pdf_wi(const Sphere& s, const Interaction& si, V3 wi, uint64_t* n_tests) {  // Sphere::pdf_wi, :310-334
  V3 p_center = xf_point(s.o2w.m, v3(0, 0, 0));
  V3 p_origin = offset_ray_origin(si.p, si.p_error, si.n, p_center - si.p);
  if (distance_squared(p_origin, p_center) <= s.radius * s.radius) {
    Ray ray = spawn_ray(si, wi);
    SphereHit h;
    if (n_tests) *n_tests += 1;
    if (!sphere_intersect(s, ray, true, &h)) return 0.0f;
    return distance_squared(si.p, h.p) / (fabsf(dot(h.n, -wi)) * sphere_area(s));
  }
  float sin_theta_max_2 = s.radius * s.radius / distance_squared(si.p, p_center);
  float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_max_2));
  return 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));  // uniform_cone_pdf, sampling/mod.rs:54-56
}


// ---------------------------------------------------------------- Disk (rc/shapes/disk.rs) and Cylinder (rc/shapes/cylinder.rs)
inline Sphere disk_new(const Transform& o2w, float height, float radius, float inner_radius, float phi_max, bool reverse_orientation) {  // Disk::new, disk.rs:25-46
  Sphere s{};
  s.kind = QUADRIC_DISK; s.o2w = o2w; s.height = height; s.radius = radius; s.inner_radius = inner_radius;
  s.phi_max = to_radians(clamp_t(phi_max, 0.0f, 360.0f));
  s.z_min = s.z_max = height; s.theta_min = s.theta_max = 0.0f;
  s.reverse_orientation = reverse_orientation; s.swaps_handedness = xf_swaps_handedness(o2w.m);
  return s;
}
inline Sphere cylinder_new(const Transform& o2w, float radius, float z_min, float z_max, float phi_max, bool reverse_orientation) {  // Cylinder::create, cylinder.rs:26-46
  Sphere s{};
  s.kind = QUADRIC_CYLINDER; s.o2w = o2w; s.radius = radius; s.z_min = z_min; s.z_max = z_max;  // (not sorted, not clamped: as the reference)
  s.phi_max = to_radians(clamp_t(phi_max, 0.0f, 360.0f));
  s.theta_min = s.theta_max = 0.0f;
  s.reverse_orientation = reverse_orientation; s.swaps_handedness = xf_swaps_handedness(o2w.m);
  return s;
}
inline float quadric_area(const Sphere& s) {
  if (s.kind == QUADRIC_DISK) return s.phi_max * 0.5f * (s.radius * s.radius - s.inner_radius * s.inner_radius);  // disk.rs:153-155
  if (s.kind == QUADRIC_CYLINDER) return (s.z_max - s.z_min) * s.radius * s.phi_max;                             // cylinder.rs:252-254
  return sphere_area(s);
}
inline B3 quadric_world_bounds(const Sphere& s) {
  if (s.kind == QUADRIC_DISK) {  // Disk::world_bounds (disk.rs:127-134) maps only two corners of the object box: wrong under a rotation, kept (reference quirk)
    V3 p1 = xf_point(s.o2w.m, v3(-s.radius, -s.radius, s.height)), p2 = xf_point(s.o2w.m, v3(s.radius, s.radius, s.height));
    return b3_from_points(v3(fminf(p1.x, p2.x), fminf(p1.y, p2.y), fminf(p1.z, p2.z)), v3(fmaxf(p1.x, p2.x), fmaxf(p1.y, p2.y), fmaxf(p1.z, p2.z)));
  }
  if (s.kind == QUADRIC_CYLINDER) {  // &object_to_world * &object_bounds (transform.rs:342-380): the union of the 8 mapped corners of from_points(..)
    const B3 ob = b3_from_points(v3(-s.radius, -s.radius, s.z_min), v3(s.radius, s.radius, s.z_max));
    B3 b = b3_empty();
    for (int i = 0; i < 8; ++i) b = b3_union_p(b, xf_point(s.o2w.m, v3(i & 1 ? ob.mx.x : ob.mn.x, i & 2 ? ob.mx.y : ob.mn.y, i & 4 ? ob.mx.z : ob.mn.z)));
    return b;
  }
  return sphere_world_bounds(s);
}
// the tail shared by the three intersect(): SurfaceInteraction::new(p_hit, p_error, uv, -ray.d, dpdu, dpdv, ..) then .transform(object_to_world)
inline void quadric_finish(const Sphere& s, V3 p_hit, V3 p_error, P2 uv, V3 d_obj, V3 dpdu, V3 dpdv, SphereHit* out) {
  V3 n = normalize(cross(dpdu, dpdv));
  if (s.reverse_orientation != s.swaps_handedness) n = n * -1.0f;
  V3 wo = normalize(normalize(-d_obj));
  out->p = xf_point_with_error(s.o2w.m, p_hit, p_error, &out->p_error);
  out->wo = normalize(normalize(xf_vector(s.o2w.m, wo)));
  out->n = normalize(xf_normal(s.o2w.m_inv, n));
  out->uv = uv;
  out->dpdu = xf_vector(s.o2w.m, dpdu); out->dpdv = xf_vector(s.o2w.m, dpdv);
  V3 sn = normalize(xf_normal(s.o2w.m_inv, n));
  out->sh_n = dot(sn, out->n) < 0.0f ? -sn : sn;
}
inline bool disk_intersect(const Sphere& s, const Ray& ray, bool fill, SphereHit* out) {  // Disk::intersect, disk.rs:65-118
  V3 o_err, d_err;
  V3 o = xf_point_err(s.o2w.m_inv, ray.o, &o_err);
  V3 d = xf_vector_err(s.o2w.m_inv, ray.d, &d_err);
  float l2 = length_squared(d);
  if (l2 > 0.0f) { float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
  if (d.z == 0.0f) return false;
  float t_shape_hit = (s.height - o.z) / d.z;
  if (t_shape_hit <= 0.0f || t_shape_hit > ray.t_max) return false;
  V3 p_hit = o + t_shape_hit * d;
  float dist2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
  if (dist2 > s.radius * s.radius || dist2 < s.inner_radius * s.inner_radius) return false;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if (phi > s.phi_max) return false;
  out->t = t_shape_hit;
  if (!fill) return true;
  float u = phi / s.phi_max;
  float r_hit = sqrtf(dist2);
  float one_minus_v = (r_hit - s.inner_radius) / (s.radius - s.inner_radius);
  float v = 1.0f - one_minus_v;
  V3 dpdu = v3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  V3 dpdv = v3(p_hit.x, p_hit.y, 0.0f) * (s.radius - s.inner_radius) / r_hit;
  p_hit.z = s.height;
  quadric_finish(s, p_hit, v3(0, 0, 0), P2{u, v}, d, dpdu, dpdv, out);
  return true;
}
inline bool cylinder_intersect(const Sphere& s, const Ray& ray, bool fill, SphereHit* out) {  // Cylinder::intersect, cylinder.rs:62-172 (intersect_p :174-250: the same decisions)
  V3 o_err, d_err;
  V3 o = xf_point_err(s.o2w.m_inv, ray.o, &o_err);
  V3 d = xf_vector_err(s.o2w.m_inv, ray.d, &d_err);
  float l2 = length_squared(d);
  if (l2 > 0.0f) { float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
  const float t_max = ray.t_max;
  EFloat ox = ef_new(o.x, o_err.x), oy = ef_new(o.y, o_err.y);
  EFloat dx = ef_new(d.x, d_err.x), dy = ef_new(d.y, d_err.y);
  EFloat a = dx * dx + dy * dy;
  EFloat b = 2.0f * (dx * ox + dy * oy);
  EFloat c = ox * ox + oy * oy - ef(s.radius) * ef(s.radius);
  EFloat t0, t1;
  if (!ef_solve_quadratic(a, b, c, &t0, &t1)) return false;
  if (t0.upper_bound() > t_max || t1.lower_bound() <= 0.0f) return false;
  EFloat ts = t0; bool is_t1 = false;
  if (ts.lower_bound() <= 0.0f) { ts = t1; is_t1 = true; if (ts.upper_bound() > t_max) return false; }
  V3 p_hit = o + ts.v * d;
  float hit_rad = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
  p_hit.x *= s.radius / hit_rad; p_hit.y *= s.radius / hit_rad;
  float phi = atan2f(p_hit.y, p_hit.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  if (p_hit.z < s.z_min || p_hit.z > s.z_max || phi > s.phi_max) {
    if (is_t1 || ts.v == t1.v) return false;
    ts = t1;
    if (t1.upper_bound() > t_max) return false;
    p_hit = o + ts.v * d;
    hit_rad = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
    p_hit.x *= s.radius / hit_rad; p_hit.y *= s.radius / hit_rad;
    phi = atan2f(p_hit.y, p_hit.x);
    if (phi < 0.0f) phi += 2.0f * kPi;
    if (p_hit.z < s.z_min || p_hit.z > s.z_max || phi > s.phi_max) return false;
  }
  out->t = ts.v;
  if (!fill) return true;
  float u = phi / s.phi_max;
  float v = (p_hit.z - s.z_min) / (s.z_max / s.z_min);  // a division where pbrt subtracts (cylinder.rs:121): kept
  V3 dpdu = v3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
  V3 dpdv = v3(0.0f, 0.0f, s.z_max - s.z_min);
  V3 p_error = gamma_n(3) * v3(fabsf(p_hit.x), fabsf(p_hit.y), 0.0f);
  quadric_finish(s, p_hit, p_error, P2{u, v}, d, dpdu, dpdv, out);
  return true;
}
inline bool quadric_intersect(const Sphere& s, const Ray& ray, bool fill, SphereHit* out) {
  if (s.kind == QUADRIC_DISK) return disk_intersect(s, ray, fill, out);
  if (s.kind == QUADRIC_CYLINDER) return cylinder_intersect(s, ray, fill, out);
  return sphere_intersect(s, ray, fill, out);
}
inline P2 concentric_sample_disk_q(P2 u) {  // sampling/mod.rs:29-47
  float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
  if (ox == 0.0f && oy == 0.0f) return P2{0.0f, 0.0f};
  float r, theta;
  if (fabsf(ox) > fabsf(oy)) { r = ox; theta = (kPi / 4.0f) * (oy / ox); }
  else { r = oy; theta = (kPi / 2.0f) - (kPi / 4.0f) * (ox / oy); }
  return P2{r * cosf(theta), r * sinf(theta)};
}
// Shape::sample of a disk / cylinder (disk.rs:136-151, cylinder.rs:256-278): a point by area
inline SpherePoint quadric_sample(const Sphere& s, P2 u, float* pdf) {
  SpherePoint it;
  if (s.kind == QUADRIC_DISK) {
    P2 pd = concentric_sample_disk_q(u);
    V3 p_obj = v3(pd.x * s.radius, pd.y * s.radius, s.height);
    it.n = normalize(xf_normal(s.o2w.m_inv, v3(0.0f, 0.0f, 1.0f)));
    if (s.reverse_orientation) it.n = -it.n;
    it.p = xf_point_with_error(s.o2w.m, p_obj, v3(0, 0, 0), &it.p_error);
  } else {
    float z = lerp_f(u.x, s.z_min, s.z_max);
    float phi = u.y * s.phi_max;
    V3 p_obj = v3(s.radius * cosf(phi), s.radius * sinf(phi), z);
    V3 n = normalize(xf_normal(s.o2w.m_inv, v3(p_obj.x, p_obj.y, 0.0f)));
    if (s.reverse_orientation) n = n * -1.0f;
    float hit_rad = sqrtf(p_obj.x * p_obj.x + p_obj.y * p_obj.y);
    p_obj.x *= s.radius / hit_rad; p_obj.y *= s.radius / hit_rad;
    V3 p_obj_error = gamma_n(3) * v3(fabsf(p_obj.x), fabsf(p_obj.y), 0.0f);
    it.p = xf_point_with_error(s.o2w.m, p_obj, p_obj_error, &it.p_error);
    it.n = n;
  }
  *pdf = 1.0f / quadric_area(s);
  return it;
}
// Shape::sample_si / pdf_wi: Sphere's overrides, or the trait defaults (shapes/mod.rs:39-68) over sample() / intersect()
inline SpherePoint quadric_sample_si(const Sphere& s, const Interaction& ref, P2 u, float* pdf_out) {
  if (s.kind == QUADRIC_SPHERE) return sphere_sample_si(s, ref, u, pdf_out);
  float pdf; SpherePoint intr = quadric_sample(s, u, &pdf);
  V3 wi = intr.p - ref.p;
  if (length_squared(wi) == 0.0f) pdf = 0.0f;
  else { wi = normalize(wi); pdf *= distance_squared(ref.p, intr.p) / fabsf(dot(intr.n, -wi)); if (std::isinf(pdf)) pdf = 0.0f; }
  *pdf_out = pdf;
  return intr;
}
inline float quadric_pdf_wi(const Sphere& s, const Interaction& ref, V3 wi, uint64_t* n_tests) {
  if (s.kind == QUADRIC_SPHERE) return sphere_pdf_wi(s, ref, wi, n_tests);
  Ray ray = spawn_ray(ref, wi);
  SphereHit h;
  if (n_tests) *n_tests += 1;
  if (!quadric_intersect(s, ray, true, &h)) return 0.0f;
  return distance_squared(ref.p, h.p) / (fabsf(dot(h.n, -wi)) * quadric_area(s));
}

}  // namespace orc
