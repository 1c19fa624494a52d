// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement (plain scalar C++, f32, no FMA contraction) of the math substrate of
// abusch/rustracer's path-tracing hot path. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything under oracle/. The product path
// (rustracer_amd/csrc + include/) never includes, links or calls this code.
//
// Parity status: the reference cannot be built here (no Rust toolchain); this
// restatement is pinned against the three known-answer vectors the reference's own tests hold
// for this path (Distribution1D::sample_discrete, find_interval, Bounds2i order) and against
// published PCG32 / (0,2)-sequence properties. Everything else is "parity unpinned" by the
// reference and argued by file:line citation. `rc/` = /root/reference/rustracer-core/src/.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

// ---------------------------------------------------------------- scalar helpers (rc/lib.rs)
static const float kInf = std::numeric_limits<float>::infinity();
static const float kPi = 3.14159265358979323846f;       // std::f32::consts::PI
static const float kInvPi = 0.318309886183790671538f;   // FRAC_1_PI
static const float kPiOver2 = 1.57079632679489661923f;  // FRAC_PI_2
static const float kPiOver4 = kPiOver2 / 2.0f;          // rc/sampling/mod.rs:11
static const float kTau = 6.28318530717958647692f;      // consts::TAU
static const float kMachineEpsilon = 1.1920929e-07f * 0.5f;  // rc/lib.rs:89 (f32::EPSILON*0.5)
static const float kOneMinusEpsilon = 0.99999994f;           // rc/lib.rs:95

// rc/lib.rs:90-92
inline float gamma_n(uint32_t n) {
  return ((float)n * kMachineEpsilon) / (1.0f - (float)n * kMachineEpsilon);
}
// rc/lib.rs:192-207 (PartialOrd versions: NaN falls to the second argument)
inline float min_po(float a, float b) { return a < b ? a : b; }
inline float max_po(float a, float b) { return a > b ? a : b; }
inline int min_po(int a, int b) { return a < b ? a : b; }
inline int max_po(int a, int b) { return a > b ? a : b; }
// Rust f32::min/max == IEEE minNum/maxNum == fminf/fmaxf
inline float fmin_ieee(float a, float b) { return fminf(a, b); }
inline float fmax_ieee(float a, float b) { return fmaxf(a, b); }
// rc/lib.rs:264-275
template <class T>
inline T clamp_t(T v, T lo, T hi) {
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}
// rc/lib.rs:107-117  a*(1-t) + b*t
inline float lerp_f(float t, float a, float b) { return a * (1.0f - t) + b * t; }

inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// rc/lib.rs:227-243
inline float next_float_up(float v) {
  if (std::isinf(v) && v > 0.0f) return v;
  if (v == -0.0f) v = 0.0f;
  uint32_t ui = f2bits(v);
  if (v >= 0.0f) ui += 1; else ui -= 1;
  return bits2f(ui);
}
// rc/lib.rs:246-262
inline float next_float_down(float v) {
  if (std::isinf(v) && v < 0.0f) return v;
  if (v == 0.0f) v = -0.0f;
  uint32_t ui = f2bits(v);
  if (v > 0.0f) ui -= 1; else ui += 1;
  return bits2f(ui);
}
// Rust `f32 as i32`: saturating, NaN -> 0.
inline int32_t f2i_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int32_t)f;
}
// Rust `f32 as usize` (64-bit): saturating at 0 below.
inline uint64_t f2u_sat(float f) {
  if (f != f || f <= 0.0f) return 0;
  if (f >= 18446744073709551616.0f) return UINT64_MAX;
  return (uint64_t)f;
}
// Rust `f32 as u32`
inline uint32_t f2u32_sat(float f) {
  if (f != f || f <= 0.0f) return 0;
  if (f >= 4294967296.0f) return UINT32_MAX;
  return (uint32_t)f;
}

// rc/lib.rs:171-189 ; pred is evaluated on indices in [0,size)
template <class P>
inline size_t find_interval(size_t size, P pred) {
  size_t first = 0, len = size;
  while (len > 0) {
    size_t half = len >> 1, middle = first + half;
    if (pred(middle)) { first = middle + 1; len -= half + 1; } else { len = half; }
  }
  long v = (long)first - 1, hi = (long)size - 2;
  // clamp(val, 0, size-2): if val<low low; else if val>high high
  if (v < 0) v = 0; else if (v > hi) v = hi;
  return (size_t)v;
}

// ---------------------------------------------------------------- V3 (rc/geometry/{vector,point,normal}.rs)
struct V3 {
  float x, y, z;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector.rs:242
inline float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V3 a) { return sqrtf(length_squared(a)); }
inline V3 normalize(V3 a) { return a / length(a); }  // vector.rs:276 (divides each component)
inline V3 cross(V3 a, V3 v) {                       // vector.rs:280-286
  return {(a.y * v.z) - (a.z * v.y), (a.z * v.x) - (a.x * v.z), (a.x * v.y) - (a.y * v.x)};
}
inline V3 vabs(V3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// rc/lib.rs:120-135
inline int max_dimension(V3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }
// rc/lib.rs:137-139 (f32::max = maxNum)
inline float max_component(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
inline V3 permute(V3 v, int x, int y, int z) { return {v[x], v[y], v[z]}; }
// rc/lib.rs:158-168
inline void coordinate_system(V3 v1, V3* v2, V3* v3o) {
  if (fabsf(v1.x) > fabsf(v1.y))
    *v2 = v3(-v1.z, 0.0f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
  else
    *v2 = v3(0.0f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
  *v3o = cross(v1, *v2);
}
inline V3 face_forward(V3 v1, V3 v2) { return dot(v1, v2) < 0.0f ? -v1 : v1; }  // geometry/mod.rs:129-144
inline float distance_squared(V3 p1, V3 p2) { return length_squared(p2 - p1); }

// rc/geometry/mod.rs:203-220
inline V3 offset_ray_origin(V3 p, V3 p_error, V3 n, V3 w) {
  float d = dot(vabs(n), p_error);
  V3 offset = d * n;
  if (dot(w, n) < 0.0f) offset = -offset;
  V3 po = p + offset;
  for (int i = 0; i < 3; ++i) {
    if (offset[i] > 0.0f) po.at(i) = next_float_up(po[i]);
    else if (offset[i] < 0.0f) po.at(i) = next_float_down(po[i]);
  }
  return po;
}

// local shading-frame trig, rc/geometry/mod.rs:15-94
inline float cos_theta(V3 w) { return w.z; }
inline float cos2_theta(V3 w) { return w.z * w.z; }
inline float abs_cos_theta(V3 w) { return fabsf(w.z); }
inline float sin2_theta(V3 w) { return fmaxf(1.0f - cos2_theta(w), 0.0f); }
inline float sin_theta(V3 w) { return sqrtf(sin2_theta(w)); }
inline float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
inline float cos_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_t(w.x / s, -1.0f, 1.0f); }
inline float sin_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 0.0f : clamp_t(w.y / s, -1.0f, 1.0f); }
inline float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
inline float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }
inline float spherical_theta(V3 v) { return acosf(clamp_t(v.z, -1.0f, 1.0f)); }
inline float spherical_phi(V3 v) { float p = atan2f(v.y, v.x); return p < 0.0f ? p + 2.0f * kPi : p; }

// ---------------------------------------------------------------- RGB spectrum (rc/spectrum.rs)
struct RGB {
  float r, g, b;
  float operator[](int i) const { return i == 0 ? r : (i == 1 ? g : b); }
};
inline RGB rgb(float r, float g, float b) { return RGB{r, g, b}; }
inline RGB grey(float v) { return RGB{v, v, v}; }
inline RGB operator+(RGB a, RGB b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline RGB operator-(RGB a, RGB b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline RGB operator*(RGB a, RGB b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline RGB operator/(RGB a, RGB b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
inline RGB operator*(RGB a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline RGB operator*(float s, RGB a) { return {s * a.r, s * a.g, s * a.b}; }
inline RGB operator/(RGB a, float s) { return {a.r / s, a.g / s, a.b / s}; }
inline RGB operator+(RGB a, float s) { return {a.r + s, a.g + s, a.b + s}; }
inline RGB operator-(RGB a, float s) { return {a.r - s, a.g - s, a.b - s}; }
inline bool is_black(RGB c) { return c.r == 0.0f && c.g == 0.0f && c.b == 0.0f; }
inline bool has_nan(RGB c) { return c.r != c.r || c.g != c.g || c.b != c.b; }
inline float lum_y(RGB c) { return 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b; }  // spectrum.rs:149-152
inline float max_component_value(RGB c) { return fmaxf(fmaxf(c.r, c.g), c.b); }        // spectrum.rs:154
inline RGB clamp_pos(RGB c) {                                                          // spectrum.rs:158-164
  return {clamp_t(c.r, 0.0f, kInf), clamp_t(c.g, 0.0f, kInf), clamp_t(c.b, 0.0f, kInf)};
}
inline RGB rgb_sqrt(RGB c) { return {sqrtf(c.r), sqrtf(c.g), sqrtf(c.b)}; }
inline void to_xyz(RGB c, float xyz[3]) {  // spectrum.rs:98-106
  xyz[0] = 0.412453f * c.r + 0.357580f * c.g + 0.180423f * c.b;
  xyz[1] = 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b;
  xyz[2] = 0.019334f * c.r + 0.119193f * c.g + 0.950227f * c.b;
}
inline RGB from_xyz(const float xyz[3]) {  // spectrum.rs:91-96
  float r = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
  float g = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
  float b = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
  return {r, g, b};
}

// ---------------------------------------------------------------- 4x4 (rc/geometry/matrix.rs, rc/transform.rs)
struct M44 { float m[4][4]; };
inline M44 m44_identity() {
  M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.0f : 0.0f; return r;
}
inline M44 m44_mul(const M44& a, const M44& b) {  // matrix.rs:157-168
  M44 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
  return r;
}
inline M44 m44_transpose(const M44& a) { M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
// Gauss-Jordan with full pivoting, matrix.rs:72-145
inline M44 m44_inverse(const M44& a) {
  int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
  float minv[4][4];
  memcpy(minv, a.m, sizeof(minv));
  for (int i = 0; i < 4; ++i) {
    int irow = 0, icol = 0;
    float big = 0.0f;
    for (int j = 0; j < 4; ++j) {
      if (ipiv[j] != 1) {
        for (int k = 0; k < 4; ++k) {
          if (ipiv[k] == 0) {
            if (fabsf(minv[j][k]) >= big) { big = fabsf(minv[j][k]); irow = j; icol = k; }
          }
        }
      }
    }
    ipiv[icol] += 1;
    if (irow != icol)
      for (int k = 0; k < 4; ++k) { float t = minv[irow][k]; minv[irow][k] = minv[icol][k]; minv[icol][k] = t; }
    indxr[i] = irow; indxc[i] = icol;
    float pivinv = 1.0f / minv[icol][icol];
    minv[icol][icol] = 1.0f;
    for (int j = 0; j < 4; ++j) minv[icol][j] *= pivinv;
    for (int j = 0; j < 4; ++j) {
      if (j != icol) {
        float save = minv[j][icol];
        minv[j][icol] = 0.0f;
        for (int k = 0; k < 4; ++k) minv[j][k] -= minv[icol][k] * save;
      }
    }
  }
  for (int j = 3; j >= 0; --j)
    if (indxr[j] != indxc[j])
      for (int k = 0; k < 4; ++k) { float t = minv[k][indxr[j]]; minv[k][indxr[j]] = minv[k][indxc[j]]; minv[k][indxc[j]] = t; }
  M44 r; memcpy(r.m, minv, sizeof(minv)); return r;
}
struct Transform { M44 m, m_inv; };
inline Transform xf_identity() { return {m44_identity(), m44_identity()}; }
inline Transform xf_from_matrix(const M44& m) { return {m, m44_inverse(m)}; }
inline Transform xf_inverse(const Transform& t) { return {t.m_inv, t.m}; }
inline Transform xf_mul(const Transform& a, const Transform& b) { return {m44_mul(a.m, b.m), m44_mul(b.m_inv, a.m_inv)}; }
inline Transform xf_translate(V3 d) {  // transform.rs:70-80
  M44 m = m44_identity(), mi = m44_identity();
  m.m[0][3] = d.x; m.m[1][3] = d.y; m.m[2][3] = d.z;
  mi.m[0][3] = -d.x; mi.m[1][3] = -d.y; mi.m[2][3] = -d.z;
  return {m, mi};
}
inline Transform xf_scale(float sx, float sy, float sz) {  // transform.rs:94-117
  M44 m = m44_identity(), mi = m44_identity();
  m.m[0][0] = sx; m.m[1][1] = sy; m.m[2][2] = sz;
  mi.m[0][0] = 1.0f / sx; mi.m[1][1] = 1.0f / sy; mi.m[2][2] = 1.0f / sz;
  return {m, mi};
}
inline float to_radians(float deg) { return deg * (kPi / 180.0f); }  // core f32::to_radians
inline Transform xf_perspective(float fov, float n, float f) {  // transform.rs:157-164
  M44 persp = m44_identity();
  persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n);
  persp.m[3][2] = 1.0f; persp.m[3][3] = 0.0f;
  float inv_tan_ang = 1.0f / tanf(to_radians(fov) / 2.0f);
  return xf_mul(xf_scale(inv_tan_ang, inv_tan_ang, 1.0f), xf_from_matrix(persp));
}
inline Transform xf_look_at(V3 pos, V3 look, V3 up) {  // transform.rs:119-154
  M44 c2w = m44_identity();
  c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1.0f;
  V3 dir = normalize(look - pos);
  if (length(cross(normalize(up), dir)) == 0.0f) return xf_identity();
  V3 left = normalize(cross(normalize(up), dir));
  V3 new_up = cross(dir, left);
  c2w.m[0][0] = left.x; c2w.m[1][0] = left.y; c2w.m[2][0] = left.z; c2w.m[3][0] = 0.0f;
  c2w.m[0][1] = new_up.x; c2w.m[1][1] = new_up.y; c2w.m[2][1] = new_up.z; c2w.m[3][1] = 0.0f;
  c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.0f;
  return {m44_inverse(c2w), c2w};
}
inline V3 xf_point(const M44& m, V3 p) {  // transform.rs:264-286
  float x = p.x, y = p.y, z = p.z;
  float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
  float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
  float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
  float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
  if (wp == 1.0f) return {xp, yp, zp};
  return v3(xp, yp, zp) / wp;
}
inline V3 xf_vector(const M44& m, V3 v) {  // transform.rs:288-303
  float x = v.x, y = v.y, z = v.z;
  return {m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
          m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z};
}
inline V3 xf_normal(const M44& m_inv, V3 n) {  // transform.rs:305-318 (uses transpose of inverse)
  float x = n.x, y = n.y, z = n.z;
  return {m_inv.m[0][0] * x + m_inv.m[1][0] * y + m_inv.m[2][0] * z, m_inv.m[0][1] * x + m_inv.m[1][1] * y + m_inv.m[2][1] * z,
          m_inv.m[0][2] * x + m_inv.m[1][2] * y + m_inv.m[2][2] * z};
}
inline bool xf_swaps_handedness(const M44& m) {  // transform.rs:255-261
  float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) -
              m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
              m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
  return det < 0.0f;
}
// rc/transform.rs:382-394
inline bool solve_linear_system2x2(const float A[2][2], float B0, float B1, float* x0, float* x1) {
  float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
  if (fabsf(det) < 1e-10f) return false;
  float a = (A[1][1] * B0 - A[0][1] * B1) / det;
  float b = (A[0][0] * B1 - A[1][0] * B0) / det;
  if (a != a || b != b) return false;
  *x0 = a; *x1 = b;
  return true;
}

// ---------------------------------------------------------------- Bounds3f (rc/bounds.rs)
struct B3 { V3 mn, mx; };
inline B3 b3_empty() { float M = std::numeric_limits<float>::max(); return {{M, M, M}, {-M, -M, -M}}; }  // bounds.rs:25-32
inline B3 b3_union(const B3& a, const B3& b) {  // bounds.rs:92-108
  return {{min_po(a.mn.x, b.mn.x), min_po(a.mn.y, b.mn.y), min_po(a.mn.z, b.mn.z)},
          {max_po(a.mx.x, b.mx.x), max_po(a.mx.y, b.mx.y), max_po(a.mx.z, b.mx.z)}};
}
inline B3 b3_union_p(B3 b, V3 p) {  // bounds.rs:56-75,110-114
  if (p.x < b.mn.x) b.mn.x = p.x;
  if (p.y < b.mn.y) b.mn.y = p.y;
  if (p.z < b.mn.z) b.mn.z = p.z;
  if (p.x > b.mx.x) b.mx.x = p.x;
  if (p.y > b.mx.y) b.mx.y = p.y;
  if (p.z > b.mx.z) b.mx.z = p.z;
  return b;
}
inline B3 b3_from_points(V3 a, V3 b) {  // bounds.rs:41-46
  return {{min_po(a.x, b.x), min_po(a.y, b.y), min_po(a.z, b.z)}, {max_po(a.x, b.x), max_po(a.y, b.y), max_po(a.z, b.z)}};
}
inline V3 b3_diagonal(const B3& b) { return b.mx - b.mn; }
inline int b3_maximum_extent(const B3& b) { V3 v = b.mx - b.mn; return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }  // :77-90
inline float b3_surface_area(const B3& b) { V3 d = b3_diagonal(b); return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z); }  // :214-217
inline V3 b3_offset(const B3& b, V3 p) {  // :177-190
  V3 o = p - b.mn;
  if (b.mx.x > b.mn.x) o.x /= b.mx.x - b.mn.x;
  if (b.mx.y > b.mn.y) o.y /= b.mx.y - b.mn.y;
  if (b.mx.z > b.mn.z) o.z /= b.mx.z - b.mn.z;
  return o;
}
inline V3 b3_lerp(const B3& b, V3 t) { return {lerp_f(t.x, b.mn.x, b.mx.x), lerp_f(t.y, b.mn.y, b.mx.y), lerp_f(t.z, b.mn.z, b.mx.z)}; }  // :160-166
inline bool b3_inside(const B3& b, V3 p) { return p.x >= b.mn.x && p.x <= b.mx.x && p.y >= b.mn.y && p.y <= b.mx.y && p.z >= b.mn.z && p.z <= b.mx.z; }
inline void b3_bounding_sphere(const B3& b, V3* c, float* r) {  // :199-212
  *c = v3((b.mn.x + b.mx.x) / 2.0f, (b.mn.y + b.mx.y) / 2.0f, (b.mn.z + b.mx.z) / 2.0f);
  *r = b3_inside(b, *c) ? length(b.mx - *c) : 0.0f;
}

}  // namespace orc
