// rtx_host.cpp — host layer of include/rtx_host.h: the work rustracer's Rust host does before and
// around `renderer::render` (scene assembly, SAH BVH, flattening, camera/film set-up), handing the
// hot path to the HIP backend through the C ABI of include/rtx_hip.h. Compiled with g++
// -ffp-contract=off: the BVH bounds, SAH costs and camera matrices must come out bit-identical to
// an f32 evaluation in the reference's operation order (rc/ = rustracer-core/src/).
#include "../../include/rtx_host.h"

#include <algorithm>
#include <array>
#include <initializer_list>
#include <map>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

struct Box { float lo[3], hi[3]; };
const float kFmax = std::numeric_limits<float>::max();
inline Box box_empty() { return Box{{kFmax, kFmax, kFmax}, {-kFmax, -kFmax, -kFmax}}; }  // Bounds3::new, rc/bounds.rs:25-32
inline float pmin(float a, float b) { return a < b ? a : b; }                           // rc/lib.rs:192-207
inline float pmax(float a, float b) { return a > b ? a : b; }
inline Box box_union(const Box& a, const Box& b) {                                       // rc/bounds.rs:92-108
  Box r;
  for (int k = 0; k < 3; ++k) { r.lo[k] = pmin(a.lo[k], b.lo[k]); r.hi[k] = pmax(a.hi[k], b.hi[k]); }
  return r;
}
inline void box_extend(Box& b, const float p[3]) {                                       // rc/bounds.rs:56-75
  for (int k = 0; k < 3; ++k) { if (p[k] < b.lo[k]) b.lo[k] = p[k]; if (p[k] > b.hi[k]) b.hi[k] = p[k]; }
}
inline int box_max_extent(const Box& b) {                                                // rc/bounds.rs:77-90
  float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
  return x > y ? (x > z ? 0 : 2) : (y > z ? 1 : 2);
}
inline float box_area(const Box& b) {                                                    // rc/bounds.rs:214-217
  float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}
inline unsigned long f2usz(float f) { if (f != f || f <= 0.0f) return 0; return (unsigned long)f; }  // Rust `as usize`
inline int f2i(float f) {                                                                           // Rust `as i32`
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return -2147483647 - 1;
  return (int)f;
}

// ------------------------------------------------------------------ 4x4 matrices (rc/geometry/matrix.rs, rc/transform.rs)
struct Mat { float a[4][4]; };
Mat mat_identity() { Mat m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.a[i][j] = i == j ? 1.0f : 0.0f; return m; }
Mat mat_mul(const Mat& x, const Mat& y) {  // matrix.rs:157-168
  Mat r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j] + x.a[i][3] * y.a[3][j];
  return r;
}
Mat mat_inverse(const Mat& in) {  // Gauss-Jordan, matrix.rs:72-145
  int indxc[4] = {0}, indxr[4] = {0}, ipiv[4] = {0};
  Mat w = in;
  for (int i = 0; i < 4; ++i) {
    int irow = 0, icol = 0; float big = 0.0f;
    for (int j = 0; j < 4; ++j)
      if (ipiv[j] != 1)
        for (int k = 0; k < 4; ++k)
          if (ipiv[k] == 0 && std::fabs(w.a[j][k]) >= big) { big = std::fabs(w.a[j][k]); irow = j; icol = k; }
    ipiv[icol] += 1;
    if (irow != icol) for (int k = 0; k < 4; ++k) std::swap(w.a[irow][k], w.a[icol][k]);
    indxr[i] = irow; indxc[i] = icol;
    float pivinv = 1.0f / w.a[icol][icol];
    w.a[icol][icol] = 1.0f;
    for (int j = 0; j < 4; ++j) w.a[icol][j] *= pivinv;
    for (int j = 0; j < 4; ++j)
      if (j != icol) {
        float save = w.a[j][icol];
        w.a[j][icol] = 0.0f;
        for (int k = 0; k < 4; ++k) w.a[j][k] -= w.a[icol][k] * save;
      }
  }
  for (int j = 3; j >= 0; --j)
    if (indxr[j] != indxc[j]) for (int k = 0; k < 4; ++k) std::swap(w.a[k][indxr[j]], w.a[k][indxc[j]]);
  return w;
}
struct Xf { Mat m, inv; };
Xf xf_mul(const Xf& x, const Xf& y) { return Xf{mat_mul(x.m, y.m), mat_mul(y.inv, x.inv)}; }  // transform.rs:320-340
Xf xf_inverse(const Xf& x) { return Xf{x.inv, x.m}; }
Xf xf_scale(float sx, float sy, float sz) {  // transform.rs:94-117
  Xf t{mat_identity(), mat_identity()};
  t.m.a[0][0] = sx; t.m.a[1][1] = sy; t.m.a[2][2] = sz;
  t.inv.a[0][0] = 1.0f / sx; t.inv.a[1][1] = 1.0f / sy; t.inv.a[2][2] = 1.0f / sz;
  return t;
}
Xf xf_translate(float x, float y, float z) {  // transform.rs:70-80
  Xf t{mat_identity(), mat_identity()};
  t.m.a[0][3] = x; t.m.a[1][3] = y; t.m.a[2][3] = z;
  t.inv.a[0][3] = -x; t.inv.a[1][3] = -y; t.inv.a[2][3] = -z;
  return t;
}
Xf xf_perspective(float fov, float n, float f) {  // transform.rs:157-164
  Mat persp = mat_identity();
  persp.a[2][2] = f / (f - n); persp.a[2][3] = -f * n / (f - n); persp.a[3][2] = 1.0f; persp.a[3][3] = 0.0f;
  const float pi = 3.14159265358979323846f;
  float inv_tan = 1.0f / std::tan((fov * (pi / 180.0f)) / 2.0f);
  return xf_mul(xf_scale(inv_tan, inv_tan, 1.0f), Xf{persp, mat_inverse(persp)});
}
void xf_point(const Mat& m, const float p[3], float out[3]) {  // transform.rs:264-286
  float x = p[0], y = p[1], z = p[2];
  float xp = m.a[0][0] * x + m.a[0][1] * y + m.a[0][2] * z + m.a[0][3];
  float yp = m.a[1][0] * x + m.a[1][1] * y + m.a[1][2] * z + m.a[1][3];
  float zp = m.a[2][0] * x + m.a[2][1] * y + m.a[2][2] * z + m.a[2][3];
  float wp = m.a[3][0] * x + m.a[3][1] * y + m.a[3][2] * z + m.a[3][3];
  if (wp == 1.0f) { out[0] = xp; out[1] = yp; out[2] = zp; }
  else { out[0] = xp / wp; out[1] = yp / wp; out[2] = zp / wp; }
}

// ------------------------------------------------------------------ filters (rc/filter/*.rs)
float mitchell_1d(float x, float B, float C) {  // mitchell.rs:25-41
  float fx = std::fabs(x) * 2.0f;
  if (fx < 1.0f) return ((12.0f - 9.0f * B - 6.0f * C) * fx * fx * fx + (-18.0f + 12.0f * B + 6.0f * C) * fx * fx + (6.0f - 2.0f * B)) * (1.0f / 6.0f);
  if (fx < 2.0f) return ((-B - 6.0f * C) * fx * fx * fx + (6.0f * B + 30.0f * C) * fx * fx + (-12.0f * B - 48.0f * C) * fx + (8.0f * B + 24.0f * C)) * (1.0f / 6.0f);
  return 0.0f;
}
float filter_eval(int kind, const float prm[4], float x, float y) {
  const float xw = prm[0], yw = prm[1];
  switch (kind) {
    case 0: return 1.0f;                                                                           // boxfilter.rs:26-28
    case 1: return std::fmax(0.0f, xw - std::fabs(x)) * std::fmax(0.0f, yw - std::fabs(y));         // triangle.rs:29-31
    case 2: {                                                                                      // gaussian.rs:15-38
      float alpha = prm[2];
      float ex = std::exp(-alpha * xw * xw), ey = std::exp(-alpha * yw * yw);
      return std::fmax(std::exp(-alpha * x * x) - ex, 0.0f) * std::fmax(std::exp(-alpha * y * y) - ey, 0.0f);
    }
    default: return mitchell_1d(x * (1.0f / xw), prm[2], prm[3]) * mitchell_1d(y * (1.0f / yw), prm[2], prm[3]);  // mitchell.rs:54-56
  }
}

struct MipLevels {
  int trilinear, wrap; float max_aniso;
  std::vector<int> w, h; std::vector<uint64_t> off; std::vector<float> texels;  // RGB
};
long modl(long a, long b) { long r = a % b; return r < 0 ? r + b : r; }
// MIPMap::texel (mipmap.rs:208-225) on a finished level
void mip_texel(const MipLevels& m, int level, long s, long t, float out[3]) {
  long us = m.w[level], vs = m.h[level];
  if (m.wrap == RT_WRAP_REPEAT) { s = modl(s, us); t = modl(t, vs); }
  else if (m.wrap == RT_WRAP_CLAMP) { s = s < 0 ? 0 : (s > us - 1 ? us - 1 : s); t = t < 0 ? 0 : (t > vs - 1 ? vs - 1 : t); }
  else if (s < 0 || s >= us || t < 0 || t >= vs) { out[0] = out[1] = out[2] = 0.0f; return; }
  const float* p = &m.texels[3 * (m.off[level] + (uint64_t)t * us + s)];
  out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}
void mip_triangle(const MipLevels& m, int level, float sx, float sy, float out[3]) {  // mipmap.rs:285-308
  int nl = (int)m.w.size();
  level = level < 0 ? 0 : (level > nl - 1 ? nl - 1 : level);
  float s = sx * (float)m.w[level] - 0.5f, t = sy * (float)m.h[level] - 0.5f;
  long s0 = f2i(std::floor(s)), t0 = f2i(std::floor(t));
  float ds = s - (float)s0, dt = t - (float)t0;
  float a[3], b[3], c[3], d[3];
  mip_texel(m, level, s0, t0, a); mip_texel(m, level, s0, t0 + 1, b); mip_texel(m, level, s0 + 1, t0, c); mip_texel(m, level, s0 + 1, t0 + 1, d);
  for (int k = 0; k < 3; ++k) out[k] = a[k] * (1.0f - ds) * (1.0f - dt) + b[k] * (1.0f - ds) * dt + c[k] * ds * (1.0f - dt) + d[k] * ds * dt;
}
void mip_lookup(const MipLevels& m, float sx, float sy, float width, float out[3]) {  // mipmap.rs:227-245
  int nl = (int)m.w.size();
  float level = (float)nl - 1.0f + std::log2(std::fmax(width, 1e-8f));
  if (level < 0.0f) { mip_triangle(m, 0, sx, sy, out); return; }
  if (level >= (float)nl - 1.0f) { mip_texel(m, nl - 1, 0, 0, out); return; }
  float il = std::floor(level), delta = level - il;
  float a[3], b[3];
  mip_triangle(m, (int)f2usz(il), sx, sy, a); mip_triangle(m, (int)f2usz(il) + 1, sx, sy, b);
  for (int k = 0; k < 3; ++k) out[k] = a[k] * (1.0f - delta) + b[k] * delta;
}

struct Dist1 { std::vector<float> func, cdf; float func_int; };
void dist1_init(Dist1& d, const float* f, size_t n) {  // Distribution1D::new, distribution1d.rs:11-42
  d.func.assign(f, f + n); d.cdf.assign(n + 1, 0.0f);
  for (size_t i = 1; i < n + 1; ++i) d.cdf[i] = d.cdf[i - 1] + d.func[i - 1] / (float)n;
  d.func_int = d.cdf[n];
  if (d.func_int == 0.0f) for (size_t i = 1; i < n + 1; ++i) d.cdf[i] = (float)i / (float)n;
  else for (size_t i = 1; i < n + 1; ++i) d.cdf[i] /= d.func_int;
}

struct HostLight {
  rt_light l;
  int tri_source = -1;  // input-order triangle (area lights)
  std::vector<float> dfunc, dcdf, dint, mfunc, mcdf;
};

}  // namespace

struct rtxh_scene {
  // input soup
  std::vector<float> P, N, UV, S; std::vector<int32_t> idx, tri_mat, tri_light; std::vector<uint8_t> tri_flags;
  std::vector<int32_t> tri_alpha;  // 2 per triangle {alpha, shadowalpha} float-texture ids or -1 (empty: no mesh carries a mask)
  // Shape "sphere": primitive ids n_tris() .. n_prims() - 1
  struct HostSphere { rt_sphere s; int32_t material, light; };
  std::vector<HostSphere> spheres; std::vector<rt_sphere> f_spheres;
  // ObjectBegin / ObjectInstance: primitive ids after the spheres. An object keeps its own soup (object space), tree and leaf order.
  struct HostObject {
    std::vector<float> P, N, UV, S; std::vector<int32_t> idx, tri_mat; std::vector<uint8_t> tri_flags;
    std::vector<int32_t> tri_emit;  // per triangle: an unlisted emitter (rtxh_scene_add_emitter) or -1; empty = none
    // round 6: an object holds whatever the reference's object definition holds (api.rs:1019-1051 pushes any Primitive): quadrics in OBJECT space (their own
    // object_to_world = the CTM inside the definition; light = -1 or -2 - k: unlisted emitter k) and alpha / shadowalpha masks on its triangles. Primitive ids inside the
    // object: the triangles, then the quadrics.
    std::vector<HostSphere> spheres;
    std::vector<int32_t> tri_alpha;  // 2 per triangle or empty
    std::vector<rt_bvh_node> nodes; std::vector<int32_t> ordered;
    size_t n_tris() const { return idx.size() / 3; }
    size_t n_prims() const { return n_tris() + spheres.size(); }
  };
  struct HostInstance { int32_t object; float o2w[16], w2o[16]; };
  std::vector<HostObject> objects; std::vector<HostInstance> instances; std::vector<rt_instance> f_instances;
  // emitters that are in no light list (a shape with an AreaLightSource inside an object definition, rc/api.rs:954-964): DiffuseAreaLight parameters
  typedef rtxh_emitter_info Emitter;
  std::vector<Emitter> emitters;
  std::vector<rt_bvh_node> f_nodes;  // with instances: the top-level tree followed by the objects' trees (what rt_scene_desc::nodes points at)
  size_t n_top_prims = 0;            // ... and the top-level primitives' share of the flattened arrays
  size_t n_listed_lights = 0;        // f_lights: the scene's lights first, then `emitters`
  size_t n_prims() const { return n_tris() + spheres.size() + instances.size(); }
  std::vector<rt_texture> textures; std::vector<rt_material> materials; std::vector<MipLevels> mips; std::vector<HostLight> lights;
  // BVH products
  std::vector<rt_bvh_node> nodes; std::vector<int32_t> ordered;
  // flattened arrays (leaf order)
  std::vector<float> f_p, f_n, f_uv, f_s; std::vector<rt_tri_meta> f_meta; std::vector<int32_t> f_alpha; std::vector<rt_light> f_lights; std::vector<rt_image> f_images;
  bool committed = false; unsigned commit_gen = 0, multi_gen = 0;  // multi_gen: the commit the replicas of `multi` were made from
  rt_scene* dev = nullptr;
  rt_multi* multi = nullptr; std::vector<int32_t> multi_devices;  // replicas for rtxh_render_multi, kept while the device list stays the same
  size_t n_tris() const { return idx.size() / 3; }
};

namespace {

// ------------------------------------------------------------------ SAH BVH (rc/bvh/mod.rs:80-358)
struct Builder {
  // primitive_info as parallel arrays (BVHPrimitiveInfo, :522-536), owned by the caller; a builder permutes them in place
  int32_t* prim; Box* pb; float* cx; float* cy; float* cz;
  struct BNode { Box b; int axis, first, count, left, right; };
  std::vector<BNode> pool; int32_t* ordered; int max_prims;  // ordered: n_tris entries, written at explicit offsets
  // Subtrees of at most `task_size` primitives are not built by build() but recorded as tasks (placeholder node index
  // -(task + 2)), so that they can be built concurrently: a subtree owns the index range [start, end) of the primitive arrays
  // and the slice [obase, obase + end - start) of `ordered` (the right child's leaves come first, :290-299), nothing else.
  struct Task { size_t start, end, obase; };
  std::vector<Task>* tasks = nullptr; size_t task_size = 0;

  float centroid(size_t i, int dim) const { return dim == 0 ? cx[i] : (dim == 1 ? cy[i] : cz[i]); }
  void swap_info(size_t a, size_t b) { std::swap(prim[a], prim[b]); std::swap(pb[a], pb[b]); std::swap(cx[a], cx[b]); std::swap(cy[a], cy[b]); std::swap(cz[a], cz[b]); }
  int make_leaf(size_t start, size_t end, size_t obase, const Box& b) {
    BNode n{b, 0, (int)obase, (int)(end - start), -1, -1};
    for (size_t i = start; i < end; ++i) ordered[obase + (i - start)] = prim[i];
    pool.push_back(n);
    return (int)pool.size() - 1;
  }
  int bucket(const Box& cb, size_t i, int dim) const {  // (N_BUCKETS as f32 * centroids_bounds.offset(c)[dim]) as usize, :220-225
    float o = centroid(i, dim) - cb.lo[dim];
    if (cb.hi[dim] > cb.lo[dim]) o /= cb.hi[dim] - cb.lo[dim];  // Bounds3::offset, bounds.rs:177-190
    int b = (int)f2usz(12.0f * o);
    return b == 12 ? 11 : b;
  }
  int build(size_t start, size_t end, size_t obase) {
    const size_t n = end - start;
    if (tasks && n <= task_size && n > 1) { tasks->push_back(Task{start, end, obase}); return -(int)tasks->size() - 1; }
    Box bounds = box_empty();
    for (size_t i = start; i < end; ++i) bounds = box_union(bounds, pb[i]);
    if (n == 1) return make_leaf(start, end, obase, bounds);
    Box cb = box_empty();
    for (size_t i = start; i < end; ++i) { float c[3] = {cx[i], cy[i], cz[i]}; box_extend(cb, c); }
    const int dim = box_max_extent(cb);
    if (cb.lo[dim] == cb.hi[dim]) return make_leaf(start, end, obase, bounds);
    size_t mid;
    if (n <= 2) {
      mid = (start + end) / 2;
      if (start != end - 1 && centroid(end - 1, dim) < centroid(start, dim)) swap_info(start, end - 1);
    } else {
      int cnt[12] = {0}; Box bb[12];
      for (int k = 0; k < 12; ++k) bb[k] = box_empty();
      for (size_t i = start; i < end; ++i) { int b = bucket(cb, i, dim); cnt[b] += 1; bb[b] = box_union(bb[b], pb[i]); }
      float best = 0.0f; int best_k = 0;
      for (int k = 0; k < 11; ++k) {
        Box b0 = box_empty(), b1 = box_empty(); int c0 = 0, c1 = 0;
        for (int j = 0; j <= k; ++j) { b0 = box_union(b0, bb[j]); c0 += cnt[j]; }
        for (int j = k + 1; j < 12; ++j) { b1 = box_union(b1, bb[j]); c1 += cnt[j]; }
        float cost = 1.0f + ((float)c0 * box_area(b0) + (float)c1 * box_area(b1)) / box_area(bounds);
        if (k == 0 || cost < best) { best = cost; best_k = k; }
      }
      if ((int)n > max_prims || best < (float)n) {
        // itertools::partition (0.10.3): front cursor finds a failing element, back cursor the next passing one; swap
        size_t split = 0, front = start, back = end;
        bool done = false;
        while (!done && front < back) {
          size_t f = front++;
          if (!(bucket(cb, f, dim) <= best_k)) {
            bool found = false;
            while (front < back) { size_t b = --back; if (bucket(cb, b, dim) <= best_k) { swap_info(f, b); found = true; break; } }
            if (!found) { done = true; break; }
          }
          split += 1;
        }
        mid = start + split;
      } else return make_leaf(start, end, obase, bounds);
    }
    int right = build(mid, end, obase);  // right child first (:290-299): its leaves come first in ordered_prims
    int left = build(start, mid, obase + (end - mid));
    // BVHBuildNode::interior unions the children's bounds (:64-78), which are the unions of their primitives' boxes: `bounds`
    BNode nd{bounds, dim, 0, 0, left, right};
    pool.push_back(nd);
    return (int)pool.size() - 1;
  }
};

int finish_commit(rtxh_scene* s);

// Sphere::world_bounds (sphere.rs:212-225): the 8 corners of the object-space box through object_to_world, in the reference's order
static Box sphere_world_box(const rt_sphere& sp) {
  auto map = [&](float x, float y, float z, float* p) {  // Transform * Point3f, transform.rs:264-286
    for (int r = 0; r < 3; ++r) p[r] = sp.o2w[4 * r] * x + sp.o2w[4 * r + 1] * y + sp.o2w[4 * r + 2] * z + sp.o2w[4 * r + 3];
    const float wp = sp.o2w[12] * x + sp.o2w[13] * y + sp.o2w[14] * z + sp.o2w[15];
    if (wp != 1.0f) for (int r = 0; r < 3; ++r) p[r] = p[r] / wp;
  };
  if (sp.kind == 1) {  // Disk::world_bounds (disk.rs:127-134) maps two corners of the object box only - wrong under a rotation, kept as it is
    float p1[3], p2[3]; map(-sp.radius, -sp.radius, sp.height, p1); map(sp.radius, sp.radius, sp.height, p2);
    Box bb; for (int k = 0; k < 3; ++k) { bb.lo[k] = std::fmin(p1[k], p2[k]); bb.hi[k] = std::fmax(p1[k], p2[k]); }
    return bb;
  }
  // Sphere: the 8 corners in the order of sphere.rs:212-225; Cylinder: &object_to_world * &object_bounds (transform.rs:342-380) - the same 8 corners of
  // from_points((-r, -r, z_min), (r, r, z_max)), whose min / max do not depend on the order
  const float lo[3] = {-sp.radius, -sp.radius, sp.kind == 2 ? std::fmin(sp.z_min, sp.z_max) : sp.z_min}, hi[3] = {sp.radius, sp.radius, sp.kind == 2 ? std::fmax(sp.z_min, sp.z_max) : sp.z_max};
  const int order[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  Box bb; const float M = std::numeric_limits<float>::max();
  for (int k = 0; k < 3; ++k) { bb.lo[k] = M; bb.hi[k] = -M; }  // Bounds3f::new, bounds.rs:25-32
  for (int c = 0; c < 8; ++c) {
    const float x = order[c][0] ? hi[0] : lo[0], y = order[c][1] ? hi[1] : lo[1], z = order[c][2] ? hi[2] : lo[2];
    float p[3];
    for (int r = 0; r < 3; ++r) p[r] = sp.o2w[4 * r] * x + sp.o2w[4 * r + 1] * y + sp.o2w[4 * r + 2] * z + sp.o2w[4 * r + 3];
    const float wp = sp.o2w[12] * x + sp.o2w[13] * y + sp.o2w[14] * z + sp.o2w[15];  // Transform * Point3f, transform.rs:264-286
    if (wp != 1.0f) for (int r = 0; r < 3; ++r) p[r] = p[r] / wp;
    box_extend(bb, p);
  }
  return bb;
}

// BVH::new + flatten_bvh over `nt` primitives given by their world boxes (rc/bvh/mod.rs:80-358): nodes in pre-order, `ordered` = leaf order -> primitive
static void build_tree(size_t nt, const std::vector<Box>& boxes, int max_prims_per_node, std::vector<rt_bvh_node>& nodes, std::vector<int32_t>& ordered) {
  Builder b; b.max_prims = max_prims_per_node > 255 ? 255 : max_prims_per_node;
  std::vector<int32_t> a_prim(nt); std::vector<Box> a_pb(boxes); std::vector<float> a_cx(nt), a_cy(nt), a_cz(nt);
  b.prim = a_prim.data(); b.pb = a_pb.data(); b.cx = a_cx.data(); b.cy = a_cy.data(); b.cz = a_cz.data();
  for (size_t t = 0; t < nt; ++t) {
    const Box& bb = boxes[t];
    b.prim[t] = (int32_t)t;
    b.cx[t] = 0.5f * bb.lo[0] + 0.5f * bb.hi[0]; b.cy[t] = 0.5f * bb.lo[1] + 0.5f * bb.hi[1]; b.cz[t] = 0.5f * bb.lo[2] + 0.5f * bb.hi[2];  // :532
  }
  ordered.assign(nt, -1);
  b.ordered = ordered.data();
  int root;
  unsigned n_threads = std::thread::hardware_concurrency(); if (n_threads > 16) n_threads = 16; if (n_threads < 1) n_threads = 1;
  if (nt < 65536 || n_threads == 1) root = b.build(0, nt, 0);
  else {
    // top of the tree serially (every level is one partition pass over its range), subtrees of <= nt / (8 threads) primitives concurrently
    std::vector<Builder::Task> tasks;
    b.tasks = &tasks; b.task_size = nt / (8 * (size_t)n_threads) + 1;
    root = b.build(0, nt, 0);
    b.tasks = nullptr;
    std::vector<std::vector<Builder::BNode>> pools(tasks.size());
    std::vector<int> roots(tasks.size(), -1);
    std::atomic<size_t> next{0};
    auto run = [&]() {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= tasks.size()) return;
        Builder lb = b;  // same arrays, same `ordered`; its own node pool
        lb.pool.clear(); lb.tasks = nullptr;
        roots[k] = lb.build(tasks[k].start, tasks[k].end, tasks[k].obase);
        pools[k].swap(lb.pool);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_threads; ++t) th.emplace_back(run);
    run();
    for (auto& x : th) x.join();
    std::vector<int> task_root(tasks.size());
    for (size_t k = 0; k < tasks.size(); ++k) {
      const int off = (int)b.pool.size();
      for (Builder::BNode nd : pools[k]) { if (nd.count == 0) { nd.left += off; nd.right += off; } b.pool.push_back(nd); }
      task_root[k] = off + roots[k];
    }
    auto fix = [&](int& c) { if (c < -1) c = task_root[(size_t)(-c - 2)]; };
    for (auto& nd : b.pool) if (nd.count == 0) { fix(nd.left); fix(nd.right); }
    if (root < -1) root = task_root[(size_t)(-root - 2)];
  }
  // flatten_bvh (:314-358): pre-order, left child adjacent, second child offset patched afterwards
  nodes.clear(); nodes.reserve(b.pool.size());
  struct Item { int node; int parent_flat; };
  std::vector<Item> st; st.push_back({root, -1});
  while (!st.empty()) {
    Item it = st.back(); st.pop_back();
    const Builder::BNode& n = b.pool[it.node];
    rt_bvh_node out{};
    for (int k = 0; k < 3; ++k) { out.bmin[k] = n.b.lo[k]; out.bmax[k] = n.b.hi[k]; }
    int me = (int)nodes.size();
    if (it.parent_flat >= 0) nodes[it.parent_flat].offset = (uint32_t)me;  // this node is its parent's second child
    if (n.count > 0) { out.offset = (uint32_t)n.first; out.n_prims = (uint16_t)n.count; out.axis = 0; }
    else { out.offset = 0; out.n_prims = 0; out.axis = (uint8_t)n.axis; }
    nodes.push_back(out);
    if (n.count == 0) { st.push_back({n.right, me}); st.push_back({n.left, -1}); }
  }
}
static Box triangle_box(const std::vector<float>& P, const std::vector<int32_t>& idx, size_t t) {  // Triangle::world_bounds, mesh.rs:603-608
  const float* p0 = &P[3 * idx[3 * t]]; const float* p1 = &P[3 * idx[3 * t + 1]]; const float* p2 = &P[3 * idx[3 * t + 2]];
  Box bb;
  for (int k = 0; k < 3; ++k) { bb.lo[k] = pmin(p0[k], p1[k]); bb.hi[k] = pmax(p0[k], p1[k]); }
  box_extend(bb, p2);
  return bb;
}
// TransformedPrimitive::world_bounds (primitive.rs:86-88): Transform * Bounds3f - the 8 corners of the object's bounds (transform.rs:342-378)
static Box instance_world_box(const rtxh_scene::HostObject& o, const float* o2w) {
  Box ob;
  if (o.n_prims() == 1) ob = o.n_tris() == 1 ? triangle_box(o.P, o.idx, 0) : sphere_world_box(o.spheres[0].s);  // a single primitive is wrapped as it is (api.rs:1073-1082)
  else { for (int k = 0; k < 3; ++k) { ob.lo[k] = o.nodes[0].bmin[k]; ob.hi[k] = o.nodes[0].bmax[k]; } }
  const int order[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
  Box bb;
  for (int c = 0; c < 8; ++c) {
    const float x = order[c][0] ? ob.hi[0] : ob.lo[0], y = order[c][1] ? ob.hi[1] : ob.lo[1], z = order[c][2] ? ob.hi[2] : ob.lo[2];
    float p[3];
    for (int r = 0; r < 3; ++r) p[r] = o2w[4 * r] * x + o2w[4 * r + 1] * y + o2w[4 * r + 2] * z + o2w[4 * r + 3];
    const float wp = o2w[12] * x + o2w[13] * y + o2w[14] * z + o2w[15];  // Transform * Point3f, transform.rs:264-286
    if (wp != 1.0f) for (int r = 0; r < 3; ++r) p[r] = p[r] / wp;
    if (c == 0) { for (int k = 0; k < 3; ++k) bb.lo[k] = bb.hi[k] = p[k]; }  // Bounds3f::from_point
    else box_extend(bb, p);
  }
  return bb;
}

int commit_scene(rtxh_scene* s, int max_prims_per_node) {
  const size_t n_triangles = s->n_tris(), n_geom = n_triangles + s->spheres.size();
  const size_t nt = s->n_prims();  // every primitive: the triangles, then the spheres, then the object instances
  if (nt == 0) return fail(RT_ERR_INVALID, "no triangles");
  for (auto& o : s->objects) {  // make_accelerator over the object's primitives, with the scene's accelerator parameters (api.rs:1073-1080)
    const size_t on = o.n_prims(), otri = o.n_tris();
    if (on == 0) return fail(RT_ERR_INVALID, "an object without primitives");
    std::vector<Box> ob(on);
    for (size_t t = 0; t < on; ++t) ob[t] = t < otri ? triangle_box(o.P, o.idx, t) : sphere_world_box(o.spheres[t - otri].s);  // (a quadric's box under its own object_to_world: the object's space)
    build_tree(on, ob, max_prims_per_node, o.nodes, o.ordered);
  }
  std::vector<Box> boxes(nt);
  for (size_t t = 0; t < nt; ++t) {
    if (t >= n_geom) { const auto& in = s->instances[t - n_geom]; boxes[t] = instance_world_box(s->objects[(size_t)in.object], in.o2w); }
    else if (t >= n_triangles) boxes[t] = sphere_world_box(s->spheres[t - n_triangles].s);
    else boxes[t] = triangle_box(s->P, s->idx, t);
  }
  build_tree(nt, boxes, max_prims_per_node, s->nodes, s->ordered);
  return finish_commit(s);
}

// BVH by the device's linear builder (rt_bvh_build) instead of the SAH recursion above; everything after the tree is shared
int commit_scene_device(rtxh_scene* s, int max_prims_per_node, float* ms_device) {
  const size_t nt = s->n_tris();
  if (nt == 0) return fail(RT_ERR_INVALID, "no triangles");
  if (!s->spheres.empty() || !s->instances.empty()) return fail(RT_ERR_UNSUPPORTED, "the device BVH builder takes triangles only: a scene with analytic spheres or object instances is built by the host");
  std::vector<float> tp(nt * 9);
  for (size_t t = 0; t < nt; ++t) for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) tp[9 * t + 3 * v + k] = s->P[3 * s->idx[3 * t + v] + k];
  s->nodes.assign(2 * nt - 1, rt_bvh_node{}); s->ordered.assign(nt, -1);
  uint32_t n_nodes = 0;
  const int rc = rt_bvh_build(tp.data(), (uint32_t)nt, max_prims_per_node, s->nodes.data(), &n_nodes, s->ordered.data(), ms_device);
  if (rc != RT_OK) { s->nodes.clear(); s->ordered.clear(); g_err.clear(); return rc; }
  s->nodes.resize(n_nodes);
  return finish_commit(s);
}

int finish_commit(rtxh_scene* s) {
  const size_t n_triangles = s->n_tris();
  const size_t nt = s->n_prims();
  // world bounding sphere for distant / infinite lights (Scene::new -> Light::preprocess, scene.rs:29-49; bounds.rs:199-212)
  const rt_bvh_node& r0 = s->nodes[0];
  float c[3] = {(r0.bmin[0] + r0.bmax[0]) / 2.0f, (r0.bmin[1] + r0.bmax[1]) / 2.0f, (r0.bmin[2] + r0.bmax[2]) / 2.0f};
  bool inside = c[0] >= r0.bmin[0] && c[0] <= r0.bmax[0] && c[1] >= r0.bmin[1] && c[1] <= r0.bmax[1] && c[2] >= r0.bmin[2] && c[2] <= r0.bmax[2];
  float dx = r0.bmax[0] - c[0], dy = r0.bmax[1] - c[1], dz = r0.bmax[2] - c[2];
  float world_radius = inside ? std::sqrt(dx * dx + dy * dy + dz * dz) : 0.0f;

  // flatten geometry into leaf order
  std::vector<int32_t> leaf_of(nt, -1);
  for (size_t i = 0; i < nt; ++i) leaf_of[s->ordered[i]] = (int32_t)i;
  bool any_n = !s->N.empty(), any_uv = !s->UV.empty(), any_s = !s->S.empty();
  const size_t n_geom = n_triangles + s->spheres.size();
  // object primitives follow the top level's in the same arrays: every INSTANCED object once, in the order of its first instance
  std::vector<int64_t> obj_prim_base(s->objects.size(), -1), obj_node_base(s->objects.size(), -1);
  std::vector<size_t> obj_sphere_base(s->objects.size(), 0);  // where an instanced object's quadrics sit behind the top level's in the scene's quadric table
  std::vector<size_t> instanced;  // the instanced objects in that order
  size_t n_all = nt, n_all_nodes = s->nodes.size(), n_obj_spheres = 0; bool any_obj_alpha = false;
  for (const auto& in : s->instances) {
    const size_t o = (size_t)in.object;
    if (obj_prim_base[o] >= 0) continue;
    instanced.push_back(o);
    obj_prim_base[o] = (int64_t)n_all; obj_node_base[o] = (int64_t)n_all_nodes;
    obj_sphere_base[o] = n_obj_spheres; n_obj_spheres += s->objects[o].spheres.size(); any_obj_alpha = any_obj_alpha || !s->objects[o].tri_alpha.empty();
    n_all += s->objects[o].n_prims(); n_all_nodes += s->objects[o].nodes.size();
    any_n = any_n || !s->objects[o].N.empty(); any_uv = any_uv || !s->objects[o].UV.empty(); any_s = any_s || !s->objects[o].S.empty();
  }
  s->n_top_prims = nt;
  std::vector<int64_t> emitter_prim(s->emitters.size(), -1);  // a primitive that carries each unlisted emitter (its DLight names one; nothing samples it)
  s->f_p.assign(n_all * 9, 0.0f); s->f_meta.assign(n_all, rt_tri_meta{});
  s->f_n.clear(); s->f_uv.clear(); s->f_s.clear();
  if (any_n) s->f_n.assign(n_all * 9, 0.0f); if (any_uv) s->f_uv.assign(n_all * 6, 0.0f); if (any_s) s->f_s.assign(n_all * 9, 0.0f);
  s->f_instances.clear(); s->f_nodes.clear();
  if (!s->instances.empty()) {
    s->f_nodes = s->nodes; s->f_nodes.resize(n_all_nodes);
    for (size_t o = 0; o < s->objects.size(); ++o) {
      if (obj_prim_base[o] < 0) continue;
      const auto& ob = s->objects[o];
      std::copy(ob.nodes.begin(), ob.nodes.end(), s->f_nodes.begin() + obj_node_base[o]);
      const bool on = !ob.N.empty(), ouv = !ob.UV.empty(), os = !ob.S.empty();
      for (size_t i = 0; i < ob.n_prims(); ++i) {
        const size_t g = (size_t)obj_prim_base[o] + i; const int32_t t = ob.ordered[i];
        if ((size_t)t >= ob.n_tris()) {  // a quadric of the object: its box (object space) in p0 / p1, its index in the scene's quadric table as the bits of p2.x
          const auto& hs = ob.spheres[(size_t)t - ob.n_tris()];
          const Box bb = sphere_world_box(hs.s);
          for (int k = 0; k < 3; ++k) { s->f_p[9 * g + k] = bb.lo[k]; s->f_p[9 * g + 3 + k] = bb.hi[k]; }
          const uint32_t si = (uint32_t)(s->spheres.size() + obj_sphere_base[o] + ((size_t)t - ob.n_tris())); memcpy(&s->f_p[9 * g + 6], &si, 4);
          const uint32_t flip = (hs.s.reverse_orientation != 0) != (hs.s.swaps_handedness != 0) ? RT_TRI_FLIP : 0u;
          int32_t sl = -1;
          if (hs.light <= -2) {
            const size_t k = (size_t)(-2 - (int64_t)hs.light);
            if (k >= s->emitters.size()) return fail(RT_ERR_INVALID, "emitter index out of range");
            sl = (int32_t)(s->lights.size() + k);
            if (emitter_prim[k] < 0) emitter_prim[k] = (int64_t)g;
          }
          s->f_meta[g] = rt_tri_meta{hs.material, sl, RT_PRIM_SPHERE | flip, (uint32_t)t};
          continue;
        }
        for (int v = 0; v < 3; ++v) {
          const int32_t vi = ob.idx[3 * t + v];
          for (int k = 0; k < 3; ++k) s->f_p[9 * g + 3 * v + k] = ob.P[3 * vi + k];
          if (on) for (int k = 0; k < 3; ++k) s->f_n[9 * g + 3 * v + k] = ob.N[3 * vi + k];
          if (os) for (int k = 0; k < 3; ++k) s->f_s[9 * g + 3 * v + k] = ob.S[3 * vi + k];
          if (ouv) for (int k = 0; k < 2; ++k) s->f_uv[6 * g + 2 * v + k] = ob.UV[2 * vi + k];
        }
        const int32_t em = ob.tri_emit.empty() ? -1 : ob.tri_emit[t];
        s->f_meta[g] = rt_tri_meta{ob.tri_mat[t], em < 0 ? -1 : (int32_t)(s->lights.size() + (size_t)em), (uint32_t)ob.tri_flags[t], (uint32_t)t};
        if (em >= 0 && emitter_prim[(size_t)em] < 0) emitter_prim[(size_t)em] = (int64_t)g;
      }
    }
    for (const auto& in : s->instances) {
      const auto& ob = s->objects[(size_t)in.object];
      rt_instance ri{};
      memcpy(ri.o2w, in.o2w, 64); memcpy(ri.w2o, in.w2o, 64);
      ri.node_base = (uint32_t)obj_node_base[(size_t)in.object]; ri.n_nodes = ob.n_prims() == 1 ? 0u : (uint32_t)ob.nodes.size();
      ri.prim_base = (uint32_t)obj_prim_base[(size_t)in.object]; ri.n_prims = (uint32_t)ob.n_prims();
      s->f_instances.push_back(ri);
    }
  }
  s->f_spheres.clear();
  for (const auto& hs : s->spheres) s->f_spheres.push_back(hs.s);
  for (const size_t o : instanced) for (const auto& hs : s->objects[o].spheres) s->f_spheres.push_back(hs.s);  // (the order of obj_sphere_base: of the objects' FIRST INSTANCES, not of their definitions)
  for (size_t i = 0; i < nt; ++i) {
    const int32_t t = s->ordered[i];
    if ((size_t)t >= n_geom) {  // an object instance: world box in p0 / p1, its index as the bits of p2.x
      const auto& in = s->instances[(size_t)t - n_geom];
      const Box bb = instance_world_box(s->objects[(size_t)in.object], in.o2w);
      for (int k = 0; k < 3; ++k) { s->f_p[9 * i + k] = bb.lo[k]; s->f_p[9 * i + 3 + k] = bb.hi[k]; }
      const uint32_t ii = (uint32_t)((size_t)t - n_geom); memcpy(&s->f_p[9 * i + 6], &ii, 4);
      s->f_meta[i] = rt_tri_meta{-1, -1, RT_PRIM_INSTANCE, (uint32_t)t};
      continue;
    }
    if ((size_t)t >= n_triangles) {  // a sphere: world box in p0 / p1, its index as the bits of p2.x
      const auto& hs = s->spheres[(size_t)t - n_triangles];
      const Box bb = sphere_world_box(hs.s);
      for (int k = 0; k < 3; ++k) { s->f_p[9 * i + k] = bb.lo[k]; s->f_p[9 * i + 3 + k] = bb.hi[k]; }
      const uint32_t si = (uint32_t)((size_t)t - n_triangles); memcpy(&s->f_p[9 * i + 6], &si, 4);
      const uint32_t flip = (hs.s.reverse_orientation != 0) != (hs.s.swaps_handedness != 0) ? RT_TRI_FLIP : 0u;
      int32_t sl = hs.light;
      if (sl <= -2) {  // -2 - k: unlisted emitter k (a quadric of an object definition, placed by an instance)
        const size_t k = (size_t)(-2 - (int64_t)sl);
        if (k >= s->emitters.size()) return fail(RT_ERR_INVALID, "emitter index out of range");
        sl = (int32_t)(s->lights.size() + k);
        if (emitter_prim[k] < 0) emitter_prim[k] = (int64_t)i;
      }
      s->f_meta[i] = rt_tri_meta{hs.material, sl, RT_PRIM_SPHERE | flip, (uint32_t)t};
      continue;
    }
    for (int v = 0; v < 3; ++v) {
      const int32_t vi = s->idx[3 * t + v];
      for (int k = 0; k < 3; ++k) s->f_p[9 * i + 3 * v + k] = s->P[3 * vi + k];
      if (!s->N.empty()) for (int k = 0; k < 3; ++k) s->f_n[9 * i + 3 * v + k] = s->N[3 * vi + k];
      if (!s->S.empty()) for (int k = 0; k < 3; ++k) s->f_s[9 * i + 3 * v + k] = s->S[3 * vi + k];
      if (!s->UV.empty()) for (int k = 0; k < 2; ++k) s->f_uv[6 * i + 2 * v + k] = s->UV[2 * vi + k];
    }
    int32_t tl = s->tri_light[t];
    if (tl <= -2) {  // -2 - k: unlisted emitter k (a written-out instance of an object that holds an AreaLightSource)
      const size_t k = (size_t)(-2 - (int64_t)tl);
      if (k >= s->emitters.size()) return fail(RT_ERR_INVALID, "emitter index out of range");
      tl = (int32_t)(s->lights.size() + k);
      if (emitter_prim[k] < 0) emitter_prim[k] = (int64_t)i;
    }
    s->f_meta[i] = rt_tri_meta{s->tri_mat[t], tl, (uint32_t)s->tri_flags[t], (uint32_t)t};
  }
  s->f_alpha.clear();
  if (!s->tri_alpha.empty() || any_obj_alpha) {
    s->f_alpha.assign(n_all * 2, -1);
    for (size_t o = 0; o < s->objects.size(); ++o) {  // masks on the triangles of instanced objects
      if (obj_prim_base[o] < 0 || s->objects[o].tri_alpha.empty()) continue;
      const auto& ob = s->objects[o];
      for (size_t i = 0; i < ob.n_prims(); ++i) {
        const int32_t t = ob.ordered[i];
        if ((size_t)t >= ob.n_tris()) continue;
        for (int k = 0; k < 2; ++k) {
          const int32_t id = ob.tri_alpha[2 * (size_t)t + k];
          if (id >= (int32_t)s->textures.size()) return fail(RT_ERR_INVALID, "alpha texture out of range");
          if (id >= 0) { s->f_alpha[2 * ((size_t)obj_prim_base[o] + i) + k] = id; s->f_meta[(size_t)obj_prim_base[o] + i].flags |= (k == 0 ? RT_TRI_HAS_ALPHA : RT_TRI_HAS_SHADOW_ALPHA); }
        }
      }
    }
    for (size_t i = 0; i < nt && !s->tri_alpha.empty(); ++i) {
      const int32_t t = s->ordered[i];
      if ((size_t)t >= n_triangles) continue;  // spheres and instances carry no mask
      for (int k = 0; k < 2; ++k) {
        const int32_t id = s->tri_alpha[2 * (size_t)t + k];
        if (id >= (int32_t)s->textures.size()) return fail(RT_ERR_INVALID, "alpha texture out of range");
        if (id >= 0) { s->f_alpha[2 * i + k] = id; s->f_meta[i].flags |= (k == 0 ? RT_TRI_HAS_ALPHA : RT_TRI_HAS_SHADOW_ALPHA); }
      }
    }
  }
  // images
  s->f_images.clear();
  for (const MipLevels& m : s->mips) {
    rt_image im{};
    im.n_levels = (int32_t)m.w.size();
    for (int l = 0; l < im.n_levels; ++l) { im.width[l] = m.w[l]; im.height[l] = m.h[l]; im.offset[l] = m.off[l]; }
    im.texels = m.texels.data(); im.n_texels = m.texels.size() / 3; im.trilinear = m.trilinear; im.max_anisotropy = m.max_aniso; im.wrap = m.wrap;
    s->f_images.push_back(im);
  }
  // lights
  s->f_lights.clear();
  for (HostLight& hl : s->lights) {
    rt_light l = hl.l;
    if (l.kind == RT_LIGHT_DIFFUSE_AREA) {
      if (hl.tri_source <= -2) hl.tri_source = (int)(n_triangles + (size_t)(-2 - hl.tri_source));  // -2 - k: the light sits on sphere k
      if (hl.tri_source < 0 || (size_t)hl.tri_source >= n_geom) return fail(RT_ERR_INVALID, "area light triangle out of range");
      l.prim = leaf_of[hl.tri_source];
      if ((size_t)hl.tri_source >= n_triangles) {
        const rt_sphere& sp = s->spheres[(size_t)hl.tri_source - n_triangles].s;
        l.area = sp.kind == 1 ? sp.phi_max * 0.5f * (sp.radius * sp.radius - sp.inner_radius * sp.inner_radius)  // Disk::area, disk.rs:153-155
                 : sp.kind == 2 ? (sp.z_max - sp.z_min) * sp.radius * sp.phi_max                              // Cylinder::area, cylinder.rs:252-254
                                : sp.phi_max * sp.radius * (sp.z_max - sp.z_min);                             // Sphere::area, sphere.rs:336-338
        if (l.kind == RT_LIGHT_DISTANT || l.kind == RT_LIGHT_INFINITE) l.world_radius = world_radius;
        s->f_lights.push_back(l);
        continue;
      }
      const float* p0 = &s->P[3 * s->idx[3 * hl.tri_source]]; const float* p1 = &s->P[3 * s->idx[3 * hl.tri_source + 1]]; const float* p2 = &s->P[3 * s->idx[3 * hl.tri_source + 2]];
      float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, bq[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      float cxr[3] = {(a[1] * bq[2]) - (a[2] * bq[1]), (a[2] * bq[0]) - (a[0] * bq[2]), (a[0] * bq[1]) - (a[1] * bq[0])};
      l.area = 0.5f * std::sqrt(cxr[0] * cxr[0] + cxr[1] * cxr[1] + cxr[2] * cxr[2]);  // Triangle::area, mesh.rs:588-594
    }
    if (l.kind == RT_LIGHT_DISTANT || l.kind == RT_LIGHT_INFINITE) l.world_radius = world_radius;
    if (l.kind == RT_LIGHT_INFINITE) {
      l.dist_func = hl.dfunc.data(); l.dist_cdf = hl.dcdf.data(); l.dist_func_int = hl.dint.data(); l.marg_func = hl.mfunc.data(); l.marg_cdf = hl.mcdf.data();
    }
    s->f_lights.push_back(l);
  }
  s->n_listed_lights = s->f_lights.size();
  for (size_t k = 0; k < s->emitters.size(); ++k) {  // the emitters no list holds follow the scene's lights
    rt_light l{}; l.kind = RT_LIGHT_DIFFUSE_AREA; l.prim = (int32_t)std::max<int64_t>(emitter_prim[k], 0); l.two_sided = s->emitters[k].two_sided; l.area = 1.0f; l.image = -1;
    for (int c = 0; c < 3; ++c) l.rgb[c] = s->emitters[k].rgb[c];
    s->f_lights.push_back(l);
  }
  s->committed = true; s->commit_gen += 1;
  return RT_OK;
}

rt_scene_desc make_desc(rtxh_scene* s) {
  rt_scene_desc d{};
  d.n_nodes = (uint32_t)s->nodes.size(); d.nodes = s->nodes.data();
  if (!s->f_instances.empty()) {
    d.n_top_nodes = d.n_nodes; d.n_nodes = (uint32_t)s->f_nodes.size(); d.nodes = s->f_nodes.data();
    d.n_instances = (uint32_t)s->f_instances.size(); d.instances = s->f_instances.data();
  }
  d.n_tris = (uint32_t)s->n_tris(); d.tri_p = s->f_p.data();
  d.tri_n = s->f_n.empty() ? nullptr : s->f_n.data(); d.tri_uv = s->f_uv.empty() ? nullptr : s->f_uv.data(); d.tri_s = s->f_s.empty() ? nullptr : s->f_s.data();
  d.tri_meta = s->f_meta.data(); d.tri_alpha = s->f_alpha.empty() ? nullptr : s->f_alpha.data();
  d.n_tris = (uint32_t)s->n_prims(); d.n_spheres = (uint32_t)s->f_spheres.size();
  if (!s->f_instances.empty()) { d.n_top_prims = d.n_tris; d.n_tris = (uint32_t)(s->f_meta.size()); } d.spheres = s->f_spheres.empty() ? nullptr : s->f_spheres.data();
  d.n_textures = (uint32_t)s->textures.size(); d.textures = s->textures.data();
  d.n_images = (uint32_t)s->f_images.size(); d.images = s->f_images.data();
  d.n_materials = (uint32_t)s->materials.size(); d.materials = s->materials.data();
  d.n_lights = (uint32_t)s->n_listed_lights; d.n_unlisted_lights = (uint32_t)(s->f_lights.size() - s->n_listed_lights); d.lights = s->f_lights.data();
  return d;
}

struct CamFilm { rt_camera cam; rt_film_desc film; };
// PerspectiveCamera::new (camera.rs:30-72) with create()'s default screen window (:86-97); Film::new (film.rs:58-115)
int setup_camera_film(const rtxh_render_params* p, CamFilm& out) {
  if (p->xres <= 0 || p->yres <= 0) return fail(RT_ERR_INVALID, "bad resolution");
  float frame = (float)p->xres / (float)p->yres;
  float sw[4];
  if (frame > 1.0f) { sw[0] = -frame; sw[1] = frame; sw[2] = -1.0f; sw[3] = 1.0f; }
  else { sw[0] = -1.0f; sw[1] = 1.0f; sw[2] = -1.0f / frame; sw[3] = 1.0f / frame; }
  if (p->screen_window[1] > p->screen_window[0]) memcpy(sw, p->screen_window, 16);  // "screenwindow" / "frameaspectratio" (camera.rs:86-107)
  Xf camera_to_screen = xf_perspective(p->fov, 1e-2f, 1000.0f);
  Xf screen_to_raster = xf_mul(xf_mul(xf_scale((float)p->xres, (float)p->yres, 1.0f), xf_scale(1.0f / (sw[1] - sw[0]), 1.0f / (sw[2] - sw[3]), 1.0f)),
                               xf_translate(-sw[0], -sw[3], 0.0f));
  Xf raster_to_camera = xf_mul(xf_inverse(camera_to_screen), xf_inverse(screen_to_raster));
  memcpy(out.cam.raster_to_camera, raster_to_camera.m.a, 64);
  memcpy(out.cam.camera_to_world, p->cam_to_world, 64);
  const float o[3] = {0, 0, 0}, ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
  float po[3], px[3], py[3];
  xf_point(raster_to_camera.m, o, po); xf_point(raster_to_camera.m, ex, px); xf_point(raster_to_camera.m, ey, py);
  for (int k = 0; k < 3; ++k) { out.cam.dx_camera[k] = px[k] - po[k]; out.cam.dy_camera[k] = py[k] - po[k]; }
  out.cam.lens_radius = p->lens_radius; out.cam.focal_distance = p->focal_distance;
  // film
  rt_film_desc& f = out.film;
  int ax = f2i(std::ceil((float)p->xres * p->crop[0])), ay = f2i(std::ceil((float)p->yres * p->crop[2]));
  int bx = f2i(std::ceil((float)p->xres * p->crop[1])), by = f2i(std::ceil((float)p->yres * p->crop[3]));
  f.cropped_pixel_bounds[0] = std::min(ax, bx); f.cropped_pixel_bounds[1] = std::min(ay, by);
  f.cropped_pixel_bounds[2] = std::max(ax, bx); f.cropped_pixel_bounds[3] = std::max(ay, by);
  const float xw = p->filter_params[0], yw = p->filter_params[1];
  f.filter_radius[0] = xw; f.filter_radius[1] = yw;
  for (int y = 0; y < 16; ++y) {
    float fy = ((float)y + 0.5f) * (yw / 16.0f);
    for (int x = 0; x < 16; ++x) { float fx = ((float)x + 0.5f) * (xw / 16.0f); f.filter_table[y * 16 + x] = filter_eval(p->filter_kind, p->filter_params, fx, fy); }
  }
  f.max_sample_luminance = p->max_sample_luminance;
  // Film::get_sample_bounds (film.rs:249-257)
  float x0 = std::floor((float)f.cropped_pixel_bounds[0] + 0.5f - xw), y0 = std::floor((float)f.cropped_pixel_bounds[1] + 0.5f - yw);
  float x1 = std::ceil((float)f.cropped_pixel_bounds[2] - 0.5f + xw), y1 = std::ceil((float)f.cropped_pixel_bounds[3] - 0.5f + yw);
  f.sample_bounds[0] = f2i(pmin(x0, x1)); f.sample_bounds[1] = f2i(pmin(y0, y1)); f.sample_bounds[2] = f2i(pmax(x0, x1)); f.sample_bounds[3] = f2i(pmax(y0, y1));
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rtxh_last_error(void) { return g_err.empty() ? rt_last_error() : g_err.c_str(); }
rtxh_scene* rtxh_scene_new(void) { return new rtxh_scene(); }
void rtxh_scene_free(rtxh_scene* s) { if (!s) return; if (s->dev) rt_scene_destroy(s->dev); if (s->multi) rt_multi_destroy(s->multi); delete s; }

int rtxh_scene_set_mesh(rtxh_scene* s, const float* P, int32_t nv, const int32_t* idx, int32_t nt, const float* N, const float* UV, const float* S,
                        const int32_t* tri_material, const int32_t* tri_light, const uint8_t* tri_flags) {
  if (!s || !P || !idx || nv <= 0 || nt <= 0 || !tri_material || !tri_light || !tri_flags) return fail(RT_ERR_INVALID, "bad mesh arguments");
  s->P.assign(P, P + 3 * (size_t)nv);
  s->N.clear(); s->UV.clear(); s->S.clear();
  if (N) s->N.assign(N, N + 3 * (size_t)nv);
  if (UV) s->UV.assign(UV, UV + 2 * (size_t)nv);
  if (S) s->S.assign(S, S + 3 * (size_t)nv);
  s->idx.assign(idx, idx + 3 * (size_t)nt);
  for (int32_t v : s->idx) if (v < 0 || v >= nv) return fail(RT_ERR_INVALID, "vertex index out of range");
  s->tri_mat.assign(tri_material, tri_material + nt); s->tri_light.assign(tri_light, tri_light + nt); s->tri_flags.assign(tri_flags, tri_flags + nt);
  for (int32_t t = 0; t < nt; ++t) {
    uint8_t f = tri_flags[t];
    if (((f & RT_TRI_HAS_N) && !N) || ((f & RT_TRI_HAS_UV) && !UV) || ((f & RT_TRI_HAS_S) && !S)) return fail(RT_ERR_INVALID, "flags promise a missing attribute");
  }
  s->committed = false;
  return RT_OK;
}

// Shape "sphere" (Sphere::create sphere.rs:53-68 + Sphere::new :29-51). Returns the sphere's index k; an area light on it is added with tri = -2 - k.
int rtxh_scene_add_sphere(rtxh_scene* s, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max, int32_t reverse_orientation,
                          int32_t material, int32_t light) {
  return rtxh_scene_add_quadric(s, 0, o2w16, w2o16, radius, z_min, z_max, phi_max, reverse_orientation, material, light);
}
// the record of a quadric (Sphere::new sphere.rs:29-51, Disk::new disk.rs:25-46, Cylinder::create cylinder.rs:26-46) under its object_to_world
static int make_quadric(int32_t kind, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max, int32_t reverse_orientation,
                        int32_t material, int32_t light, rtxh_scene::HostSphere& hs) {
  if (kind < 0 || kind > 2) return fail(RT_ERR_INVALID, "unknown quadric kind");
  if (!o2w16 || !w2o16) return fail(RT_ERR_INVALID, "bad sphere arguments");
  hs = rtxh_scene::HostSphere{}; rt_sphere& sp = hs.s;
  memcpy(sp.o2w, o2w16, 64); memcpy(sp.w2o, w2o16, 64);
  auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };  // lib.rs:264-275
  sp.radius = radius; sp.kind = kind;
  if (kind == 0) {  // Sphere::new, sphere.rs:29-51
    sp.z_min = clampf(std::fmin(z_min, z_max), -radius, radius); sp.z_max = clampf(std::fmax(z_min, z_max), -radius, radius);
    sp.theta_min = std::acos(clampf(std::fmin(z_min, z_max) / radius, -1.0f, 1.0f)); sp.theta_max = std::acos(clampf(std::fmax(z_min, z_max) / radius, -1.0f, 1.0f));
  } else if (kind == 1) {  // Disk::new(height = z_min, inner_radius = z_max), disk.rs:25-46 (it asserts radius > 0, inner_radius >= 0, phi_max > 0)
    if (!(radius > 0.0f) || !(z_max >= 0.0f) || !(phi_max > 0.0f)) return fail(RT_ERR_INVALID, "disk needs radius > 0, innerradius >= 0, phimax > 0 (the reference asserts it)");
    sp.height = z_min; sp.inner_radius = z_max; sp.z_min = sp.z_max = z_min;
  } else { sp.z_min = z_min; sp.z_max = z_max; }  // Cylinder::create keeps z_min / z_max as given, cylinder.rs:26-46
  sp.phi_max = clampf(phi_max, 0.0f, 360.0f) * (3.14159265358979323846f / 180.0f);  // f32::to_radians
  sp.reverse_orientation = reverse_orientation ? 1 : 0;
  const float* m = o2w16;  // Transform::swaps_handedness, transform.rs:255-261
  const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
  sp.swaps_handedness = det < 0.0f ? 1 : 0;
  hs.material = material; hs.light = light;
  return RT_OK;
}
int rtxh_scene_add_quadric(rtxh_scene* s, int32_t kind, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max,
                           int32_t reverse_orientation, int32_t material, int32_t light) {
  if (!s) return fail(RT_ERR_INVALID, "bad sphere arguments");
  rtxh_scene::HostSphere hs;
  const int rc = make_quadric(kind, o2w16, w2o16, radius, z_min, z_max, phi_max, reverse_orientation, material, light, hs);
  if (rc != RT_OK) return rc;
  s->spheres.push_back(hs);
  s->committed = false;
  return (int)s->spheres.size() - 1;
}
// A quadric INSIDE an object definition (api.rs:1019-1051: ObjectBegin collects every primitive): o2w / w2o = the CTM inside the definition (object space). emitter: -1, or
// the index of an unlisted emitter (rtxh_scene_add_emitter: a shape under an AreaLightSource inside an object definition is in no light list, api.rs:954-964).
// Returns the quadric's primitive id inside the object (after the object's triangles).
int rtxh_object_add_quadric(rtxh_scene* s, int32_t object, int32_t kind, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max,
                            int32_t reverse_orientation, int32_t material, int32_t emitter) {
  if (!s || object < 0 || (size_t)object >= s->objects.size()) return fail(RT_ERR_INVALID, "bad object");
  if (emitter >= (int32_t)s->emitters.size()) return fail(RT_ERR_INVALID, "emitter index out of range");
  rtxh_scene::HostSphere hs;
  const int rc = make_quadric(kind, o2w16, w2o16, radius, z_min, z_max, phi_max, reverse_orientation, material, emitter >= 0 ? -2 - emitter : -1, hs);
  if (rc != RT_OK) return rc;
  auto& o = s->objects[(size_t)object];
  o.spheres.push_back(hs); s->committed = false;
  return (int)(o.n_prims() - 1);
}
// alpha / shadowalpha float textures of an object's triangles (2 per triangle, -1 = none; NULL clears)
int rtxh_object_set_alpha(rtxh_scene* s, int32_t object, const int32_t* tri_alpha2) {
  if (!s || object < 0 || (size_t)object >= s->objects.size()) return fail(RT_ERR_INVALID, "bad object");
  auto& o = s->objects[(size_t)object];
  if (!tri_alpha2) o.tri_alpha.clear(); else o.tri_alpha.assign(tri_alpha2, tri_alpha2 + 2 * o.n_tris());
  s->committed = false;
  return RT_OK;
}

int rtxh_scene_add_object(rtxh_scene* s, const float* P, int32_t nv, const int32_t* idx, int32_t nt, const float* N, const float* UV, const float* S,
                          const int32_t* tri_material, const uint8_t* tri_flags) {
  if (!s || nv < 0 || nt < 0 || (nt > 0 && (!P || !idx || !tri_material || !tri_flags || nv <= 0))) return fail(RT_ERR_INVALID, "bad object mesh");
  rtxh_scene::HostObject o;
  if (nt == 0) { s->objects.push_back(std::move(o)); s->committed = false; return (int)s->objects.size() - 1; }  // (an object of quadrics only: rtxh_object_add_quadric)
  o.P.assign(P, P + 3 * (size_t)nv); o.idx.assign(idx, idx + 3 * (size_t)nt);
  if (N) o.N.assign(N, N + 3 * (size_t)nv); if (UV) o.UV.assign(UV, UV + 2 * (size_t)nv); if (S) o.S.assign(S, S + 3 * (size_t)nv);
  o.tri_mat.assign(tri_material, tri_material + nt); o.tri_flags.assign(tri_flags, tri_flags + nt);
  for (int32_t i = 0; i < 3 * nt; ++i) if (idx[i] < 0 || idx[i] >= nv) return fail(RT_ERR_INVALID, "object vertex index out of range");
  for (int32_t i = 0; i < nt; ++i) {
    const uint8_t f = tri_flags[i];
    if (((f & RT_TRI_HAS_N) && !N) || ((f & RT_TRI_HAS_UV) && !UV) || ((f & RT_TRI_HAS_S) && !S)) return fail(RT_ERR_INVALID, "object triangle flags name a missing attribute array");
  }
  s->objects.push_back(std::move(o)); s->committed = false;
  return (int)s->objects.size() - 1;
}
int rtxh_scene_add_emitter(rtxh_scene* s, const float* rgb, int32_t two_sided) {
  if (!s || !rgb) return fail(RT_ERR_INVALID, "bad emitter");
  rtxh_scene::Emitter e; e.rgb[0] = rgb[0]; e.rgb[1] = rgb[1]; e.rgb[2] = rgb[2]; e.two_sided = two_sided ? 1 : 0;
  s->emitters.push_back(e); s->committed = false;
  return (int)s->emitters.size() - 1;
}
int rtxh_scene_object_emitters(rtxh_scene* s, int32_t object, const int32_t* tri_emitter) {
  if (!s || !tri_emitter || object < 0 || (size_t)object >= s->objects.size()) return fail(RT_ERR_INVALID, "bad object emitters");
  auto& o = s->objects[(size_t)object];
  for (size_t t = 0; t < o.n_tris(); ++t) if (tri_emitter[t] >= (int32_t)s->emitters.size()) return fail(RT_ERR_INVALID, "emitter index out of range");
  o.tri_emit.assign(tri_emitter, tri_emitter + o.n_tris()); s->committed = false;
  return RT_OK;
}
int rtxh_scene_add_instance(rtxh_scene* s, int32_t object, const float* o2w16, const float* w2o16) {
  if (!s || !o2w16 || !w2o16 || object < 0 || (size_t)object >= s->objects.size()) return fail(RT_ERR_INVALID, "bad object instance");
  rtxh_scene::HostInstance in; in.object = object; memcpy(in.o2w, o2w16, 64); memcpy(in.w2o, w2o16, 64);
  s->instances.push_back(in); s->committed = false;
  return (int)s->instances.size() - 1;
}
int rtxh_scene_set_alpha(rtxh_scene* s, const int32_t* tri_alpha2) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  if (!tri_alpha2) s->tri_alpha.clear(); else s->tri_alpha.assign(tri_alpha2, tri_alpha2 + 2 * s->n_tris());
  s->committed = false;
  return RT_OK;
}

// MIPMap::resample_weights / lanczos (mipmap.rs:362-408)
static float mip_lanczos(float f) {
  float tau = 2.0f, x = std::fabs(f);
  if (x < 1e-5f) return 1.0f;
  if (x > 1.0f) return 0.0f;
  x *= 3.14159265358979323846f;
  float s = std::sin(x * tau) / (x * tau);
  float l = std::sin(x) / x;
  return s * l;
}
struct MipResampleWeight { int first_texel; float w[4]; };
static std::vector<MipResampleWeight> mip_resample_weights(int old_res, int new_res) {
  std::vector<MipResampleWeight> wt((size_t)new_res);
  const float filter_width = 2.0f;
  for (int i = 0; i < new_res; ++i) {
    float center = ((float)i + 0.5f) * (float)old_res / (float)new_res;
    float first = std::floor((center - filter_width) + 0.5f);
    float w[4];
    for (int j = 0; j < 4; ++j) { float pos = first + (float)j + 0.5f; w[j] = mip_lanczos((pos - center) / filter_width); }
    float inv = 1.0f / (w[0] + w[1] + w[2] + w[3]);
    wt[(size_t)i].first_texel = (int)first;  // |first| < 2^24: exact
    for (int j = 0; j < 4; ++j) wt[(size_t)i].w[j] = w[j] * inv;
  }
  return wt;
}
static int round_up_pow2(int v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }  // lib.rs:215-224

static thread_local bool g_device_ingest = false;  // per calling thread: two threads building scenes do not see each other's choice
void rtxh_set_device_ingest(int32_t on) { g_device_ingest = on != 0; }

// MIPMap::new with the arithmetic on the GPU (rt_mip_build): the host only supplies the Lanczos taps and the level geometry
static int add_mipmap_device(rtxh_scene* s, int32_t w, int32_t h, const float* rgb_in, int32_t trilinear, float max_aniso, int32_t wrap) {
  MipLevels m; m.trilinear = trilinear; m.wrap = wrap; m.max_aniso = max_aniso;
  const int px = round_up_pow2(w), py = round_up_pow2(h);
  std::vector<int32_t> sf, tf; std::vector<float> sw, tw;
  if (px != w || py != h) {
    auto taps = [](int old_res, int new_res, std::vector<int32_t>& first, std::vector<float>& wts) {
      const std::vector<MipResampleWeight> wt = mip_resample_weights(old_res, new_res);
      first.resize(wt.size()); wts.resize(wt.size() * 4);
      for (size_t i = 0; i < wt.size(); ++i) { first[i] = wt[i].first_texel; for (int j = 0; j < 4; ++j) wts[4 * i + j] = wt[i].w[j]; }
    };
    taps(w, px, sf, sw); taps(h, py, tf, tw);
  }
  const int n_levels = 1 + (int)f2usz(std::log2((float)std::max(px, py)));
  uint64_t off = 0;
  for (int i = 0; i < n_levels; ++i) {
    const int lw = i == 0 ? px : std::max(1, m.w[i - 1] / 2), lh = i == 0 ? py : std::max(1, m.h[i - 1] / 2);
    m.w.push_back(lw); m.h.push_back(lh); m.off.push_back(off); off += (uint64_t)lw * lh;
  }
  m.texels.assign((size_t)off * 3, 0.0f);
  const int rc = rt_mip_build(rgb_in, w, h, px, py, sf.empty() ? nullptr : sf.data(), sw.empty() ? nullptr : sw.data(), tf.empty() ? nullptr : tf.data(), tw.empty() ? nullptr : tw.data(),
                              wrap, n_levels, m.w.data(), m.h.data(), m.off.data(), m.texels.data());
  if (rc != RT_OK) { g_err.clear(); return rc; }
  s->mips.push_back(std::move(m));
  s->committed = false;
  return (int)s->mips.size() - 1;
}

int rtxh_scene_add_mipmap(rtxh_scene* s, int32_t w, int32_t h, const float* rgb_in, int32_t trilinear, float max_aniso, int32_t wrap) {
  if (!s || !rgb_in || w <= 0 || h <= 0 || w > 65536 || h > 65536) return fail(RT_ERR_INVALID, "bad image");
  if (g_device_ingest) return add_mipmap_device(s, w, h, rgb_in, trilinear, max_aniso, wrap);
  MipLevels m; m.trilinear = trilinear; m.wrap = wrap; m.max_aniso = max_aniso;
  std::vector<float> resampled;
  const float* rgb = rgb_in;
  if ((w & (w - 1)) || (h & (h - 1))) {
    // MIPMap::new, mipmap.rs:75-139: 4-tap Lanczos zoom to the next powers of two, first along s (rows < h only), then along t
    const int px = round_up_pow2(w), py = round_up_pow2(h);
    auto wrap_index = [&](long i, long n) -> long {
      if (wrap == 0) { long r = i % n; return r < 0 ? r + n : r; }   // Repeat
      if (wrap == 2) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);      // Clamp
      return i;                                                        // Black
    };
    resampled.assign((size_t)px * py * 3, 0.0f);
    std::vector<MipResampleWeight> sw = mip_resample_weights(w, px);
    for (int t = 0; t < h; ++t)
      for (int sx = 0; sx < px; ++sx)
        for (int j = 0; j < 4; ++j) {
          long o = wrap_index((long)sw[(size_t)sx].first_texel + j, w);
          if (o >= 0 && o < w)
            for (int k = 0; k < 3; ++k) resampled[3 * ((size_t)t * px + sx) + k] += rgb_in[3 * ((size_t)t * w + o) + k] * sw[(size_t)sx].w[j];
        }
    std::vector<MipResampleWeight> tw = mip_resample_weights(h, py);
    std::vector<float> work((size_t)py * 3);
    for (int sx = 0; sx < px; ++sx) {
      for (int t = 0; t < py; ++t) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < 4; ++j) {
          long o = wrap_index((long)tw[(size_t)t].first_texel + j, h);
          if (o >= 0 && o < h)
            for (int k = 0; k < 3; ++k) acc[k] += resampled[3 * ((size_t)o * px + sx) + k] * tw[(size_t)t].w[j];
        }
        for (int k = 0; k < 3; ++k) work[3 * (size_t)t + k] = acc[k];
      }
      for (int t = 0; t < py; ++t)
        for (int k = 0; k < 3; ++k) { float v = work[3 * (size_t)t + k]; resampled[3 * ((size_t)t * px + sx) + k] = v < 0.0f ? 0.0f : v; }  // clamp(0, inf), lib.rs:264-275
    }
    w = px; h = py; rgb = resampled.data();
  }
  // MIPMap::new (mipmap.rs:158-187): n_levels = 1 + log2(max(res)) as usize; each level = box filter of 4 texels of the finer one
  int n_levels = 1 + (int)f2usz(std::log2((float)std::max(w, h)));
  m.w.push_back(w); m.h.push_back(h); m.off.push_back(0);
  m.texels.assign(rgb, rgb + 3 * (size_t)w * h);
  for (int i = 1; i < n_levels; ++i) {
    int sr = std::max(1, m.w[i - 1] / 2), tr = std::max(1, m.h[i - 1] / 2);
    uint64_t off = m.texels.size() / 3;
    std::vector<float> lvl((size_t)sr * tr * 3);
    for (int t = 0; t < tr; ++t)
      for (int sx = 0; sx < sr; ++sx) {
        float a[3], b[3], c[3], d[3];
        mip_texel(m, i - 1, 2 * sx, 2 * t, a); mip_texel(m, i - 1, 2 * sx + 1, 2 * t, b); mip_texel(m, i - 1, 2 * sx, 2 * t + 1, c); mip_texel(m, i - 1, 2 * sx + 1, 2 * t + 1, d);
        for (int k = 0; k < 3; ++k) lvl[3 * ((size_t)t * sr + sx) + k] = (a[k] + b[k] + c[k] + d[k]) * 0.25f;
      }
    m.texels.insert(m.texels.end(), lvl.begin(), lvl.end());
    m.w.push_back(sr); m.h.push_back(tr); m.off.push_back(off);
  }
  s->mips.push_back(std::move(m));
  s->committed = false;
  return (int)s->mips.size() - 1;
}

int rtxh_scene_add_texture(rtxh_scene* s, int32_t kind, const float* v, int32_t tex1, int32_t tex2, int32_t amount, int32_t mip, const float* mapping) {
  if (!s || !v) return fail(RT_ERR_INVALID, "bad texture arguments");
  rt_texture t{}; t.kind = kind; t.value[0] = v[0]; t.value[1] = v[1]; t.value[2] = v[2]; t.tex1 = tex1; t.tex2 = tex2; t.amount = amount; t.image = mip;
  t.mapping[0] = mapping ? mapping[0] : 1.0f; t.mapping[1] = mapping ? mapping[1] : 1.0f; t.mapping[2] = mapping ? mapping[2] : 0.0f; t.mapping[3] = mapping ? mapping[3] : 0.0f;
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}
int rtxh_scene_add_material(rtxh_scene* s, int32_t kind, const int32_t* slots, int32_t remap, int32_t bump) {
  if (!s || !slots) return fail(RT_ERR_INVALID, "bad material arguments");
  rt_material m{}; m.kind = kind; m.remap_roughness = remap; m.bump = bump;
  for (int k = 0; k < RT_N_SLOTS; ++k) m.slot[k] = slots[k];
  s->materials.push_back(m);
  return (int)s->materials.size() - 1;
}
int rtxh_scene_add_light(rtxh_scene* s, int32_t kind, int32_t tri, const float* rgb, int32_t two_sided, const float* vec, int32_t mip, const float* l2w, const float* w2l) {
  if (!s || !rgb) return fail(RT_ERR_INVALID, "bad light arguments");
  HostLight hl; memset(&hl.l, 0, sizeof(hl.l));
  rt_light& l = hl.l;
  l.kind = kind; l.prim = -1; hl.tri_source = tri; l.rgb[0] = rgb[0]; l.rgb[1] = rgb[1]; l.rgb[2] = rgb[2]; l.two_sided = two_sided; l.image = mip;
  if (vec) { l.vec[0] = vec[0]; l.vec[1] = vec[1]; l.vec[2] = vec[2]; }
  if (kind == RT_LIGHT_DISTANT) {  // DistantLight::new normalises the direction (distant.rs:27)
    float len = std::sqrt(l.vec[0] * l.vec[0] + l.vec[1] * l.vec[1] + l.vec[2] * l.vec[2]);
    l.vec[0] /= len; l.vec[1] /= len; l.vec[2] /= len;
  }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { l.l2w[4 * r + c] = l2w ? l2w[4 * r + c] : (r == c ? 1.0f : 0.0f); l.w2l[4 * r + c] = w2l ? w2l[4 * r + c] : (r == c ? 1.0f : 0.0f); }
  if (kind == RT_LIGHT_INFINITE) {
    if (mip < 0 || (size_t)mip >= s->mips.size()) return fail(RT_ERR_INVALID, "infinite light needs an environment map");
    // InfiniteAreaLight::new (infinite.rs:78-101): luminance * sin(theta) at twice the map resolution
    const MipLevels& m = s->mips[mip];
    const int width = 2 * m.w[0], height = 2 * m.h[0];
    const float filter = 0.5f / std::fmin((float)width, (float)height);
    const float pi = 3.14159265358979323846f;
    if (g_device_ingest) {  // same tables from the GPU (rt_env_distribution); sin(theta) and the level choice of the constant filter width stay on the host
      std::vector<float> sin_theta((size_t)height);
      for (int v = 0; v < height; ++v) sin_theta[(size_t)v] = std::sin(pi * ((float)v + 0.5f) / (float)height);
      const int nl = (int)m.w.size();
      const float level = (float)nl - 1.0f + std::log2(std::fmax(filter, 1e-8f));
      int mode, il = 0; float delta = 0.0f;
      if (level < 0.0f) mode = 0; else if (level >= (float)nl - 1.0f) mode = 1; else { mode = 2; const float fl = std::floor(level); il = (int)f2usz(fl); delta = level - fl; }
      rt_image im{}; im.n_levels = nl;
      for (int lv = 0; lv < nl; ++lv) { im.width[lv] = m.w[lv]; im.height[lv] = m.h[lv]; im.offset[lv] = m.off[lv]; }
      im.texels = m.texels.data(); im.n_texels = m.texels.size() / 3; im.trilinear = m.trilinear; im.max_anisotropy = m.max_aniso; im.wrap = m.wrap;
      hl.dfunc.assign((size_t)width * height, 0.0f); hl.dcdf.assign((size_t)height * (width + 1), 0.0f); hl.dint.assign((size_t)height, 0.0f); hl.mcdf.assign((size_t)height + 1, 0.0f);
      float mint = 0.0f;
      const int rc = rt_env_distribution(&im, width, height, mode, il, delta, sin_theta.data(), hl.dfunc.data(), hl.dcdf.data(), hl.dint.data(), hl.mcdf.data(), &mint);
      if (rc != RT_OK) { g_err.clear(); return rc; }
      hl.mfunc = hl.dint; l.marg_func_int = mint; l.dist_nu = width; l.dist_nv = height;
      s->lights.push_back(std::move(hl));
      s->committed = false;
      return (int)s->lights.size() - 1;
    }
    std::vector<float> img((size_t)width * height);
    for (int v = 0; v < height; ++v) {
      float vp = ((float)v + 0.5f) / (float)height;
      float sin_theta = std::sin(pi * ((float)v + 0.5f) / (float)height);
      for (int u = 0; u < width; ++u) {
        float up = ((float)u + 0.5f) / (float)width;
        float c[3]; mip_lookup(m, up, vp, filter, c);
        img[(size_t)v * width + u] = (0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]) * sin_theta;
      }
    }
    // Distribution2D::new (distribution2d.rs:11-27)
    hl.dfunc = img; hl.dcdf.resize((size_t)height * (width + 1)); hl.dint.resize(height); hl.mfunc.resize(height);
    for (int v = 0; v < height; ++v) {
      Dist1 d; dist1_init(d, &img[(size_t)v * width], width);
      memcpy(&hl.dcdf[(size_t)v * (width + 1)], d.cdf.data(), (size_t)(width + 1) * 4);
      hl.dint[v] = d.func_int; hl.mfunc[v] = d.func_int;
    }
    Dist1 md; dist1_init(md, hl.mfunc.data(), height);
    hl.mcdf = md.cdf; l.marg_func_int = md.func_int; l.dist_nu = width; l.dist_nv = height;
  }
  s->lights.push_back(std::move(hl));
  s->committed = false;
  return (int)s->lights.size() - 1;
}

int rtxh_scene_commit(rtxh_scene* s, int32_t max_prims_per_node) { if (!s) return fail(RT_ERR_INVALID, "null scene"); g_err.clear(); return commit_scene(s, max_prims_per_node); }
int rtxh_scene_commit_device_bvh(rtxh_scene* s, int32_t max_prims_per_node, float* ms_device) { if (!s) return fail(RT_ERR_INVALID, "null scene"); g_err.clear(); return commit_scene_device(s, max_prims_per_node, ms_device); }

int rtxh_scene_upload(rtxh_scene* s, int32_t device) {
  if (!s || !s->committed) return fail(RT_ERR_INVALID, "scene not committed");
  g_err.clear();
  if (s->dev) { rt_scene_destroy(s->dev); s->dev = nullptr; }
  rt_scene_desc d = make_desc(s);
  return rt_scene_create(&d, device, &s->dev);
}

int rtxh_scene_bvh_sizes(rtxh_scene* s, int32_t* n_nodes, int32_t* n_prims) { *n_nodes = (int32_t)s->nodes.size(); *n_prims = (int32_t)s->ordered.size(); return RT_OK; }
int rtxh_scene_bvh_get(rtxh_scene* s, float* bounds6, uint32_t* offset, uint16_t* n_prims, uint8_t* axis, int32_t* ordered) {
  for (size_t i = 0; i < s->nodes.size(); ++i) {
    const rt_bvh_node& n = s->nodes[i];
    for (int k = 0; k < 3; ++k) { bounds6[6 * i + k] = n.bmin[k]; bounds6[6 * i + 3 + k] = n.bmax[k]; }
    offset[i] = n.offset; n_prims[i] = n.n_prims; axis[i] = n.axis;
  }
  memcpy(ordered, s->ordered.data(), s->ordered.size() * 4);
  return RT_OK;
}
int rtxh_camera_film_setup(const rtxh_render_params* p, float* r2c, float* dxdy, float* table, int32_t* sb, int32_t* cropped) {
  CamFilm cf; int rc = setup_camera_film(p, cf); if (rc != RT_OK) return rc;
  memcpy(r2c, cf.cam.raster_to_camera, 64);
  for (int k = 0; k < 3; ++k) { dxdy[k] = cf.cam.dx_camera[k]; dxdy[3 + k] = cf.cam.dy_camera[k]; }
  memcpy(table, cf.film.filter_table, 1024); memcpy(sb, cf.film.sample_bounds, 16); memcpy(cropped, cf.film.cropped_pixel_bounds, 16);
  return RT_OK;
}
int rtxh_mip_level(rtxh_scene* s, int32_t mip, int32_t level, int32_t* w, int32_t* h, float* rgb_out) {
  if (!s || mip < 0 || (size_t)mip >= s->mips.size()) return fail(RT_ERR_INVALID, "bad mip index");
  const MipLevels& m = s->mips[mip];
  if (level < 0 || (size_t)level >= m.w.size()) return fail(RT_ERR_INVALID, "bad level");
  *w = m.w[level]; *h = m.h[level];
  if (rgb_out) memcpy(rgb_out, &m.texels[3 * m.off[level]], (size_t)m.w[level] * m.h[level] * 12);
  return (int)m.w.size();
}
// Transform::look_at (transform.rs:119-154): m = world->camera, m_inv = camera->world
int rtxh_scene_inspect(rtxh_scene* s, int32_t table, void* out, uint64_t capacity_bytes, uint64_t* n_items) {
  if (!s || !n_items) return fail(RT_ERR_INVALID, "null argument");
  const void* src = nullptr; size_t item = 0, n = 0;
  std::vector<rtxh_light_info> li; std::vector<rtxh_instance_info> ii;
  if (table >= RTXH_TABLE_OBJECT_EXTRA_BASE) {
    const size_t k = (size_t)(table - RTXH_TABLE_OBJECT_EXTRA_BASE) / 2; const int j = (table - RTXH_TABLE_OBJECT_EXTRA_BASE) % 2;
    if (k >= s->objects.size()) return fail(RT_ERR_INVALID, "unknown object table");
    const auto& o = s->objects[k];
    if (j == 0) { src = o.spheres.data(); item = sizeof(rtxh_scene::HostSphere); n = o.spheres.size(); }
    else { src = o.tri_alpha.data(); item = 8; n = o.tri_alpha.size() / 2; }
    *n_items = n;
    if (out) { if (capacity_bytes < n * item) return fail(RT_ERR_INVALID, "buffer too small"); if (n) memcpy(out, src, n * item); }
    return RT_OK;
  }
  if (table >= RTXH_TABLE_OBJECT_BASE) {
    const size_t k = (size_t)(table - RTXH_TABLE_OBJECT_BASE) / 8; const int j = (table - RTXH_TABLE_OBJECT_BASE) % 8;
    if (k >= s->objects.size() || j > 7) return fail(RT_ERR_INVALID, "unknown object table");
    const auto& o = s->objects[k];
    switch (j) {
      case 0: src = o.P.data(); item = 12; n = o.P.size() / 3; break;
      case 1: src = o.N.data(); item = 12; n = o.N.size() / 3; break;
      case 2: src = o.UV.data(); item = 8; n = o.UV.size() / 2; break;
      case 3: src = o.S.data(); item = 12; n = o.S.size() / 3; break;
      case 4: src = o.idx.data(); item = 12; n = o.idx.size() / 3; break;
      case 5: src = o.tri_mat.data(); item = 4; n = o.tri_mat.size(); break;
      case 7: src = o.tri_emit.data(); item = 4; n = o.tri_emit.size(); break;
      default: src = o.tri_flags.data(); item = 1; n = o.tri_flags.size(); break;
    }
    *n_items = n;
    if (out) { if (capacity_bytes < n * item) return fail(RT_ERR_INVALID, "buffer too small"); if (n) memcpy(out, src, n * item); }
    return RT_OK;
  }
  switch (table) {
    case RTXH_TABLE_INSTANCES:
      for (const auto& in : s->instances) { rtxh_instance_info i; i.object = in.object; memcpy(i.o2w, in.o2w, 64); memcpy(i.w2o, in.w2o, 64); ii.push_back(i); }
      src = ii.data(); item = sizeof(rtxh_instance_info); n = ii.size(); break;
    case RTXH_TABLE_EMITTERS: src = s->emitters.data(); item = sizeof(rtxh_emitter_info); n = s->emitters.size(); break;
    case RTXH_TABLE_QUADRICS: src = s->spheres.data(); item = sizeof(rtxh_scene::HostSphere); n = s->spheres.size(); break;
    case RTXH_TABLE_TEXTURES: src = s->textures.data(); item = sizeof(rt_texture); n = s->textures.size(); break;
    case RTXH_TABLE_MATERIALS: src = s->materials.data(); item = sizeof(rt_material); n = s->materials.size(); break;
    case RTXH_TABLE_LIGHTS:
      for (const HostLight& h : s->lights) {
        rtxh_light_info i; memset(&i, 0, sizeof i);
        i.kind = h.l.kind; i.tri = h.tri_source; memcpy(i.rgb, h.l.rgb, 12); i.two_sided = h.l.two_sided; memcpy(i.vec, h.l.vec, 12); i.mip = h.l.image;
        memcpy(i.l2w, h.l.l2w, 48); memcpy(i.w2l, h.l.w2l, 48);
        li.push_back(i);
      }
      src = li.data(); item = sizeof(rtxh_light_info); n = li.size(); break;
    case RTXH_TABLE_P: src = s->P.data(); item = 12; n = s->P.size() / 3; break;
    case RTXH_TABLE_N: src = s->N.data(); item = 12; n = s->N.size() / 3; break;
    case RTXH_TABLE_UV: src = s->UV.data(); item = 8; n = s->UV.size() / 2; break;
    case RTXH_TABLE_S: src = s->S.data(); item = 12; n = s->S.size() / 3; break;
    case RTXH_TABLE_INDICES: src = s->idx.data(); item = 12; n = s->idx.size() / 3; break;
    case RTXH_TABLE_TRI_MATERIAL: src = s->tri_mat.data(); item = 4; n = s->tri_mat.size(); break;
    case RTXH_TABLE_TRI_LIGHT: src = s->tri_light.data(); item = 4; n = s->tri_light.size(); break;
    case RTXH_TABLE_TRI_FLAGS: src = s->tri_flags.data(); item = 1; n = s->tri_flags.size(); break;
    case RTXH_TABLE_ENV_FUNC: case RTXH_TABLE_ENV_CDF: case RTXH_TABLE_ENV_ROW_INT: case RTXH_TABLE_ENV_MARG_CDF: {
      const HostLight* e = nullptr;
      for (const HostLight& h : s->lights) if (h.l.kind == RT_LIGHT_INFINITE) { e = &h; break; }
      if (!e) { *n_items = 0; return RT_OK; }
      const std::vector<float>& v = table == RTXH_TABLE_ENV_FUNC ? e->dfunc : table == RTXH_TABLE_ENV_CDF ? e->dcdf : table == RTXH_TABLE_ENV_ROW_INT ? e->dint : e->mcdf;
      src = v.data(); item = 4; n = v.size(); break;
    }
    default: return fail(RT_ERR_INVALID, "unknown table");
  }
  *n_items = n;
  if (out) {
    if (capacity_bytes < n * item) return fail(RT_ERR_INVALID, "buffer too small");
    if (n) memcpy(out, src, n * item);
  }
  return RT_OK;
}
int rtxh_look_at(const float* pos, const float* look, const float* up, float* m16, float* minv16) {
  auto norm = [](float v[3]) { float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
  auto cross = [](const float a[3], const float b[3], float o[3]) { o[0] = (a[1] * b[2]) - (a[2] * b[1]); o[1] = (a[2] * b[0]) - (a[0] * b[2]); o[2] = (a[0] * b[1]) - (a[1] * b[0]); };
  Mat c2w = mat_identity();
  c2w.a[0][3] = pos[0]; c2w.a[1][3] = pos[1]; c2w.a[2][3] = pos[2]; c2w.a[3][3] = 1.0f;
  float dir[3] = {look[0] - pos[0], look[1] - pos[1], look[2] - pos[2]}; norm(dir);
  float upn[3] = {up[0], up[1], up[2]}; norm(upn);
  float left[3]; cross(upn, dir, left);
  if (std::sqrt(left[0] * left[0] + left[1] * left[1] + left[2] * left[2]) == 0.0f) { Mat id = mat_identity(); memcpy(m16, id.a, 64); memcpy(minv16, id.a, 64); return RT_OK; }
  norm(left);
  float new_up[3]; cross(dir, left, new_up);
  for (int r = 0; r < 3; ++r) { c2w.a[r][0] = left[r]; c2w.a[r][1] = new_up[r]; c2w.a[r][2] = dir[r]; }
  c2w.a[3][0] = c2w.a[3][1] = c2w.a[3][2] = 0.0f;
  Mat w2c = mat_inverse(c2w);
  memcpy(m16, w2c.a, 64); memcpy(minv16, c2w.a, 64);
  return RT_OK;
}

int rtxh_render(rtxh_scene* s, const rtxh_render_params* p, void* stream, float* film_xyzw, rt_stats* stats) {
  if (!s || !p || !film_xyzw) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  if (!s->dev) { int rc = rtxh_scene_upload(s, -1); if (rc != RT_OK) return rc; }
  CamFilm cf; int rc = setup_camera_film(p, cf); if (rc != RT_OK) return rc;
  rt_sampler_desc smp{p->spp, p->sampler_dims};
  rt_path_desc path{}; path.max_depth = p->max_depth; path.rr_threshold = p->rr_threshold; path.light_strategy = p->light_strategy;
  // PathIntegrator::create (path.rs:53-69): pixel_bounds = sample bounds, optionally intersected with "pixelbounds"
  int pb[4] = {cf.film.sample_bounds[0], cf.film.sample_bounds[1], cf.film.sample_bounds[2], cf.film.sample_bounds[3]};
  if (p->has_pixel_bounds) {  // Bounds2i::intersect (bounds.rs): max of the mins, min of the maxes; an empty result is kept and renders nothing
    pb[0] = std::max(pb[0], p->pixel_bounds[0]); pb[1] = std::max(pb[1], p->pixel_bounds[2]);
    pb[2] = std::min(pb[2], p->pixel_bounds[1]); pb[3] = std::min(pb[3], p->pixel_bounds[3]);
  }
  memcpy(path.pixel_bounds, pb, 16);
  rt_shard shard{p->rank, p->world_size > 0 ? p->world_size : 1};
  return rt_render(s->dev, &cf.cam, &cf.film, &smp, &path, &shard, p->flags, stream, film_xyzw, stats);
}
int rtxh_render_multi(rtxh_scene* s, const rtxh_render_params* p, const int32_t* devices, int32_t n_devices, int32_t chunks_per_device, float* film_xyzw,
                      rt_stats* total, rt_stats* per_device) {
  if (!s || !p || !film_xyzw || !devices || n_devices < 1) return fail(RT_ERR_INVALID, "null argument");
  if (!s->committed) return fail(RT_ERR_INVALID, "scene not committed");
  g_err.clear();
  const std::vector<int32_t> want(devices, devices + n_devices);
  if (!s->multi || s->multi_devices != want || s->multi_gen != s->commit_gen) {  // a re-committed scene gets new replicas
    if (s->multi) { rt_multi_destroy(s->multi); s->multi = nullptr; }
    rt_scene_desc d = make_desc(s);
    const int rc = rt_multi_create(&d, devices, n_devices, &s->multi);
    if (rc != RT_OK) return rc;
    s->multi_devices = want; s->multi_gen = s->commit_gen;
  }
  CamFilm cf; int rc = setup_camera_film(p, cf); if (rc != RT_OK) return rc;
  rt_sampler_desc smp{p->spp, p->sampler_dims};
  rt_path_desc path{}; path.max_depth = p->max_depth; path.rr_threshold = p->rr_threshold; path.light_strategy = p->light_strategy;
  int pb[4] = {cf.film.sample_bounds[0], cf.film.sample_bounds[1], cf.film.sample_bounds[2], cf.film.sample_bounds[3]};
  if (p->has_pixel_bounds) {
    pb[0] = std::max(pb[0], p->pixel_bounds[0]); pb[1] = std::max(pb[1], p->pixel_bounds[2]);
    pb[2] = std::min(pb[2], p->pixel_bounds[1]); pb[3] = std::min(pb[3], p->pixel_bounds[3]);
  }
  memcpy(path.pixel_bounds, pb, 16);
  return rt_multi_render(s->multi, &cf.cam, &cf.film, &smp, &path, chunks_per_device, p->flags, film_xyzw, total, per_device);
}
int rtxh_trace(rtxh_scene* s, const float* rays, uint64_t n, int32_t any_hit, float* out, uint64_t counters[2]) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  g_err.clear();
  if (!s->dev) { int rc = rtxh_scene_upload(s, -1); if (rc != RT_OK) return rc; }
  return any_hit ? rt_trace_any(s->dev, rays, n, (uint32_t*)out, counters) : rt_trace_closest(s->dev, rays, n, out, counters);
}
int rtxh_trace_device(rtxh_scene* s, const void* d_rays, uint64_t n, void* d_hits, int32_t reps, void* stream, float* ms) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  g_err.clear();
  if (!s->dev) { int rc = rtxh_scene_upload(s, -1); if (rc != RT_OK) return rc; }
  return rt_trace_closest_device(s->dev, d_rays, n, d_hits, reps, stream, ms);
}
int rtxh_light_distribution(rtxh_scene* s, int32_t n_voxels[3], float* func, float* cdf, float* func_int) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  g_err.clear();
  if (!s->dev) { int rc = rtxh_scene_upload(s, -1); if (rc != RT_OK) return rc; }
  return rt_light_distribution(s->dev, n_voxels, func, cdf, func_int);
}
int rtxh_scene_query(rtxh_scene* s, int32_t what) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  g_err.clear();
  if (!s->dev) { int rc = rtxh_scene_upload(s, -1); if (rc != RT_OK) return rc; }
  return rt_scene_query(s->dev, what);
}
int rtxh_scene_link_tables(rtxh_scene* s, int32_t mid, uint32_t* link_kept, uint32_t* link_full, uint64_t capacity_words, double* stats) {
  if (!s || !s->committed) return fail(RT_ERR_INVALID, "scene not committed");
  g_err.clear();
  rt_scene_desc d = make_desc(s);
  return rt_link_tables(&d, mid, link_kept, link_full, capacity_words, stats);
}
int rtxh_sizeof(const char* name) {
  if (!name) return -1;
#define RTXH_SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T);
  RTXH_SZ(rtxh_render_params) RTXH_SZ(rtxh_emitter_info) RTXH_SZ(rtxh_instance_info) RTXH_SZ(rtxh_light_info) RTXH_SZ(rtxh_ply) RTXH_SZ(rtxh_pbrt_result)
#undef RTXH_SZ
  return rt_sizeof(name);
}


// ---------------------------------------------------------------------------------------------- PLY / PFM readers
namespace {
struct PlyProp { std::string name; int type = -1; bool is_list = false; int count_type = -1; };  // type ids below
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };
// scalar type ids: 0 char 1 uchar 2 short 3 ushort 4 int 5 uint 6 float 7 double
int ply_type_id(const std::string& t) {
  static const char* names[][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"}, {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
  for (int i = 0; i < 8; ++i) if (t == names[i][0] || t == names[i][1]) return i;
  return -1;
}
const int kPlySize[8] = {1, 1, 2, 2, 4, 4, 4, 8};
struct PlyReader {
  FILE* f = nullptr; int format = 0;  // 0 ascii, 1 little endian, 2 big endian
  bool read_scalar(int type, double& out) {
    if (format == 0) { return fscanf(f, "%lf", &out) == 1; }
    unsigned char b[8]; const int n = kPlySize[type];
    if (fread(b, 1, (size_t)n, f) != (size_t)n) return false;
    if (format == 2) for (int i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
    switch (type) {
      case 0: { int8_t v; memcpy(&v, b, 1); out = v; break; }
      case 1: { uint8_t v; memcpy(&v, b, 1); out = v; break; }
      case 2: { int16_t v; memcpy(&v, b, 2); out = v; break; }
      case 3: { uint16_t v; memcpy(&v, b, 2); out = v; break; }
      case 4: { int32_t v; memcpy(&v, b, 4); out = v; break; }
      case 5: { uint32_t v; memcpy(&v, b, 4); out = v; break; }
      case 6: { float v; memcpy(&v, b, 4); out = v; break; }
      default: { double v; memcpy(&v, b, 8); out = v; break; }
    }
    return true;
  }
};
}  // namespace

static int ply_read_unguarded(const char* path, rtxh_ply* out);
int rtxh_ply_read(const char* path, rtxh_ply* out) {
  if (!path || !out) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  memset(out, 0, sizeof(*out));
  try { return ply_read_unguarded(path, out); }  // no C++ exception may cross the C ABI (a damaged header can ask for any amount of memory)
  catch (const std::bad_alloc&) { rtxh_ply_free(out); return fail(RT_ERR_OOM, std::string("out of memory while reading ") + path); }
  catch (const std::exception& e) { rtxh_ply_free(out); return fail(RT_ERR_INVALID, std::string(e.what()) + " (" + path + ")"); }
}
static int ply_read_unguarded(const char* path, rtxh_ply* out) {
  PlyReader rd; rd.f = fopen(path, "rb");
  if (!rd.f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
  struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{rd.f};
  char line[1024];
  if (!fgets(line, sizeof line, rd.f) || strncmp(line, "ply", 3) != 0) return fail(RT_ERR_INVALID, "not a PLY file");
  std::vector<PlyElem> elems; bool have_format = false, ended = false;
  while (fgets(line, sizeof line, rd.f)) {
    char a[256] = "", b[256] = "", c[256] = "", d[256] = "", e[256] = "";
    const int n = sscanf(line, "%255s %255s %255s %255s %255s", a, b, c, d, e);
    if (n <= 0) continue;
    const std::string kw = a;
    if (kw == "end_header") { ended = true; break; }
    if (kw == "comment" || kw == "obj_info") continue;
    if (kw == "format") {
      const std::string fm = b;
      if (fm == "ascii") rd.format = 0; else if (fm == "binary_little_endian") rd.format = 1; else if (fm == "binary_big_endian") rd.format = 2;
      else return fail(RT_ERR_INVALID, "unknown PLY format " + fm);
      have_format = true;
    } else if (kw == "element" && n >= 3) { PlyElem el; el.name = b; el.count = (size_t)strtoull(c, nullptr, 10); elems.push_back(el); }
    else if (kw == "property" && !elems.empty()) {
      PlyProp p;
      if (std::string(b) == "list" && n >= 5) { p.is_list = true; p.count_type = ply_type_id(c); p.type = ply_type_id(d); p.name = e; if (p.count_type < 0) return fail(RT_ERR_INVALID, "bad PLY list count type"); }
      else if (n >= 3) { p.type = ply_type_id(b); p.name = c; }
      if (p.type < 0) return fail(RT_ERR_INVALID, std::string("unknown PLY property type in: ") + line);
      elems.back().props.push_back(p);
    } else return fail(RT_ERR_INVALID, std::string("unexpected PLY header line: ") + line);
  }
  if (!ended || !have_format) return fail(RT_ERR_INVALID, "truncated PLY header");
  size_t vertex_count = 0, face_count = 0; bool has_normals = false, has_texture = false;
  for (const PlyElem& el : elems) {
    auto has = [&](const char* k) { for (const PlyProp& p : el.props) if (p.name == k) return true; return false; };
    if (el.name == "vertex") {
      vertex_count = el.count;
      if (!has("x") || !has("y") || !has("z")) return fail(RT_ERR_INVALID, "PLY: vertex coordinate property not found");  // plymesh.rs:42-51
      has_normals = has("nx") && has("ny") && has("nz");
      has_texture = (has("u") && has("v")) || (has("s") && has("t")) || (has("texture_u") && has("texture_v")) || (has("texture_s") && has("texture_t"));
    } else if (el.name == "face") face_count = el.count;
  }
  if (vertex_count == 0 || face_count == 0) return fail(RT_ERR_INVALID, "PLY file is invalid: no face/vertex elements found");  // :73-79
  if (vertex_count > 0x7fffffffull || face_count > 0x3fffffffull) return fail(RT_ERR_INVALID, "PLY too large");
  std::vector<float> P(vertex_count * 3, 0.0f), N(has_normals ? vertex_count * 3 : 0, 0.0f), UV(has_texture ? vertex_count * 2 : 0, 0.0f);
  std::vector<int32_t> idx; idx.reserve(face_count * 3);
  int32_t dropped = 0;
  for (const PlyElem& el : elems) {
    if (el.name != "vertex" && el.name != "face") return fail(RT_ERR_INVALID, "unexpected PLY element \"" + el.name + "\"");  // panics in the reference (:97)
    for (size_t i = 0; i < el.count; ++i) {
      std::vector<int32_t> face;
      for (const PlyProp& p : el.props) {
        if (!p.is_list) {
          double v;
          if (!rd.read_scalar(p.type, v)) return fail(RT_ERR_INVALID, "truncated PLY payload");
          if (el.name == "vertex" && p.type == 6) {  // only Property::Float is taken (plymesh.rs:196-214)
            const float fv = (float)v; const std::string& k = p.name;
            if (k == "x") P[3 * i] = fv; else if (k == "y") P[3 * i + 1] = fv; else if (k == "z") P[3 * i + 2] = fv;
            else if (has_normals && k == "nx") N[3 * i] = fv; else if (has_normals && k == "ny") N[3 * i + 1] = fv; else if (has_normals && k == "nz") N[3 * i + 2] = fv;
            else if (has_texture && (k == "u" || k == "texture_u" || k == "s" || k == "texture_s")) UV[2 * i] = fv;
            else if (has_texture && (k == "v" || k == "t" || k == "texture_v" || k == "texture_t")) UV[2 * i + 1] = fv;
          }
        } else {
          double cnt;
          if (!rd.read_scalar(p.count_type, cnt) || cnt < 0 || cnt > 1e6) return fail(RT_ERR_INVALID, "bad PLY list");
          const bool take = el.name == "face" && p.name == "vertex_indices" && (p.type == 4 || p.type == 5);  // ListInt / ListUInt only (:229-234)
          for (int k = 0; k < (int)cnt; ++k) {
            double v;
            if (!rd.read_scalar(p.type, v)) return fail(RT_ERR_INVALID, "truncated PLY payload");
            if (take) face.push_back((int32_t)(int64_t)v);
          }
        }
      }
      if (el.name == "face") {
        const size_t L = face.size();
        if (L != 3 && L != 4) { ++dropped; continue; }  // :104-107
        for (size_t k = 0; k < L; ++k) if (face[k] < 0 || (size_t)face[k] >= vertex_count) return fail(RT_ERR_INVALID, "PLY face index out of range");
        idx.push_back(face[0]); idx.push_back(face[1]); idx.push_back(face[2]);
        if (L == 4) { idx.push_back(face[3]); idx.push_back(face[0]); idx.push_back(face[2]); }  // :113-118
      }
    }
  }
  auto dup = [](const void* src, size_t bytes) -> void* { void* p = malloc(bytes ? bytes : 1); if (p && bytes) memcpy(p, src, bytes); return p; };
  out->n_verts = (int32_t)vertex_count; out->n_tris = (int32_t)(idx.size() / 3); out->n_dropped_faces = dropped;
  out->P = (float*)dup(P.data(), P.size() * 4);
  out->N = has_normals ? (float*)dup(N.data(), N.size() * 4) : nullptr;
  out->UV = has_texture ? (float*)dup(UV.data(), UV.size() * 4) : nullptr;
  out->idx = (int32_t*)dup(idx.data(), idx.size() * 4);
  return RT_OK;
}
void rtxh_ply_free(rtxh_ply* p) { if (!p) return; free(p->P); free(p->N); free(p->UV); free(p->idx); memset(p, 0, sizeof(*p)); }
void rtxh_free(void* p) { free(p); }

// ---------------------------------------------------------------- sampled spectra -> RGB (rc/spectrum.rs, rc/cie.rs)
// Spectrum is RGB in the reference (spectrum.rs:12-17); spectral data enters it at scene-description time only: Metal::create's copper
// default (metal.rs:25-29), `spectrum` parameters naming SPD files and `blackbody` parameters (paramset.rs:141-152, 254-310).
#include "rtx_spectrum_tables.inl"
static float spd_interpolate(const float* lambda, const float* vals, size_t n, float l) {  // interpolate_spectrum_samples, spectrum.rs:196-211
  if (l <= lambda[0]) return vals[0];
  if (l >= lambda[n - 1]) return vals[n - 1];
  size_t first = 0, len = n;  // find_interval(n, |i| lambda[i] <= l), lib.rs:171-189
  while (len > 0) {
    const size_t half = len >> 1, middle = first + half;
    if (lambda[middle] <= l) { first = middle + 1; len -= half + 1; } else len = half;
  }
  long off = (long)first - 1; if (off < 0) off = 0; if (off > (long)n - 2) off = (long)n - 2;
  const float t = (l - lambda[off]) / (lambda[off + 1] - lambda[off]);
  return vals[off] * (1.0f - t) + vals[off + 1] * t;  // lerp, lib.rs:107-117
}
int rtxh_spectrum_from_sampled(const float* lambda, const float* v, int32_t n, float rgb[3]) {  // Spectrum::from_sampled, spectrum.rs:108-126
  if (!lambda || !v || !rgb || n < 1) return fail(RT_ERR_INVALID, "bad spectrum samples");
  for (int i = 0; i + 1 < n; ++i) if (!(lambda[i + 1] > lambda[i])) return fail(RT_ERR_INVALID, "spectrum wavelengths must increase (the reference asserts it, spectrum.rs:197-199)");
  float xyz[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < kNCieSamples; ++i) {
    const float val = n == 1 ? v[0] : spd_interpolate(lambda, v, (size_t)n, 360.0f + (float)i);
    xyz[0] += val * kCieX[i]; xyz[1] += val * kCieY[i]; xyz[2] += val * kCieZ[i];
  }
  const float scale = ((360.0f + (float)(kNCieSamples - 1)) - 360.0f) / (kCieYIntegral * (float)kNCieSamples);
  xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
  rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];  // from_xyz, spectrum.rs:91-96
  rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
  rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
  return RT_OK;
}
static float blackbody_le(float lambda_nm, float temp) {  // blackbody, spectrum.rs:168-185
  const float c = 299792458.0f, h = 6.62606957e-34f, kb = 1.3806488e-23f;
  const float l = lambda_nm * 1e-9f;
  const float lambda5 = (l * l) * (l * l) * l;
  return (2.0f * h * c * c) / (lambda5 * (std::exp((h * c) / (l * kb * temp)) - 1.0f));
}
int rtxh_spectrum_blackbody(float temperature, float scale, float rgb[3]) {  // add_blackbody_spectrum, paramset.rs:291-310 + blackbody_normalized, spectrum.rs:187-194
  if (!rgb || !(temperature > 0.0f)) return fail(RT_ERR_INVALID, "blackbody temperature must be positive (the reference asserts it, spectrum.rs:169)");
  std::vector<float> lam((size_t)kNCieSamples), le((size_t)kNCieSamples);
  const float lambda_max = 2.8977721e-3f / temperature * 1e9f;
  const float max_l = blackbody_le(lambda_max, temperature);
  for (int i = 0; i < kNCieSamples; ++i) { lam[(size_t)i] = 360.0f + (float)i; le[(size_t)i] = blackbody_le(lam[(size_t)i], temperature) / max_l; }
  const int rc = rtxh_spectrum_from_sampled(lam.data(), le.data(), kNCieSamples, rgb);
  if (rc != RT_OK) return rc;
  for (int k = 0; k < 3; ++k) rgb[k] = scale * rgb[k];
  return RT_OK;
}
void rtxh_copper(float eta_rgb[3], float k_rgb[3]) {  // the defaults of Metal::create, metal.rs:25-29
  (void)rtxh_spectrum_from_sampled(kCopperWavelengths, kCopperN, kCopperSamples, eta_rgb);
  (void)rtxh_spectrum_from_sampled(kCopperWavelengths, kCopperK, kCopperSamples, k_rgb);
}

int rtxh_pfm_read(const char* path, int32_t* width, int32_t* height, float** rgb) {
  if (!path || !width || !height || !rgb) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  FILE* f = fopen(path, "rb");
  if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  auto read_word = [&]() {  // imageio.rs:166-177: bytes up to the next ' ', '\n' or '\t' (which is consumed)
    std::string w; int c;
    while ((c = fgetc(f)) != EOF) { if (c == ' ' || c == '\n' || c == '\t') break; w.push_back((char)c); }
    return w;
  };
  const std::string magic = read_word();
  const int nc = magic == "Pf" ? 1 : (magic == "PF" ? 3 : 0);
  if (!nc) return fail(RT_ERR_INVALID, std::string("error reading PFM file ") + path);
  char* end = nullptr;
  std::string w = read_word(); const unsigned long long W = strtoull(w.c_str(), &end, 10); if (w.empty() || *end) return fail(RT_ERR_INVALID, "PFM: failed to parse width");
  w = read_word(); const unsigned long long H = strtoull(w.c_str(), &end, 10); if (w.empty() || *end) return fail(RT_ERR_INVALID, "PFM: failed to parse height");
  w = read_word(); const float scale = strtof(w.c_str(), &end); if (w.empty() || *end) return fail(RT_ERR_INVALID, "PFM: failed to parse scale");
  if (W == 0 || H == 0 || W > 65536 || H > 65536) return fail(RT_ERR_INVALID, "PFM: bad dimensions");
  const bool file_little = scale < 0.0f;
  std::vector<float> row((size_t)W * nc);
  float* out = (float*)malloc((size_t)W * H * 3 * sizeof(float));
  if (!out) return fail(RT_ERR_INVALID, "out of memory");
  for (long long y = (long long)H - 1; y >= 0; --y) {  // flip in Y: P*M has its origin at the lower left (:217)
    if (fread(row.data(), 4, row.size(), f) != row.size()) { free(out); return fail(RT_ERR_INVALID, "PFM: truncated data"); }
    for (size_t i = 0; i < row.size(); ++i) {
      unsigned char b[4]; memcpy(b, &row[i], 4);
      if (!file_little) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
      float v; memcpy(&v, b, 4);
      if (fabsf(scale) != 1.0f) v *= fabsf(scale);
      const size_t x = i / nc, ch = i % nc;
      if (nc == 1) { float* o = out + ((size_t)y * W + x) * 3; o[0] = o[1] = o[2] = v; }
      else out[((size_t)y * W + x) * 3 + ch] = v;
    }
  }
  *width = (int32_t)W; *height = (int32_t)H; *rgb = out;
  return RT_OK;
}
}  // extern "C"

// PNG / TGA / Radiance HDR decoding
#include "rtx_images.inl"

// pbrt-v3 scene files -> rtxh_scene + rtxh_render_params
#include "rtx_pbrt.inl"
