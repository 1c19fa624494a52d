// rtx_hip.hip — C-ABI implementation (include/rtx_hip.h): scene upload, the per-frame launch
// sequence of the wavefront kernels, and the kernel-level entry points used by the parity tests.
// There is no CPU fallback anywhere in this file: without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <limits>
#include <algorithm>
#include <functional>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>

#include "../../include/rtx_hip.h"
#include "rtx_kernels.h"
#include "rtx_shade_launch.h"
#include "rtx_ref_launch.h"
#include "rtx_link_tables.h"

using namespace rtx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
// The measurement knobs that remain are the A/B controls of features that are in the product (and two test hooks): read through these two.
static bool env_is(const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; }
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// Guide tables (environment-map rows, light-distribution rows) bracket a CDF search: bucket k of 2^glog holds the entries whose cdf lies in [k, k+1) / 2^glog.
// Density in quarters of an entry per bucket on average: 4 = as many buckets as entries (the search that follows is 0-2 dependent loads instead of the 4-5 of
// round 2's 16 entries per bucket; a 2048 x 1024 map's tables grow from 0.3 to 4 MB).
static long guide_quarters() { return 4; }
#define HIP_TRY(expr)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t ensure(size_t n) {
    if (n <= bytes && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, n > 0 ? n : 16);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T* as() const { return (T*)p; }
};

struct SamplerPlan {  // launch plan of K0 for one (spp, dims): stream segments, reciprocal table, retry list
  unsigned spp = 0, dims = 0, seg_len = 0, n_segs = 0;
  DevBuf segs, magic, dirty, partners;  // partners: the shuffle's swap partners of one batch, [table][pixel][spp] u16
};

struct rt_scene {
  int device = 0;
  SamplerPlan sampler_plan;
  DScene d{};
  bool small = false;
  bool lambert_materials = false;  // the material half of lambert_only: with other light kinds k_shade<3>
  bool lds_records_q = false; // a scene of quadric emitters whose triangles, lights, materials and textures all fit (QLIGHTS forms with LDSREC = 1)
  bool lds_tables = false;    // ... its lights, materials, textures and image headers do (the plain forms)
  bool lds_mats = false;      // ... at least its material and texture tables do (the LEAN forms of the other front-ends)
  bool lds_records = false;   // the scene's shade / traversal records and its light table fit the shade kernel's LDS (k_shade<1, .., LDSREC>)
  bool lambert_only = false;  // every material is matte{constant Kd, sigma == 0} and every light an area light: k_shade<1>
  bool lean_qlights = false;  // LEAN with sphere lights: the QLIGHTS forms of k_shade<3 | 5 | 6>, quadric hits routed to the generic kernel
  bool lean_shade = false;    // every light an area light on a triangle and every texture a constant: the LEAN forms of k_shade<3 | 5 | 6> (no out-of-line evaluator, three waves)
  int stack_depth = 64;  // entries the to-visit stack needs for this tree (<= 64, rc/bvh/mod.rs:374)
  unsigned n_nodes = 0, n_tris = 0; int n_lights = 0;
  DevBuf link8, link8_full; unsigned lds_nodes_tested = 0;  // link tables of the stackless LDS walks; nodes they test (of n_nodes)
  DevBuf pairs, tmin_stack;  // child-pair node records and the HBM half of the traversal stack (k_trace_pair)
  DevBuf top_pairs, deep_stack; bool use_top = false;  // k_trace_top: LDS-resident top of the tree, HBM spill of stack entries beyond the LDS ones
  bool top_for_closest = false;  // closest-hit rays through k_trace_top as well (shadow rays always, when no four-wide records exist)
  bool use_pairs = false;
  bool deep_column = false;  // top level + deepest object need more than 64 stack entries in one column: k_trace_big with 128
  DevBuf quads; bool use_quads = false; int quad_stack_depth = 0;  // four-wide records of the any-hit kernel (k_trace_quad)
  DevBuf tri_rec;  // per-triangle shade records (k_tri_records)
  DevBuf nodes, tri_p, tri_n, tri_uv, tri_s, tri_alpha, spheres, textures, images, materials, lights, texels, dist, guides, prim_class;
  bool has_spheres = false;
  bool has_instances = false;  // object instances: two-level traversal in k_trace_big<.., GENERAL>, every vertex shaded by k_shade<0, true>
  DevBuf instances;
  bool mid = false;  // plain scene of <= RT_MID_NODES / RT_MID_TRIS: occlusion rays through k_trace<.., MID>
  bool general_prims = false;  // alpha-masked triangles, quadrics, object instances: the GENERAL instantiations of the trace kernels
  bool obj_general = false;    // some instanced object holds a quadric or a masked triangle: the objects are walked by the general one-node-per-step walk (instance_intersect)
  bool has_masks = false;      // some triangle carries an alpha / shadow-alpha mask (RT_GEN_ALL; without: RT_GEN_NO_MASKS, 134 instead of 179 VGPRs)
  bool masked_emitters = false;  // ... and some of them emit: every vertex is shaded by k_shade<0, true> (Shape::pdf_wi evaluates the mask)
  std::vector<DLight> h_lights;
  // light distribution tables (built per render, rc/integrator/path.rs:86-94)
  DevBuf ld_func, ld_cdf, ld_int, ld_mark, ld_list, ld_slot, ld_guide, ld_rows8, ld_dense8;
  DevBuf self;  // `d` in device memory (DScene::self), rewritten whenever `d` changes
  int ld_strategy_built = -1; bool ld_all_voxels = false;  // the tables are a function of the scene alone: built once per strategy, kept across frames
  std::mutex render_mutex;  // rt_render shares the workspace below: concurrent calls on one rt_scene take turns
  // per-render workspace
  DevBuf ws[32];
  DevBuf film_acc, own_acc, film_out, counters, stats, filter_table, ref_samples, ref_stack;
  DevBuf bin_words, bin_sorted, bin_at;  // material binning of the shade queue (generic shade path); bin_at: u16 per queue entry
  unsigned n_materials = 0, n_code_classes = 0, n_lambert_classes = 0, n_small_classes = 0, n_wide_classes = 0;
  // sampler tables are double-buffered: K0 of batch b+1 runs on aux_stream under the path kernels of batch b
  DevBuf scrambles[2], perms[2];
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_tables[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}, ev_batch_done[2] = {nullptr, nullptr}, ev_frame_begin = nullptr;  // ev_tables[buffer][table group]
  int n_cu = 256;
  std::vector<hipEvent_t> event_pool;
  ~rt_scene() {
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { for (int g = 0; g < 3; ++g) if (ev_tables[i][g]) (void)hipEventDestroy(ev_tables[i][g]); if (ev_batch_done[i]) (void)hipEventDestroy(ev_batch_done[i]); }
    if (ev_frame_begin) (void)hipEventDestroy(ev_frame_begin);
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
  }
};

extern "C" const char* rt_last_error(void) { return g_err.c_str(); }
extern "C" const char* rt_version(void) { return "rtx-mi355x 0.5 (gfx950 wavefront path tracer)"; }
extern "C" int rt_sizeof(const char* name) {
  if (!name) return -1;
#define RT_SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T);
  RT_SZ(rt_bvh_node) RT_SZ(rt_tri_meta) RT_SZ(rt_sphere) RT_SZ(rt_instance) RT_SZ(rt_texture) RT_SZ(rt_image) RT_SZ(rt_material) RT_SZ(rt_light) RT_SZ(rt_scene_desc)
  RT_SZ(rt_camera) RT_SZ(rt_film_desc) RT_SZ(rt_sampler_desc) RT_SZ(rt_path_desc) RT_SZ(rt_shard) RT_SZ(rt_stats)
#undef RT_SZ
  return -1;
}
extern "C" int rt_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  return 1;
}

static int upload(DevBuf& b, const void* src, size_t bytes) {
  HIP_TRY(b.ensure(bytes));
  if (bytes) HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return RT_OK;
}

static void fill_ewa_lut() {
  float lut[128];
  for (int i = 0; i < 128; ++i) {  // rc/mipmap.rs:33-44
    float alpha = 2.0f;
    float r2 = (float)i / (128.0f - 1.0f);
    lut[i] = expf(-alpha * r2) - expf(-alpha);
  }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kEwaLut), lut, sizeof(lut));
  rtx_shade_set_ewa_lut(lut);  // the shade kernels' translation unit has a copy of its own
  rtx_ref_set_ewa_lut(lut);    // ... and the reference-stream kernel's
}

static size_t tmin_stack_bytes(const rt_scene* s);
static size_t deep_stack_bytes(const rt_scene* s);
// Is every quadric that carries an area light a Sphere (kind 0) that no triangle of the scene reaches into? Then a path vertex on a triangle lies outside
// every emitter sphere, and Sphere::sample_si / Sphere::pdf_wi (rc/shapes/sphere.rs:246-334) take their cone branches for it: `distance_squared(p_origin,
// p_center) <= radius^2` - the reference's own inside test, world-space distance against the object-space radius - is false with a margin of 1e-3 radius
// (p_origin is the vertex moved by its error bounds, orders of magnitude less). Exact point-triangle distances in double; gives up (false) beyond 2e8 pairs.
static bool sphere_lights_clear(const rt_scene_desc* desc) {
  std::vector<uint32_t> emitters;
  for (uint32_t i = 0; i < desc->n_lights; ++i) {
    const rt_light& l = desc->lights[i];
    if (l.kind != RT_LIGHT_DIFFUSE_AREA || l.prim < 0 || (uint32_t)l.prim >= desc->n_tris) continue;
    if (!(desc->tri_meta[l.prim].flags & RT_PRIM_SPHERE)) continue;
    uint32_t k; memcpy(&k, &desc->tri_p[9 * (size_t)l.prim + 6], 4);  // a quadric's leaf record: its index in p2.x
    if (k >= desc->n_spheres || desc->spheres[k].kind != 0) return false;
    emitters.push_back(k);
  }
  if ((double)emitters.size() * (double)desc->n_tris > 2e8) return false;
  for (uint32_t k : emitters) {
    const rt_sphere& sp = desc->spheres[k];
    const double c[3] = {sp.o2w[3], sp.o2w[7], sp.o2w[11]};
    const double r = (double)sp.radius * 1.001 + 1e-6 * std::max(std::max(fabs(c[0]), fabs(c[1])), fabs(c[2])), r2 = r * r;
    for (uint32_t t = 0; t < desc->n_tris; ++t) {
      if (desc->tri_meta[t].flags & (RT_PRIM_SPHERE | RT_PRIM_INSTANCE)) continue;
      const float* q = desc->tri_p + 9 * (size_t)t;
      // closest point of triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5), relative to p
      double a[3], b[3], cc[3];
      for (int j = 0; j < 3; ++j) { a[j] = q[j] - c[j]; b[j] = q[3 + j] - c[j]; cc[j] = q[6 + j] - c[j]; }
      auto dot3 = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
      double ab[3], ac[3]; for (int j = 0; j < 3; ++j) { ab[j] = b[j] - a[j]; ac[j] = cc[j] - a[j]; }
      double best[3];
      const double d1 = -dot3(ab, a), d2 = -dot3(ac, a);
      const double d3 = -dot3(ab, b), d4 = -dot3(ac, b), d5 = -dot3(ab, cc), d6 = -dot3(ac, cc);
      const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
      if (d1 <= 0 && d2 <= 0) { for (int j = 0; j < 3; ++j) best[j] = a[j]; }
      else if (d3 >= 0 && d4 <= d3) { for (int j = 0; j < 3; ++j) best[j] = b[j]; }
      else if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int j = 0; j < 3; ++j) best[j] = a[j] + v * ab[j]; }
      else if (d6 >= 0 && d5 <= d6) { for (int j = 0; j < 3; ++j) best[j] = cc[j]; }
      else if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); for (int j = 0; j < 3; ++j) best[j] = a[j] + w * ac[j]; }
      else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int j = 0; j < 3; ++j) best[j] = b[j] + w * (cc[j] - b[j]); }
      else { const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den; for (int j = 0; j < 3; ++j) best[j] = a[j] + ab[j] * v + ac[j] * w; }
      if (!(dot3(best, best) > r2)) return false;  // (a NaN vertex fails too)
    }
  }
  return true;
}

extern "C" int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
  if (!desc || !out) return fail(RT_ERR_INVALID, "null argument");
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  if (desc->n_nodes == 0 || desc->n_tris == 0 || !desc->nodes || !desc->tri_p || !desc->tri_meta) return fail(RT_ERR_INVALID, "empty scene");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  int dev = 0; HIP_TRY(hipGetDevice(&dev));
  rt_scene* s = new rt_scene();
  s->device = dev;
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, dev) == hipSuccess) s->n_cu = prop.multiProcessorCount;
  s->n_nodes = desc->n_nodes; s->n_tris = desc->n_tris; s->n_lights = (int)desc->n_lights;
  int rc;
#define TRY_RC(x) do { rc = (x); if (rc != RT_OK) { delete s; return rc; } } while (0)
  TRY_RC(upload(s->nodes, desc->nodes, sizeof(rt_bvh_node) * (size_t)desc->n_nodes));
  // 48-byte triangle records: coordinates + meta in the w lanes
  std::vector<float> tp((size_t)desc->n_tris * 12);
  for (size_t i = 0; i < desc->n_tris; ++i) {
    const float* p = desc->tri_p + 9 * i; const rt_tri_meta& m = desc->tri_meta[i];
    float* q = &tp[12 * i];
    q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; memcpy(&q[3], &m.material, 4);
    q[4] = p[3]; q[5] = p[4]; q[6] = p[5]; memcpy(&q[7], &m.light, 4);
    q[8] = p[6]; q[9] = p[7]; q[10] = p[8]; memcpy(&q[11], &m.flags, 4);
    if ((m.flags & RT_TRI_HAS_N) && !desc->tri_n) { delete s; return fail(RT_ERR_INVALID, "tri flags need tri_n"); }
    if ((m.flags & RT_TRI_HAS_UV) && !desc->tri_uv) { delete s; return fail(RT_ERR_INVALID, "tri flags need tri_uv"); }
    if ((m.flags & RT_TRI_HAS_S) && !desc->tri_s) { delete s; return fail(RT_ERR_INVALID, "tri flags need tri_s"); }
    if (m.flags & RT_PRIM_INSTANCE) {
      uint32_t k; memcpy(&k, p + 6, 4);
      if (!desc->instances || k >= desc->n_instances || i >= desc->n_top_prims) { delete s; return fail(RT_ERR_INVALID, "instance index out of range"); }
      if (m.flags != RT_PRIM_INSTANCE || m.light >= 0) { delete s; return fail(RT_ERR_INVALID, "an instance primitive carries triangle attributes or a light"); }
      continue;
    }
    if (m.material < 0 || (uint32_t)m.material >= desc->n_materials) { delete s; return fail(RT_ERR_INVALID, "material index out of range"); }
    if (m.light >= (int)(desc->n_lights + desc->n_unlisted_lights)) { delete s; return fail(RT_ERR_INVALID, "light index out of range"); }
    if (m.flags & (RT_TRI_HAS_ALPHA | RT_TRI_HAS_SHADOW_ALPHA)) {
      if (!desc->tri_alpha) { delete s; return fail(RT_ERR_INVALID, "tri flags need tri_alpha"); }
      for (int k = 0; k < 2; ++k)
        if ((m.flags & (k == 0 ? RT_TRI_HAS_ALPHA : RT_TRI_HAS_SHADOW_ALPHA)) && (desc->tri_alpha[2 * i + k] < 0 || (uint32_t)desc->tri_alpha[2 * i + k] >= desc->n_textures)) {
          delete s; return fail(RT_ERR_INVALID, "alpha texture out of range");
        }
      s->general_prims = true; s->has_masks = true;
    }
  }
  if (s->general_prims) TRY_RC(upload(s->tri_alpha, desc->tri_alpha, (size_t)desc->n_tris * 8));
  static_assert(sizeof(rt_sphere) == sizeof(DSphere), "rt_sphere and DSphere are the same record");
  for (size_t i = 0; i < desc->n_tris; ++i)
    if (desc->tri_meta[i].flags & RT_PRIM_SPHERE) {
      uint32_t k; memcpy(&k, desc->tri_p + 9 * i + 6, 4);
      if (!desc->spheres || k >= desc->n_spheres) { delete s; return fail(RT_ERR_INVALID, "sphere index out of range"); }
      if (desc->tri_meta[i].flags & (RT_TRI_HAS_N | RT_TRI_HAS_UV | RT_TRI_HAS_S | RT_TRI_HAS_ALPHA | RT_TRI_HAS_SHADOW_ALPHA)) { delete s; return fail(RT_ERR_INVALID, "a sphere primitive carries triangle attributes"); }
      s->has_spheres = true;
    }
  if (s->has_spheres) { TRY_RC(upload(s->spheres, desc->spheres, (size_t)desc->n_spheres * sizeof(rt_sphere))); s->general_prims = true; }
  const uint32_t n_top_nodes = desc->n_instances ? desc->n_top_nodes : desc->n_nodes, n_top_prims = desc->n_instances ? desc->n_top_prims : desc->n_tris;
  if (desc->n_instances) {  // rt_instance -> DInstance: + the first hit id of each instance
    if (!desc->instances || n_top_nodes == 0 || n_top_nodes > desc->n_nodes || n_top_prims == 0 || n_top_prims > desc->n_tris) { delete s; return fail(RT_ERR_INVALID, "bad instance tables"); }
    std::vector<DInstance> di(desc->n_instances);
    uint64_t id = n_top_prims;
    for (uint32_t k = 0; k < desc->n_instances; ++k) {
      const rt_instance& in = desc->instances[k];
      if (in.n_prims == 0 || (uint64_t)in.prim_base + in.n_prims > desc->n_tris || in.prim_base < n_top_prims || (in.n_nodes == 0 && in.n_prims != 1) ||
          (in.n_nodes != 0 && ((uint64_t)in.node_base + in.n_nodes > desc->n_nodes || in.node_base < n_top_nodes))) { delete s; return fail(RT_ERR_INVALID, "instance ranges out of bounds"); }
      // An object holds triangles (masked or not) and quadrics (round 6: TransformedPrimitive wraps whatever the object definition collected, primitive.rs:79-118) - not
      // another instance (the reference's ObjectInstance inside an object definition is an error, api.rs:1056-1059), and no light of the scene's list (an emitter inside
      // an object is never a listed light, api.rs:954-964)
      for (uint32_t t = in.prim_base; t < in.prim_base + in.n_prims; ++t) {
        if ((desc->tri_meta[t].flags & RT_PRIM_INSTANCE) || (desc->tri_meta[t].light >= 0 && (uint32_t)desc->tri_meta[t].light < desc->n_lights)) {
          delete s; return fail(RT_ERR_UNSUPPORTED, "an instanced object holds triangles and quadrics only, and no light of the scene's list");
        }
        if (desc->tri_meta[t].flags & (RT_PRIM_SPHERE | RT_TRI_HAS_ALPHA | RT_TRI_HAS_SHADOW_ALPHA)) s->obj_general = true;
      }
      memcpy(di[k].o2w, in.o2w, 64); memcpy(di[k].w2o, in.w2o, 64);
      di[k].node_base = in.node_base; di[k].n_nodes = in.n_nodes; di[k].prim_base = in.prim_base; di[k].n_prims = in.n_prims; di[k].id_base = (unsigned)id;
      id += in.n_prims;
      if (id >= (1ull << 31)) { delete s; return fail(RT_ERR_UNSUPPORTED, "more than 2^31 instanced primitives"); }
    }
    TRY_RC(upload(s->instances, di.data(), di.size() * sizeof(DInstance)));
    s->has_instances = true; s->general_prims = true;
  }
  TRY_RC(upload(s->tri_p, tp.data(), tp.size() * 4));
  if (desc->tri_n) TRY_RC(upload(s->tri_n, desc->tri_n, (size_t)desc->n_tris * 36));
  if (desc->tri_uv) TRY_RC(upload(s->tri_uv, desc->tri_uv, (size_t)desc->n_tris * 24));
  if (desc->tri_s) TRY_RC(upload(s->tri_s, desc->tri_s, (size_t)desc->n_tris * 36));
  // images: one blob of float4 texels, every level cut into tiles of 4 x 2 texels = one 128-byte line (mip_texel in rtx_dev_shading.h): a bilinear
  // or EWA footprint then touches fewer lines than with 12-byte row-major texels, and a texel is one aligned 16-byte load. The layout changes no value.
  std::vector<DImage> himg(desc->n_images);
  {
    auto is_pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    size_t total = 0;  // in float4 texels, levels padded to whole tiles
    for (uint32_t i = 0; i < desc->n_images; ++i) {
      const rt_image& im = desc->images[i];
      if (im.n_levels <= 0 || im.n_levels > RT_MAX_MIP_LEVELS) { delete s; return fail(RT_ERR_INVALID, "bad mip level count"); }
      for (int l = 0; l < im.n_levels; ++l) {
        // MIPMap::new resamples to powers of two and halves from there (rc/mipmap.rs:75-139), which is what lets Repeat wrap by a mask
        if (!is_pow2(im.width[l]) || !is_pow2(im.height[l])) { delete s; return fail(RT_ERR_INVALID, "MIP level sizes must be powers of two (rc/mipmap.rs:75-139)"); }
        if ((uint64_t)im.offset[l] + (uint64_t)im.width[l] * im.height[l] > im.n_texels) { delete s; return fail(RT_ERR_INVALID, "MIP level outside the texel array"); }
        total += (size_t)std::max(im.width[l], 4) * std::max(im.height[l], 2);
      }
    }
    std::vector<float> blob((total + 1) * 4, 0.0f);
    size_t base = 0;
    for (uint32_t i = 0; i < desc->n_images; ++i) {
      const rt_image& im = desc->images[i];
      DImage& d = himg[i];
      d.n_levels = im.n_levels; d.trilinear = im.trilinear; d.max_aniso = im.max_anisotropy; d.wrap = im.wrap;
      for (int l = 0; l < 16; ++l) { d.w[l] = 0; d.h[l] = 0; d.off[l] = 0; d.tshift[l] = 0; }
      for (int l = 0; l < im.n_levels; ++l) {
        const int w = im.width[l], h = im.height[l], pw = std::max(w, 4), ph = std::max(h, 2);
        int ts = 0; while ((4 << ts) < pw) ++ts;  // tiles per row = 2^ts
        d.w[l] = w; d.h[l] = h; d.off[l] = base; d.tshift[l] = ts;
        const float* src = im.texels + 3 * (size_t)im.offset[l];
        for (int t = 0; t < h; ++t)
          for (int x = 0; x < w; ++x) {
            const size_t idx = base + ((((size_t)(t >> 1) << ts) + (size_t)(x >> 2)) << 3) + (size_t)((t & 1) << 2) + (size_t)(x & 3);
            const float* q = src + 3 * ((size_t)t * w + x);
            float* o = &blob[idx * 4]; o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
          }
        base += (size_t)pw * ph;
      }
    }
    TRY_RC(upload(s->texels, blob.data(), blob.size() * 4));
    for (auto& d : himg) d.texels = s->texels.as<float4>();
    TRY_RC(upload(s->images, himg.data(), himg.size() * sizeof(DImage)));
  }
  std::vector<DTexture> htex(desc->n_textures);
  for (uint32_t i = 0; i < desc->n_textures; ++i) {
    const rt_texture& t = desc->textures[i]; DTexture& d = htex[i];
    d.kind = t.kind; d.v[0] = t.value[0]; d.v[1] = t.value[1]; d.v[2] = t.value[2];
    d.tex1 = t.tex1; d.tex2 = t.tex2; d.amount = t.amount; d.image = t.image;
    d.su = t.mapping[0]; d.sv = t.mapping[1]; d.du = t.mapping[2]; d.dv = t.mapping[3];
    auto ok = [&](int id) { return id >= 0 && (uint32_t)id < desc->n_textures; };
    const bool comb = t.kind == RT_TEX_SCALE || t.kind == RT_TEX_MIX || t.kind == RT_TEX_CHECKER;
    if (t.kind < RT_TEX_CONST || t.kind > RT_TEX_FBM) { delete s; return fail(RT_ERR_INVALID, "unknown texture kind"); }
    if (comb && (!ok(t.tex1) || !ok(t.tex2))) { delete s; return fail(RT_ERR_INVALID, "texture operand out of range"); }
    if (t.kind == RT_TEX_MIX && !ok(t.amount)) { delete s; return fail(RT_ERR_INVALID, "mix amount out of range"); }
    if (t.kind == RT_TEX_IMAGE && (t.image < 0 || (uint32_t)t.image >= desc->n_images)) { delete s; return fail(RT_ERR_INVALID, "image index out of range"); }
  }
  // combinators (scale / mix / checkerboard) nested deeper than two levels are not expanded on the device
  auto is_comb = [&](int id) { int k = desc->textures[id].kind; return k == RT_TEX_SCALE || k == RT_TEX_MIX || k == RT_TEX_CHECKER; };
  for (uint32_t i = 0; i < desc->n_textures; ++i) {
    const rt_texture& t = desc->textures[i];
    if (!is_comb((int)i)) continue;
    if (t.kind == RT_TEX_MIX && is_comb(t.amount)) { delete s; return fail(RT_ERR_INVALID, "mix amount must be a leaf texture"); }
    const int ops[2] = {t.tex1, t.tex2};
    for (int k = 0; k < 2; ++k) {
      const rt_texture& c = desc->textures[ops[k]];
      if (!is_comb(ops[k])) continue;
      if (c.kind == RT_TEX_MIX && is_comb(c.amount)) { delete s; return fail(RT_ERR_INVALID, "mix amount must be a leaf texture"); }
      if (is_comb(c.tex1) || is_comb(c.tex2)) { delete s; return fail(RT_ERR_INVALID, "texture nesting deeper than 2"); }
    }
  }
  TRY_RC(upload(s->textures, htex.data(), htex.size() * sizeof(DTexture)));
  std::vector<DMaterial> hmat(desc->n_materials);
  for (uint32_t i = 0; i < desc->n_materials; ++i) {
    const rt_material& m = desc->materials[i];
    hmat[i].kind = m.kind; hmat[i].remap = m.remap_roughness;
    hmat[i].bump = (m.kind != RT_MAT_MIX && m.bump >= 0) ? m.bump : -1;
    if (hmat[i].bump >= 0 && (uint32_t)hmat[i].bump >= desc->n_textures) { delete s; return fail(RT_ERR_INVALID, "bump texture out of range"); }
    for (int k = 0; k < 16; ++k) hmat[i].slot[k] = m.slot[k];
    if (m.kind == RT_MAT_MIX) {
      for (int side = 0; side < 2; ++side) {
        int c = m.slot[RT_SLOT_M1 + side];
        if (c < 0 || (uint32_t)c >= desc->n_materials) { delete s; return fail(RT_ERR_INVALID, "mix operand out of range"); }
        if (desc->materials[c].kind == RT_MAT_MIX)
          for (int q = 0; q < 2; ++q) { int g = desc->materials[c].slot[RT_SLOT_M1 + q]; if (g < 0 || (uint32_t)g >= desc->n_materials || desc->materials[g].kind == RT_MAT_MIX) { delete s; return fail(RT_ERR_INVALID, "mix nesting deeper than 2"); } }
      }
    }
  }
  {  // code classes: materials of one kind (and roughness remap / bump presence) whose slots hold textures of the same shape run the same code
    std::map<std::vector<int>, int> classes;
    std::function<void(int, int, std::vector<int>&)> tex_sig = [&](int id, int depth, std::vector<int>& sig) {
      if (id < 0 || (uint32_t)id >= desc->n_textures) { sig.push_back(-1); return; }
      const rt_texture& t = desc->textures[id];
      sig.push_back(t.kind);
      if (t.kind == RT_TEX_CONST) sig.push_back(t.value[0] == 0.0f ? 0 : 1);  // sigma == 0 (Lambert, not Oren-Nayar), roughness == 0 (specular lobes), ...
      if (t.kind == RT_TEX_IMAGE) { const rt_image& im = desc->images[t.image]; sig.push_back(im.trilinear ? 1 : 0); }
      if (t.kind == RT_TEX_CHECKER) sig.push_back(t.amount);
      if ((t.kind == RT_TEX_SCALE || t.kind == RT_TEX_MIX || t.kind == RT_TEX_CHECKER) && depth < 3) {
        tex_sig(t.tex1, depth + 1, sig); tex_sig(t.tex2, depth + 1, sig);
        if (t.kind == RT_TEX_MIX) tex_sig(t.amount, depth + 1, sig);
      }
    };
    // an uber material whose opacity, Kr and Kt are constants with 1 - opacity, Kr and Kt black builds Lambert + microfacet reflection only (uber.rs:76-121)
    auto uber_two_lobes = [&](const rt_material& m) {
      auto konst = [&](int id) -> const rt_texture* { return id >= 0 && (uint32_t)id < desc->n_textures && desc->textures[id].kind == RT_TEX_CONST ? &desc->textures[id] : nullptr; };
      const rt_texture *op = konst(m.slot[RT_SLOT_OPACITY]), *kr = konst(m.slot[RT_SLOT_KR]), *kt = konst(m.slot[RT_SLOT_KT]);
      if (!op || !kr || !kt) return false;
      for (int c = 0; c < 3; ++c) {
        const float o = std::max(op->value[c], 0.0f);
        if (!(std::max(1.0f - o, 0.0f) == 0.0f) || !std::isfinite(o) || !(std::max(kr->value[c], 0.0f) == 0.0f) || !(std::max(kt->value[c], 0.0f) == 0.0f)) return false;
      }
      return true;
    };
    std::function<void(int, int, std::vector<int>&)> mat_sig = [&](int id, int depth, std::vector<int>& sig) {
      const rt_material& m = desc->materials[id];
      sig.push_back(1000 + m.kind); sig.push_back(m.remap_roughness ? 1 : 0);
      for (int k = 0; k < RT_SLOT_M1; ++k) tex_sig(m.slot[k], 0, sig);
      if (m.kind == RT_MAT_MIX) { if (depth < 2) { mat_sig(m.slot[RT_SLOT_M1], depth + 1, sig); mat_sig(m.slot[RT_SLOT_M2], depth + 1, sig); } }
      else { sig.push_back(m.kind == RT_MAT_DISNEY ? m.slot[RT_SLOT_M1] : 0); tex_sig(m.bump, 0, sig); }
      if (m.kind == RT_MAT_UBER) sig.push_back(uber_two_lobes(m) ? 1 : 0);
    };
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
      std::vector<int> sig; mat_sig((int)i, 0, sig);
      auto it = classes.find(sig);
      if (it == classes.end()) it = classes.emplace(sig, (int)classes.size()).first;
      hmat[i].code_class = it->second;
    }
    s->n_code_classes = (unsigned)classes.size();
    // classes the register-resident front-end can shade (matte, sigma == 0, no bump, Kd any texture: SingleLambertT) get the lowest ids, so that
    // after binning they are one contiguous range of the queue
    auto is_const = [&](int id) { return id >= 0 && (uint32_t)id < desc->n_textures && desc->textures[id].kind == RT_TEX_CONST; };
    // the register-resident front-ends evaluate constants in place and image maps through tex_image_q; a material with any other texture shape in a slot
    // (scale / mix / checkerboard / uv / fbm) is shaded by the generic kernel, whose evaluator handles them all
    auto leaf_slots = [&](const rt_material& m) {
      for (int k = 0; k < RT_SLOT_M1; ++k) {
        const int id = m.slot[k];
        if (id < 0 || (uint32_t)id >= desc->n_textures) continue;
        if (desc->textures[id].kind != RT_TEX_CONST && desc->textures[id].kind != RT_TEX_IMAGE) return false;
      }
      return true;
    };
    std::vector<int> lambert(classes.size(), 0), remap(classes.size(), -1);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
      const rt_material& m = desc->materials[i];
      lambert[hmat[i].code_class] = m.kind == RT_MAT_MATTE && m.slot[RT_SLOT_KD] >= 0 && is_const(m.slot[RT_SLOT_SIGMA]) && m.bump < 0 &&
                                    desc->textures[m.slot[RT_SLOT_SIGMA]].value[0] <= 0.0f && leaf_slots(m);
    }
    // then the classes of the two-lobe front-end (SmallBsdfT<false>): matte with sigma > 0, plastic, metal, mirror; then of its wide form: glass,
    // substrate, opaque uber; no bump map
    std::vector<int> small(classes.size(), 0), wide(classes.size(), 0);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
      const rt_material& m = desc->materials[i];
      small[hmat[i].code_class] = !lambert[hmat[i].code_class] && m.bump < 0 && leaf_slots(m) && (m.kind == RT_MAT_MATTE || m.kind == RT_MAT_PLASTIC || m.kind == RT_MAT_METAL || m.kind == RT_MAT_MIRROR);
      wide[hmat[i].code_class] = m.bump < 0 && leaf_slots(m) && (m.kind == RT_MAT_GLASS || m.kind == RT_MAT_SUBSTRATE || (m.kind == RT_MAT_UBER && uber_two_lobes(m)));
    }
    int next = 0;
    for (size_t c = 0; c < classes.size(); ++c) if (lambert[c]) remap[c] = next++;
    s->n_lambert_classes = (unsigned)next;
    for (size_t c = 0; c < classes.size(); ++c) if (small[c]) remap[c] = next++;
    s->n_small_classes = (unsigned)next - s->n_lambert_classes;
    for (size_t c = 0; c < classes.size(); ++c) if (wide[c]) remap[c] = next++;
    s->n_wide_classes = (unsigned)next - s->n_lambert_classes - s->n_small_classes;
    for (size_t c = 0; c < classes.size(); ++c) if (!lambert[c] && !small[c] && !wide[c]) remap[c] = next++;
    for (uint32_t i = 0; i < desc->n_materials; ++i) hmat[i].code_class = remap[hmat[i].code_class];
  }
  TRY_RC(upload(s->materials, hmat.data(), hmat.size() * sizeof(DMaterial)));
  {  // per primitive: the code class of its material (bit 15: a quadric) - what the vertex queue is binned by (k_bin_count), ONE two-byte gather instead of the primitive's
     // 128-byte shade record and then its material (round 6)
    std::vector<uint16_t> pc(desc->n_tris, 0);
    for (size_t i = 0; i < desc->n_tris; ++i) {
      const rt_tri_meta& m = desc->tri_meta[i];
      if (m.flags & RT_PRIM_INSTANCE) continue;
      const int c = (m.material >= 0 && (uint32_t)m.material < desc->n_materials) ? hmat[m.material].code_class : 0;
      pc[i] = (uint16_t)(std::min(c, 0x7ffe) | ((m.flags & RT_PRIM_SPHERE) ? 0x8000 : 0));
    }
    TRY_RC(upload(s->prim_class, pc.data(), pc.size() * 2));
  }
  // lights (+ env distributions in one blob)
  const uint32_t n_all_lights = desc->n_lights + desc->n_unlisted_lights;  // sampled lights, then the emitters no light list holds
  s->h_lights.resize(n_all_lights);
  {
    size_t total = 0;
    for (uint32_t i = 0; i < n_all_lights; ++i) {
      const rt_light& l = desc->lights[i];
      if (l.kind == RT_LIGHT_INFINITE) total += 2 * (size_t)l.dist_nv * (l.dist_nu + 1) + (size_t)l.dist_nv * 3 + 1;
    }
    std::vector<float> blob(total + 4);
    TRY_RC([&]() { return s->dist.ensure(blob.size() * 4) == hipSuccess ? RT_OK : fail(RT_ERR_OOM, "dist alloc"); }());
    // guide tables of the environment maps' CDF searches (DLight::guide): 2^glog buckets per row (guide_quarters)
    auto guide_log = [](int n) { const long q = guide_quarters(); int g = 0; while (g < 16 && (long)(2 << g) * q <= 4l * n) ++g; return g; };
    auto guide_row = [](const float* cdf, int n, int glog, unsigned short* out) {  // out[k] = #{i in [0, n] : cdf[i] <= k / 2^glog}
      const int G = 1 << glog; int i = 0;
      for (int k = 0; k <= G; ++k) { const float x = (float)k / (float)G; while (i <= n && cdf[i] <= x) ++i; out[k] = (unsigned short)i; }
    };
    size_t guide_total = 0;
    for (uint32_t i = 0; i < n_all_lights; ++i) {
      const rt_light& l = desc->lights[i];
      if (l.kind == RT_LIGHT_INFINITE) {
        if (l.dist_nu < 1 || l.dist_nv < 1 || l.dist_nu > 65534 || l.dist_nv > 65534 || !l.dist_cdf || !l.marg_cdf) { delete s; return fail(RT_ERR_INVALID, "infinite light tables missing or larger than 65534 entries per row"); }
        guide_total += (size_t)l.dist_nv * ((1u << guide_log(l.dist_nu)) + 1) + ((1u << guide_log(l.dist_nv)) + 1);
      }
    }
    std::vector<unsigned short> gblob(guide_total + 1);
    TRY_RC([&]() { return s->guides.ensure(gblob.size() * 2) == hipSuccess ? RT_OK : fail(RT_ERR_OOM, "guide alloc"); }());
    size_t gbase = 0;
    size_t base = 0; int n_inf = 0;
    for (uint32_t i = 0; i < n_all_lights; ++i) {
      const rt_light& l = desc->lights[i]; DLight& d = s->h_lights[i];
      memset(&d, 0, sizeof(d));
      d.kind = l.kind; d.prim = l.prim; d.rgb[0] = l.rgb[0]; d.rgb[1] = l.rgb[1]; d.rgb[2] = l.rgb[2]; d.two_sided = l.two_sided;
      d.vec[0] = l.vec[0]; d.vec[1] = l.vec[1]; d.vec[2] = l.vec[2]; d.area = l.area; d.world_radius = l.world_radius; d.image = l.image;
      memcpy(d.l2w, l.l2w, 48); memcpy(d.w2l, l.w2l, 48);
      if (l.kind == RT_LIGHT_DIFFUSE_AREA && (l.prim < 0 || (uint32_t)l.prim >= desc->n_tris)) { delete s; return fail(RT_ERR_INVALID, "area light prim out of range"); }
      if (i >= desc->n_lights && l.kind != RT_LIGHT_DIFFUSE_AREA) { delete s; return fail(RT_ERR_INVALID, "an unlisted emitter must be a diffuse area light"); }
      if (l.kind == RT_LIGHT_INFINITE) {
        if (n_inf >= 4) { delete s; return fail(RT_ERR_INVALID, "more than 4 infinite lights"); }
        if (l.image < 0 || (uint32_t)l.image >= desc->n_images) { delete s; return fail(RT_ERR_INVALID, "infinite light image out of range"); }
        s->d.infinite_ids[n_inf++] = (int)i;
        d.nu = l.dist_nu; d.nv = l.dist_nv; d.mfunc_int = l.marg_func_int;
        float* db = s->dist.as<float>();
        size_t nfunc = (size_t)l.dist_nv * l.dist_nu, ncdf = (size_t)l.dist_nv * (l.dist_nu + 1);
        d.cf = db + base;  // (cdf, func) pairs, nu + 1 per row
        for (int r = 0; r < l.dist_nv; ++r)
          for (int i = 0; i <= l.dist_nu; ++i) {
            blob[base++] = l.dist_cdf[(size_t)r * (l.dist_nu + 1) + i];
            blob[base++] = i < l.dist_nu ? l.dist_func[(size_t)r * l.dist_nu + i] : 0.0f;
          }
        (void)nfunc;
        memcpy(&blob[base], l.dist_func_int, (size_t)l.dist_nv * 4); d.func_int = db + base; base += l.dist_nv;
        memcpy(&blob[base], l.marg_func, (size_t)l.dist_nv * 4); d.mfunc = db + base; base += l.dist_nv;
        memcpy(&blob[base], l.marg_cdf, ((size_t)l.dist_nv + 1) * 4); d.mcdf = db + base; base += (size_t)l.dist_nv + 1;
        d.glog = guide_log(l.dist_nu); d.mglog = guide_log(l.dist_nv);
        const size_t gw = ((size_t)1 << d.glog) + 1;
        d.guide = s->guides.as<unsigned short>() + gbase;
        for (int r = 0; r < l.dist_nv; ++r) guide_row(l.dist_cdf + (size_t)r * (l.dist_nu + 1), l.dist_nu, d.glog, &gblob[gbase + (size_t)r * gw]);
        gbase += (size_t)l.dist_nv * gw;
        d.mguide = s->guides.as<unsigned short>() + gbase;
        guide_row(l.marg_cdf, l.dist_nv, d.mglog, &gblob[gbase]);
        gbase += ((size_t)1 << d.mglog) + 1;
      }
    }
    s->d.n_infinite = n_inf;
    if (hipMemcpy(s->dist.p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { delete s; return fail(RT_ERR_HIP, "dist upload"); }
    if (hipMemcpy(s->guides.p, gblob.data(), gblob.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { delete s; return fail(RT_ERR_HIP, "guide upload"); }
    TRY_RC(upload(s->lights, s->h_lights.data(), s->h_lights.size() * sizeof(DLight)));
  }
#undef TRY_RC
  DScene& d = s->d;
  d.nodes = s->nodes.as<float4>(); d.n_nodes = desc->n_nodes;
  d.tri_p = s->tri_p.as<float4>(); d.n_tris = desc->n_tris;
  d.tri_n = s->tri_n.as<float>(); d.tri_uv = s->tri_uv.as<float>(); d.tri_s = s->tri_s.as<float>();
  d.tri_alpha = s->tri_alpha.p ? s->tri_alpha.as<int2>() : nullptr;
  d.spheres = s->has_spheres ? s->spheres.as<DSphere>() : nullptr;
  d.instances = s->has_instances ? s->instances.as<DInstance>() : nullptr; d.n_instances = s->has_instances ? desc->n_instances : 0u; d.n_top_prims = n_top_prims;
  d.textures = s->textures.as<DTexture>(); d.images = s->images.as<DImage>(); d.materials = s->materials.as<DMaterial>(); d.lights = s->lights.as<DLight>();
  d.n_materials = (int)desc->n_materials; d.n_textures = (int)desc->n_textures; d.n_images = (int)desc->n_images;
  d.n_lights = (int)desc->n_lights;
  d.wb_min = f3{desc->nodes[0].bmin[0], desc->nodes[0].bmin[1], desc->nodes[0].bmin[2]};
  d.wb_max = f3{desc->nodes[0].bmax[0], desc->nodes[0].bmax[1], desc->nodes[0].bmax[2]};
  d.ld_uniform = 1; d.nvox[0] = d.nvox[1] = d.nvox[2] = 1; d.ld_glog = -1; d.ld_guide = nullptr;
  d.needs_differentials = 0;
#ifdef RT_ABLATE
  d.dbg = env_int("RTX_DBG", 0);
#endif
  for (uint32_t i = 0; i < desc->n_textures; ++i) {
    const rt_texture& t = desc->textures[i];
    if (t.kind == RT_TEX_IMAGE || t.kind == RT_TEX_FBM || (t.kind == RT_TEX_CHECKER && t.amount != 0)) d.needs_differentials = 1;
  }
  for (uint32_t i = 0; i < desc->n_materials; ++i) if (desc->materials[i].kind != RT_MAT_MIX && desc->materials[i].bump >= 0) d.needs_differentials = 1;  // bump() reads dudx..
  s->lambert_only = true;
  for (uint32_t i = 0; i < desc->n_materials; ++i) {
    const rt_material& m = desc->materials[i];
    auto is_const = [&](int id) { return id >= 0 && (uint32_t)id < desc->n_textures && desc->textures[id].kind == RT_TEX_CONST; };
    if (m.kind != RT_MAT_MATTE || !is_const(m.slot[RT_SLOT_KD]) || !is_const(m.slot[RT_SLOT_SIGMA]) || m.bump >= 0) { s->lambert_only = false; break; }
    const float sg = desc->textures[m.slot[RT_SLOT_SIGMA]].value[0];
    if (!(sg <= 0.0f)) { s->lambert_only = false; break; }  // clamp(sigma, 0, 1) == 0 (matte.rs:51)
  }
  s->lambert_materials = true;  // the same with Kd any texture: SingleLambertT<true>
  for (uint32_t i = 0; i < desc->n_materials; ++i) {
    const rt_material& m = desc->materials[i];
    const bool sigma_zero = m.slot[RT_SLOT_SIGMA] >= 0 && (uint32_t)m.slot[RT_SLOT_SIGMA] < desc->n_textures && desc->textures[m.slot[RT_SLOT_SIGMA]].kind == RT_TEX_CONST &&
                            desc->textures[m.slot[RT_SLOT_SIGMA]].value[0] <= 0.0f;
    const int kd = m.slot[RT_SLOT_KD];
    const bool kd_leaf = kd >= 0 && (uint32_t)kd < desc->n_textures && (desc->textures[kd].kind == RT_TEX_CONST || desc->textures[kd].kind == RT_TEX_IMAGE);  // what k_shade<3> evaluates (tex_eval_leaf)
    if (m.kind != RT_MAT_MATTE || !kd_leaf || !sigma_zero || m.bump >= 0) { s->lambert_materials = false; break; }
  }
  for (uint32_t i = 0; i < desc->n_lights; ++i) if (desc->lights[i].kind != RT_LIGHT_DIFFUSE_AREA) s->lambert_only = false;
  for (uint32_t i = 0; i < desc->n_lights; ++i)
    if (desc->lights[i].kind == RT_LIGHT_DIFFUSE_AREA && (desc->tri_meta[desc->lights[i].prim].flags & RT_TRI_HAS_ALPHA)) s->masked_emitters = true;
  if (s->has_spheres || s->has_instances) s->masked_emitters = true;  // quadric / instance hits, quadric emitters, masked emitters: the GENERAL instantiations of the shade kernels
  s->lean_shade = !s->masked_emitters;
  for (uint32_t i = 0; i < desc->n_lights; ++i) if (desc->lights[i].kind != RT_LIGHT_DIFFUSE_AREA) s->lean_shade = false;
  for (uint32_t i = 0; i < desc->n_textures; ++i) if (desc->textures[i].kind != RT_TEX_CONST) s->lean_shade = false;
  // The LEAN forms with sphere lights (QLIGHTS): constant textures, every light a diffuse area light on a triangle or on a Sphere that no triangle reaches
  // into, no masks, no instances. Vertices on quadrics are binned apart and shaded by the generic GENERAL kernel (route_quadric_hits), so the scene must be
  // one whose shade queue is binned (several material classes - rt_render checks that).
  s->lean_qlights = false;
  if (s->has_spheres && !s->has_instances && !s->has_masks) {
    bool ok = true;
    for (uint32_t i = 0; i < desc->n_lights; ++i) if (desc->lights[i].kind != RT_LIGHT_DIFFUSE_AREA) ok = false;
    for (uint32_t i = 0; i < desc->n_textures; ++i) if (desc->textures[i].kind != RT_TEX_CONST) ok = false;
    s->lean_qlights = ok && sphere_lights_clear(desc);
  }
  if (s->masked_emitters) s->lambert_only = false;  // (the constant-matte kernel has no GENERAL form: such scenes shade through the Lambert front-end k_shade<3, true>)
  s->n_materials = desc->n_materials;
  s->small = desc->n_nodes <= RT_SMALL_NODES && desc->n_tris <= RT_SMALL_TRIS && !s->has_instances;  // quadrics and masked triangles: the GENERAL form of the LDS kernel
  int max_obj_depth = 0;
  {  // tree height bounds the number of simultaneously pending stack entries
    // one tree: nodes [base, base + nn), child offsets relative to base, leaf ranges within its np primitives
    auto tree_depth = [&](uint32_t base, uint32_t nn, uint32_t np, int& maxd) -> bool {
      std::vector<int> depth(nn, 0); maxd = 0;
      for (uint32_t i = 0; i < nn; ++i) {
        const rt_bvh_node& n = desc->nodes[base + i];
        if (n.n_prims == 0) {
          if (i + 1 >= nn || n.offset >= nn || n.offset <= i) return false;
          depth[i + 1] = depth[i] + 1; depth[n.offset] = depth[i] + 1;
        } else if ((uint64_t)n.offset + n.n_prims > np) return false;
        if (depth[i] > maxd) maxd = depth[i];
      }
      return true;
    };
    int maxd = 0;
    if (!tree_depth(0, n_top_nodes, n_top_prims, maxd)) { delete s; return fail(RT_ERR_INVALID, "malformed BVH"); }
    if (maxd + 1 > 64) { delete s; return fail(RT_ERR_INVALID, "BVH deeper than the 64-entry traversal stack"); }
    s->stack_depth = maxd + 1;
    int max_obj = 0;
    for (uint32_t k = 0; k < desc->n_instances; ++k) {
      const rt_instance& in = desc->instances[k];
      int od = 0;
      if (in.n_nodes && (!tree_depth(in.node_base, in.n_nodes, in.n_prims, od) || od + 1 > 64)) { delete s; return fail(RT_ERR_INVALID, "malformed or too deep object BVH"); }
      if (in.n_nodes) max_obj = std::max(max_obj, od + 1);
    }
    // an object's walk uses the entries of the lane's stack column above the top level's pending ones: the column holds both
    // The reference gives each BVH a 64-entry stack of its own (bvh/mod.rs:374), so a 36-deep top level over a 30-deep object is a valid scene: past 64 entries
    // in one column the scene is traced by the one-node-per-step kernel with a 128-entry column (k_trace_big<.., 64, 128>; no pair / four-wide records)
    s->stack_depth += max_obj; max_obj_depth = max_obj;
    s->deep_column = s->stack_depth > 64;
  }
  d.pairs = nullptr; d.quads = nullptr; d.obj_pairs = 0; d.obj_general = s->obj_general ? 1 : 0; d.link8 = nullptr; d.link8_full = nullptr;
  // mid-size scenes (round 5): too large for the 256-node LDS kernels, small enough for one workgroup's 160 KB - occlusion rays walk link tables in LDS (k_trace<.., MID>)
  s->mid = !s->small && !s->general_prims && !s->has_instances && desc->n_nodes <= RT_MID_NODES && desc->n_tris <= RT_MID_TRIS;
  for (uint32_t i = 0; i < desc->n_nodes && s->mid; ++i) if (desc->nodes[i].n_prims > 15) s->mid = false;  // (the link word's count field)
  if (s->mid) {  // the kernels declare 157.7 KB of LDS: were a device (or a driver's reservation) to leave a workgroup less, they could not launch - the HBM kernels then
    int fit_any = 0, fit_closest = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_any, (const void*)(k_trace<true, false, 1024, 16, 0, 1>), 1024, 0) != hipSuccess) fit_any = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_closest, (const void*)(k_trace<false, false, 1024, 16, 0, 1>), 1024, 0) != hipSuccess) fit_closest = 0;
    (void)hipGetLastError();
    if (fit_any < 1 || fit_closest < 1) s->mid = false;
  }
  if (s->small || s->mid) {
    // link tables (round 5: per octant and node where the stackless walk goes on, over all nodes and over the nodes a calibration on
    // synthetic path rays found worth testing): rtx_link_tables.h
    RtLinkTables lt;
    rt_build_link_tables(desc, s->mid, false, lt);
    s->lds_nodes_tested = lt.nodes_tested;
    int rcl = upload(s->link8_full, lt.link_full.data(), lt.link_full.size() * 4);
    if (rcl == RT_OK) rcl = upload(s->link8, lt.link_kept.data(), lt.link_kept.size() * 4);
    if (rcl != RT_OK) { delete s; return rcl; }
    d.link8_full = s->link8_full.as<unsigned>(); d.link8 = s->link8.as<unsigned>();
  }
  if (!s->small && !s->deep_column) {  // LDS-resident scenes keep the one-node-per-step loop: the pair form measured no faster there
    // With object instances the records cover the top-level tree (objects are walked one node per step, their child offsets are relative to the object).
    // A leaf of a GENERAL scene that holds anything but plain triangles carries RT_PAIR_GENERAL.
    const uint32_t n_pair_nodes = n_top_nodes;
    const bool gen = s->general_prims;
    auto general_leaf = [&](const rt_bvh_node& n) {
      for (uint32_t t = n.offset; t < n.offset + n.n_prims; ++t)
        if (desc->tri_meta[t].flags & (RT_TRI_HAS_ALPHA | RT_TRI_HAS_SHADOW_ALPHA | RT_PRIM_SPHERE | RT_PRIM_INSTANCE)) return true;
      return false;
    };
    // child-pair records for k_trace_pair: {A.min.xyz, A.max.x} {A.max.yz, code A, code B} {B.min.xyz, B.max.x} {B.max.yz, -, -}
    // (a root that is itself a leaf - every centroid coincides - is never seen by code_of(): its count must fit the 5-bit field too)
    bool ok = n_pair_nodes < (1u << 29) && n_top_prims < (gen ? (1u << 25) : (1u << 26)) && desc->nodes[0].n_prims <= 32;
    auto code_of = [&](uint32_t c, bool& good) -> uint32_t {
      const rt_bvh_node& n = desc->nodes[c];
      if (n.n_prims > 0) { if (n.n_prims > 32) good = false; return 0x80000000u | n.offset | ((uint32_t)(n.n_prims - 1) << 26) | (gen && general_leaf(n) ? RT_PAIR_GENERAL : 0u); }
      return c | ((uint32_t)n.axis << 29);
    };
    std::vector<float> pr((size_t)(s->has_instances ? desc->n_nodes : n_pair_nodes) * 16, 0.0f);  // (with instances: the objects' records behind the top level's)
    for (uint32_t i = 0; i < n_pair_nodes && ok; ++i) {
      const rt_bvh_node& n = desc->nodes[i];
      if (n.n_prims != 0) continue;
      const rt_bvh_node& a = desc->nodes[i + 1]; const rt_bvh_node& b = desc->nodes[n.offset];
      float* q = pr.data() + (size_t)i * 16;
      uint32_t ca = code_of(i + 1, ok), cb = code_of(n.offset, ok);
      q[0] = a.bmin[0]; q[1] = a.bmin[1]; q[2] = a.bmin[2]; q[3] = a.bmax[0]; q[4] = a.bmax[1]; q[5] = a.bmax[2]; memcpy(q + 6, &ca, 4); memcpy(q + 7, &cb, 4);
      q[8] = b.bmin[0]; q[9] = b.bmin[1]; q[10] = b.bmin[2]; q[11] = b.bmax[0]; q[12] = b.bmax[1]; q[13] = b.bmax[2];
    }
    // the objects' trees as child pairs as well (nested_pair_walk): records at the nodes' global indices, child codes LOCAL to the object - node indices and leaf
    // ranges are relative to the object's bases in its flattened tree already. Objects hold plain triangles (checked above).
    bool obj_ok = ok && s->has_instances;
    if (obj_ok) {
      std::vector<char> done(desc->n_nodes, 0);
      for (uint32_t k = 0; k < desc->n_instances && obj_ok; ++k) {
        const rt_instance& in = desc->instances[k];
        if (in.n_nodes == 0 || done[in.node_base]) continue;
        done[in.node_base] = 1;
        if (in.n_nodes >= (1u << 29) || in.n_prims >= (1u << 26)) { obj_ok = false; break; }
        auto local_code = [&](uint32_t c) -> uint32_t {
          const rt_bvh_node& n = desc->nodes[in.node_base + c];
          if (n.n_prims > 0) { if (n.n_prims > 32) obj_ok = false; return 0x80000000u | n.offset | ((uint32_t)(n.n_prims - 1) << 26); }
          return c | ((uint32_t)n.axis << 29);
        };
        if (desc->nodes[in.node_base].n_prims > 32) obj_ok = false;
        for (uint32_t i = 0; i < in.n_nodes && obj_ok; ++i) {
          const rt_bvh_node& n = desc->nodes[in.node_base + i];
          if (n.n_prims != 0) continue;
          const rt_bvh_node& a = desc->nodes[in.node_base + i + 1]; const rt_bvh_node& b = desc->nodes[in.node_base + n.offset];
          float* q = pr.data() + (size_t)(in.node_base + i) * 16;
          const uint32_t ca = local_code(i + 1), cb = local_code(n.offset);
          q[0] = a.bmin[0]; q[1] = a.bmin[1]; q[2] = a.bmin[2]; q[3] = a.bmax[0]; q[4] = a.bmax[1]; q[5] = a.bmax[2]; memcpy(q + 6, &ca, 4); memcpy(q + 7, &cb, 4);
          q[8] = b.bmin[0]; q[9] = b.bmin[1]; q[10] = b.bmin[2]; q[11] = b.bmax[0]; q[12] = b.bmax[1]; q[13] = b.bmax[2];
        }
      }
    }
    if (ok) {
      int rc2 = upload(s->pairs, pr.data(), pr.size() * 4);
      if (rc2 != RT_OK) { delete s; return rc2; }
      d.pairs = s->pairs.as<float4>(); s->use_pairs = true;
      d.obj_pairs = (obj_ok && !s->obj_general) ? 1 : 0;  // (nested_pair_walk tests plain triangles only)
    }
    d.top_pairs = nullptr; d.n_top = 0;
    if (ok && n_pair_nodes < (1u << 28) && desc->nodes[0].n_prims == 0 && !s->has_instances) {  // (an object's walk needs a contiguous stack column: k_trace_pair)
      // the first levels of the tree, breadth first, for k_trace_top: up to RT_TOP_MAX interior nodes; a child that is itself one of them is named by its slot
      std::vector<uint32_t> top; std::vector<int> slot_of(n_pair_nodes, -1);
      top.push_back(0); slot_of[0] = 0;
      for (size_t head = 0; head < top.size(); ++head) {
        const uint32_t P = top[head]; const uint32_t kids[2] = {P + 1, desc->nodes[P].offset};
        for (uint32_t c : kids)
          if (desc->nodes[c].n_prims == 0 && top.size() < RT_TOP_MAX) { slot_of[c] = (int)top.size(); top.push_back(c); }
      }
      std::vector<float> tp(top.size() * 16);
      for (size_t k = 0; k < top.size(); ++k) {
        float* q = tp.data() + 16 * k; memcpy(q, pr.data() + (size_t)top[k] * 16, 64);
        for (int side = 0; side < 2; ++side) {
          uint32_t code; memcpy(&code, q + 6 + side, 4);
          if (!(code & 0x80000000u) && slot_of[code & 0x0fffffffu] >= 0) { code = (code & 0x60000000u) | RT_PAIR_TOP | (uint32_t)slot_of[code & 0x0fffffffu]; memcpy(q + 6 + side, &code, 4); }
        }
      }
      int rc2 = upload(s->top_pairs, tp.data(), tp.size() * 4);
      if (rc2 != RT_OK) { delete s; return rc2; }
      d.top_pairs = s->top_pairs.as<float4>(); d.n_top = (unsigned)top.size(); s->use_top = true;
      // Measured (scripts/ab_bench.sh, one box; k_trace_top vs k_trace_pair): shadow rays -9 % (S2) .. -12 % (S4); closest-hit rays -8 % on S4 (25 MB of
      // pair records), +2 % on S3 (83 KB: L1-resident either way) and +6 % on S2 (128 MB: a sixth wave per SIMD evicts more of the tree from L2 than
      // it hides). 64, 128 or 256 LDS-resident nodes measured the same: the top of the tree was already served by the CU's L1 - what the kernel
      // gains is its sixth wave per SIMD, and that pays where the tree fits the 32 MB of L2 without fitting an L1.
      const size_t pair_bytes = (size_t)n_pair_nodes * 64;
      s->top_for_closest = pair_bytes <= ((size_t)64 << 20);  // measured with the gated leaf phase: S3 (0.1 MB) 81 -> 78 ms, S4 (13 MB) 1743 -> 1659 ms, S2 (67 MB) 94 -> 102 ms
    }
    // four-wide records for the any-hit kernel (k_trace_quad): an interior node's grandchildren (a leaf child stands for itself), 128 B
    // per node: 24 floats = boxes of slots 0..3 (slots 0,1: first child's part, 2,3: second child's), 4 codes (0xffffffff = empty slot),
    // {axis of the first child | axis of the second child << 2}.
    if (ok) {
      std::vector<float> qr((size_t)n_pair_nodes * 32, 0.0f);
      std::vector<int> need(n_pair_nodes, 0);  // stack entries the four-wide walk can have pending below this node
      for (uint32_t i = n_pair_nodes; i-- > 0;) {
        const rt_bvh_node& n = desc->nodes[i];
        if (n.n_prims != 0) continue;
        float* q = qr.data() + (size_t)i * 32;
        uint32_t codes[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, axes = 0; int n_entries = 0, deepest = 0;
        const uint32_t child[2] = {i + 1, n.offset};
        for (int side = 0; side < 2; ++side) {
          const rt_bvh_node& c = desc->nodes[child[side]];
          uint32_t ids[2]; int cnt;
          if (c.n_prims != 0) { ids[0] = child[side]; cnt = 1; }
          else { ids[0] = child[side] + 1; ids[1] = c.offset; cnt = 2; axes |= (uint32_t)c.axis << (2 * side); }
          for (int k = 0; k < cnt; ++k) {
            const rt_bvh_node& g = desc->nodes[ids[k]];
            float* bq = q + 6 * (2 * side + k);
            bq[0] = g.bmin[0]; bq[1] = g.bmin[1]; bq[2] = g.bmin[2]; bq[3] = g.bmax[0]; bq[4] = g.bmax[1]; bq[5] = g.bmax[2];
            codes[2 * side + k] = code_of(ids[k], ok);
            n_entries += 1; if (need[ids[k]] > deepest) deepest = need[ids[k]];
          }
        }
        memcpy(q + 24, codes, 16); memcpy(q + 28, &axes, 4);
        { const uint32_t ca = code_of(child[0], ok), cb = code_of(child[1], ok); memcpy(q + 29, &ca, 4); memcpy(q + 30, &cb, 4); }  // the children themselves: what the closest-hit step pushes for the far side
        need[i] = deepest + n_entries - 1;
      }
      s->quad_stack_depth = need[0] + 1 + max_obj_depth;  // (+ the deepest object's walk above the pending entries)
      // (closest-hit rays through the same records - near side first, exact - were built in round 5 and measured slower: S2 91.0 -> 95.9 ms, S4 1371 -> 1554; MEASUREMENTS R5)
      if (ok && s->quad_stack_depth <= 32) {  // (any hit: beyond the 32-entry LDS stack the larger stack costs more residency than the wider step returns)
        int rc2 = upload(s->quads, qr.data(), qr.size() * 4);
        if (rc2 != RT_OK) { delete s; return rc2; }
        d.quads = s->quads.as<float4>(); s->use_quads = true;
      }
    }
  }
  if (const size_t nb = tmin_stack_bytes(s)) {
    if (s->tmin_stack.ensure(nb) != hipSuccess) { delete s; return fail(RT_ERR_OOM, "traversal stack allocation failed"); }
  }
  if (const size_t nb = deep_stack_bytes(s)) {
    if (s->deep_stack.ensure(nb) != hipSuccess) { delete s; return fail(RT_ERR_OOM, "traversal stack allocation failed"); }
  }
  fill_ewa_lut();
  // per-triangle shade records and per-emitter constants, computed on the device by the per-vertex path's own expressions
  if (s->tri_rec.ensure((size_t)desc->n_tris * 128) != hipSuccess) { delete s; return fail(RT_ERR_OOM, "shade record allocation failed"); }
  d.tri_rec = s->tri_rec.as<float4>();
  hipLaunchKernelGGL(k_tri_records, dim3((desc->n_tris + 255u) / 256u), dim3(256), 0, nullptr, d, s->tri_rec.as<float4>());
  d.n_lights_all = (int)n_all_lights;
  s->lds_records = s->small && !s->has_instances && !s->has_spheres && desc->n_tris <= RT_SMALL_TRIS && n_all_lights <= RT_LDS_LIGHTS && desc->n_materials <= RT_LDS_MATERIALS && desc->n_textures <= RT_LDS_TEXTURES && !(env_is("RTX_SHADE_LDSREC", '0'));  // k_shade<1, .., LDSREC> (RTX_SHADE_LDSREC=0: measurement knob)
  s->lds_records_q = desc->n_tris <= RT_SMALL_TRIS && !s->has_instances && n_all_lights <= RT_LDS_LIGHTS && desc->n_materials <= RT_LDS_MATERIALS && desc->n_textures <= RT_LDS_TEXTURES && !(env_is("RTX_SHADE_LDSREC", '0'));
  s->lds_tables = n_all_lights <= RT_LDS_LIGHTS && desc->n_materials <= RT_LDS_MATERIALS && desc->n_textures <= RT_LDS_TEXTURES && desc->n_images <= RT_LDS_IMAGES && !(env_is("RTX_SHADE_LDSREC", '0'));
  s->lds_mats = desc->n_materials <= RT_LDS_MATERIALS && desc->n_textures <= RT_LDS_TEXTURES && !(env_is("RTX_SHADE_LDSREC", '0'));  // the LEAN forms: material and texture tables in LDS
  if (n_all_lights) hipLaunchKernelGGL(k_light_consts, dim3((n_all_lights + 255u) / 256u), dim3(256), 0, nullptr, d, s->lights.as<DLight>(), (int)n_all_lights);
  if (hipGetLastError() != hipSuccess) { delete s; return fail(RT_ERR_HIP, "constant precomputation launch failed"); }
  if (s->self.ensure(sizeof(DScene)) != hipSuccess) { delete s; return fail(RT_ERR_OOM, "scene record allocation failed"); }
  d.prim_class = s->prim_class.as<unsigned short>();
  d.route_quadric_hits = (s->lean_qlights && s->n_code_classes > 1 && !s->lambert_materials) ? 1 : 0;  // (the condition of rt_render's use_bins)
  d.self = s->self.as<DScene>();
  HIP_TRY(hipMemcpy(s->self.p, &d, sizeof(DScene), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  *out = s;
  return RT_OK;
}

#include "rtx_bvh_build.h"
#include "rtx_ingest_build.h"

extern "C" void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  delete s;
}
__global__ void k_offset_ray_origin(const float* __restrict__ p, const float* __restrict__ pe, const float* __restrict__ n, const float* __restrict__ w, unsigned long long count, float* __restrict__ out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const f3 r = offset_ray_origin(mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), mk3(pe[3 * i], pe[3 * i + 1], pe[3 * i + 2]), mk3(n[3 * i], n[3 * i + 1], n[3 * i + 2]), mk3(w[3 * i], w[3 * i + 1], w[3 * i + 2]));
  out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
extern "C" int rt_offset_ray_origin(const float* p, const float* p_error, const float* n, const float* w, uint64_t count, float* out) {
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  if (!p || !p_error || !n || !w || !out || count == 0) return fail(RT_ERR_INVALID, "bad rt_offset_ray_origin arguments");
  DevBuf b[5];
  const size_t bytes = (size_t)count * 12;
  const float* src[4] = {p, p_error, n, w};
  for (int k = 0; k < 5; ++k) HIP_TRY(b[k].ensure(bytes));
  for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpy(b[k].p, src[k], bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_offset_ray_origin, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, b[0].as<float>(), b[1].as<float>(), b[2].as<float>(), b[3].as<float>(), (unsigned long long)count, b[4].as<float>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, b[4].p, bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}
extern "C" int rt_link_tables(const rt_scene_desc* desc, int32_t mid, uint32_t* link_kept, uint32_t* link_full, uint64_t capacity_words, double* stats) {
  if (!desc || !desc->nodes || desc->n_nodes == 0 || !desc->tri_p || !desc->tri_meta || !link_kept || !link_full) return fail(RT_ERR_INVALID, "null argument");
  const size_t words = (size_t)9 * desc->n_nodes + 9;
  if (desc->n_nodes > (mid ? RT_MID_NODES : RT_SMALL_NODES) || desc->n_tris > (uint64_t)(mid ? RT_MID_TRIS : RT_SMALL_TRIS) || desc->n_instances != 0) return fail(RT_ERR_INVALID, "not an LDS-sized scene");
  if (capacity_words < words) return fail(RT_ERR_INVALID, "link table capacity too small");
  {  // the tree as rt_scene_create requires it: pre-order, first child at i + 1, second child right behind the first one's subtree, leaf ranges inside the primitives
    const uint32_t nn = desc->n_nodes;
    std::vector<uint32_t> size(nn, 1u);
    for (uint32_t i = nn; i-- > 0;) {
      const rt_bvh_node& n = desc->nodes[i];
      if (n.n_prims != 0) {
        if ((uint64_t)n.offset + n.n_prims > desc->n_tris) return fail(RT_ERR_INVALID, "leaf range out of bounds");
        // a leaf's link word holds its primitive count above the first primitive's bits (RT_LINK_LEAF_N): 4 bits in a mid-size scene's word, 8 otherwise
        if (n.n_prims >= (1u << (15 - RT_LINK_OFF_BITS(mid ? RT_MID_NODES : RT_SMALL_NODES)))) return fail(RT_ERR_INVALID, "a leaf holds more primitives than a link word can name");
        continue;
      }
      if (i + 1 >= nn || n.offset != i + 1 + size[i + 1] || n.offset >= nn) return fail(RT_ERR_INVALID, "nodes are not a pre-order tree");
      size[i] = 1u + size[i + 1] + size[n.offset];
    }
    if (size[0] != nn) return fail(RT_ERR_INVALID, "nodes are not one tree");
    for (uint64_t t = 0; t < desc->n_tris; ++t)
      if (desc->tri_meta[t].flags & RT_PRIM_SPHERE) {
        uint32_t k; memcpy(&k, desc->tri_p + 9 * t + 6, 4);
        if (!desc->spheres || k >= desc->n_spheres) return fail(RT_ERR_INVALID, "sphere index out of range");
      }
  }
  RtLinkTables lt;
  rt_build_link_tables(desc, mid != 0, stats != nullptr, lt);
  memcpy(link_kept, lt.link_kept.data(), words * 4); memcpy(link_full, lt.link_full.data(), words * 4);
  if (stats) for (int w = 0; w < 9; ++w) { stats[w] = (double)lt.n_rays[w]; stats[9 + w] = lt.tests_all[w]; stats[18 + w] = lt.tests_kept[w]; }
  return RT_OK;
}
extern "C" int rt_scene_query(rt_scene* s, int32_t what) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  if (what == RT_QUERY_LDS_RESIDENT) return s->small ? 1 : 0;
  if (what == RT_QUERY_LDS_NODES_TESTED) return (s->small || s->mid) ? (int)s->lds_nodes_tested : 0;
  if (what == RT_QUERY_LDS_OCCLUSION) return s->mid ? 1 : 0;
  return fail(RT_ERR_INVALID, "unknown rt_scene_query item");
}

// ---------------------------------------------------------------------------------------------- light distribution
// PathIntegrator::preprocess (rc/integrator/path.rs:86-94) + SpatialLightDistribution::new (rc/lightdistrib.rs:67-99)
static int build_light_distribution(rt_scene* s, int strategy, hipStream_t stream, bool all_voxels = false) {
  DScene& d = s->d;
  // PathIntegrator::preprocess runs once per render in the reference (renderer.rs:30) - on an immutable scene it builds the same tables every
  // time, so they are kept (a table built for every voxel serves a frame as well: the frame only reads voxels that hold a surface point)
  if (s->ld_strategy_built == strategy && (s->ld_all_voxels || !all_voxels)) return RT_OK;
  s->ld_strategy_built = -1;  // a rebuild that fails half way leaves no table that a later frame could take for valid
  const int nl = s->n_lights;
  size_t ld_rows_built = 1;  // rows of the tables: one (uniform) or one per built voxel
  const bool uniform = strategy == 1 || nl == 1 || nl == 0;
  // guide tables for the rows' CDF searches (DScene::ld_guide) where a search is long enough to gain from one: >= 64 lights, <= 65534 (u16 entries)
  int glog = -1;
  if (nl >= 64 && nl <= 65534) { const long q = guide_quarters(); glog = 0; while (glog < 16 && (long)(2 << glog) * q <= 4l * nl) ++glog; }
  d.ld_glog = glog; d.ld_guide = nullptr;
  if (uniform) {
    const int n = nl > 0 ? nl : 1;
    std::vector<float> func(n, 1.0f), cdf(n + 1, 0.0f);  // Distribution1D::new over [1.0; n]
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
    float func_int = cdf[n];
    if (func_int == 0.0f) for (int i = 1; i < n + 1; ++i) cdf[i] = (float)i / (float)n;
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= func_int;
    int rc;
    if ((rc = upload(s->ld_func, func.data(), func.size() * 4)) != RT_OK) return rc;
    if ((rc = upload(s->ld_cdf, cdf.data(), cdf.size() * 4)) != RT_OK) return rc;
    if ((rc = upload(s->ld_int, &func_int, 4)) != RT_OK) return rc;
    if (glog >= 0) {
      const int G = 1 << glog; std::vector<unsigned short> g((size_t)G + 1); int i = 0;
      for (int k = 0; k <= G; ++k) { const float x = (float)k / (float)G; while (i <= n && cdf[i] <= x) ++i; g[k] = (unsigned short)i; }
      if ((rc = upload(s->ld_guide, g.data(), g.size() * 2)) != RT_OK) return rc;
    }
    d.ld_uniform = 1; d.nvox[0] = d.nvox[1] = d.nvox[2] = 1;
  } else {
    float diag[3] = {d.wb_max.x - d.wb_min.x, d.wb_max.y - d.wb_min.y, d.wb_max.z - d.wb_min.z};
    int ext = diag[0] > diag[1] ? (diag[0] > diag[2] ? 0 : 2) : (diag[1] > diag[2] ? 1 : 2);
    float b_max = diag[ext];
    size_t total = 1;
    for (int i = 0; i < 3; ++i) {
      float r = roundf(diag[i] / b_max * 64.0f);
      unsigned v = (r != r || r <= 0.0f) ? 0u : (unsigned)r;
      d.nvox[i] = (int)(v > 1u ? v : 1u);
      total *= (size_t)d.nvox[i];
    }
    d.ld_uniform = 0;
    // only voxels that can hold a surface point are built (k_lightdist_mark) and only they get table rows; the table API asks for all
    HIP_TRY(s->ld_mark.ensure(total)); HIP_TRY(s->ld_list.ensure((total + 1) * 4)); HIP_TRY(s->ld_slot.ensure(total * 4));
    unsigned* n_list = s->ld_list.as<unsigned>(); unsigned* list = n_list + 1;
    HIP_TRY(hipMemsetAsync(s->ld_slot.p, 0xff, total * 4, stream));  // -1 = not built
    if (all_voxels) hipLaunchKernelGGL(k_lightdist_iota, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (unsigned)total, list, n_list);
    else {
      HIP_TRY(hipMemsetAsync(s->ld_mark.p, 0, total, stream)); HIP_TRY(hipMemsetAsync(n_list, 0, 4, stream));
      hipLaunchKernelGGL(k_lightdist_mark, dim3((s->n_tris + 255u) / 256u), dim3(256), 0, stream, d, s->ld_mark.as<unsigned char>());
      hipLaunchKernelGGL(k_lightdist_compact, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, s->ld_mark.as<unsigned char>(), (unsigned)total, 0, list, n_list);
    }
    unsigned n_built = 0;  // once per scene and strategy (the tables are kept): a host round trip here is off every frame's path
    HIP_TRY(hipMemcpyAsync(&n_built, n_list, 4, hipMemcpyDeviceToHost, stream)); HIP_TRY(hipStreamSynchronize(stream));
    const size_t rows = std::max<size_t>(n_built, 1);
    ld_rows_built = rows;
    if (s->ld_func.ensure(rows * nl * 4) != hipSuccess || s->ld_cdf.ensure(rows * (nl + 1) * 4) != hipSuccess || s->ld_int.ensure(rows * 4) != hipSuccess)
      return fail(RT_ERR_OOM, "light distribution tables do not fit (built voxels x lights)");
    if (glog >= 0 && s->ld_guide.ensure(rows * (((size_t)1 << glog) + 1) * 2) != hipSuccess) return fail(RT_ERR_OOM, "light distribution guide tables do not fit");
    d.ld_func = s->ld_func.as<float>(); d.ld_cdf = s->ld_cdf.as<float>(); d.ld_int = s->ld_int.as<float>(); d.ld_slot = s->ld_slot.as<int>();
    unsigned lights_pad = 1; while (lights_pad < (unsigned)nl) lights_pad <<= 1;
    const unsigned long long contrib_blocks = lights_pad <= 128u ? (rows + 128u / lights_pad - 1) / (128u / lights_pad) : (unsigned long long)rows * (unsigned)((nl + 127) / 128);
    if (contrib_blocks > 0x7fffffffull) return fail(RT_ERR_INVALID, "light distribution grid too large");
    if (s->has_spheres) hipLaunchKernelGGL(k_lightdist_contrib<true>, dim3((unsigned)contrib_blocks), dim3(128), 0, stream, d, list, n_list, lights_pad, (unsigned)((nl + 127) / 128), s->ld_func.as<float>());
    else hipLaunchKernelGGL(k_lightdist_contrib<false>, dim3((unsigned)contrib_blocks), dim3(128), 0, stream, d, list, n_list, lights_pad, (unsigned)((nl + 127) / 128), s->ld_func.as<float>());
    hipLaunchKernelGGL(k_lightdist_finish, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, stream, d, list, n_list, s->ld_func.as<float>(), s->ld_cdf.as<float>(), s->ld_int.as<float>(), s->ld_slot.as<int>(),
                       glog >= 0 ? s->ld_guide.as<unsigned short>() : nullptr, glog);
    HIP_TRY(hipGetLastError());
  }
  d.ld_func = s->ld_func.as<float>(); d.ld_cdf = s->ld_cdf.as<float>(); d.ld_int = s->ld_int.as<float>(); d.ld_slot = s->ld_slot.as<int>();
  d.ld_guide = glog >= 0 ? s->ld_guide.as<unsigned short>() : nullptr;
  d.ld_rows8 = nullptr; d.ld_dense8 = nullptr;
  if (nl >= 1 && nl <= 3 && !(env_is("RTX_LD_ROWS8", '0'))) {  // (measurement knob: 0 = the three separate tables)
    const size_t n_rows = ld_rows_built;
    HIP_TRY(s->ld_rows8.ensure(n_rows * 32));
    hipLaunchKernelGGL(k_lightdist_rows8, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, s->ld_func.as<float>(), s->ld_cdf.as<float>(), s->ld_int.as<float>(), (int)nl, (unsigned)n_rows, s->ld_rows8.as<float4>());
    HIP_TRY(hipGetLastError());
    d.ld_rows8 = s->ld_rows8.as<float4>();
    const unsigned long long n_vox = (unsigned long long)d.nvox[0] * d.nvox[1] * d.nvox[2];
    const int dense_mb = env_int("RTX_LD_DENSE_MB", 16);  // (measurement knob, read per build: 0 = never)
    if (!uniform && n_vox * 32ull <= (unsigned long long)dense_mb << 20) {
      HIP_TRY(s->ld_dense8.ensure((size_t)n_vox * 32));
      hipLaunchKernelGGL(k_lightdist_dense8, dim3((unsigned)((n_vox + 255) / 256)), dim3(256), 0, stream, s->ld_rows8.as<float4>(), s->ld_slot.as<int>(), n_vox, s->ld_dense8.as<float4>());
      HIP_TRY(hipGetLastError());
      d.ld_dense8 = s->ld_dense8.as<float4>();
    }
  }
  s->ld_strategy_built = strategy; s->ld_all_voxels = all_voxels;
  HIP_TRY(hipMemcpyAsync(s->self.p, &s->d, sizeof(DScene), hipMemcpyHostToDevice, stream));
  return RT_OK;
}

extern "C" int rt_light_distribution(rt_scene* s, int32_t n_voxels[3], float* func, float* cdf, float* func_int) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  int rc = build_light_distribution(s, 0, nullptr, true);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (s->d.ld_uniform) { n_voxels[0] = n_voxels[1] = n_voxels[2] = 0; return RT_OK; }
  n_voxels[0] = s->d.nvox[0]; n_voxels[1] = s->d.nvox[1]; n_voxels[2] = s->d.nvox[2];
  size_t total = (size_t)s->d.nvox[0] * s->d.nvox[1] * s->d.nvox[2];
  if (func) HIP_TRY(hipMemcpy(func, s->ld_func.p, total * s->n_lights * 4, hipMemcpyDeviceToHost));
  if (cdf) HIP_TRY(hipMemcpy(cdf, s->ld_cdf.p, total * (s->n_lights + 1) * 4, hipMemcpyDeviceToHost));
  if (func_int) HIP_TRY(hipMemcpy(func_int, s->ld_int.p, total * 4, hipMemcpyDeviceToHost));
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------- trace launches
// LDS a workgroup of the trace kernels declares, and the persistent grid that fills every CU at that residency
template <bool ANY, bool SMALL, int BLOCK, int DEPTH>
static unsigned trace_grid(const rt_scene* s, bool links = false) {  // links: an LDS-resident scene's rays that count no visits (link rows instead of a stack, k_trace)
  unsigned lds = (unsigned)(DEPTH * BLOCK * (SMALL ? 2 : 4) + (SMALL ? (8 * RT_SMALL_NODES + 10 * RT_SMALL_TRIS) * 4 : 32));
  unsigned waves = 8;
  if (SMALL && links) { lds = (unsigned)((8 * RT_SMALL_NODES + 10 * RT_SMALL_TRIS) * 4 + (ANY ? 1 : 8) * RT_SMALL_NODES * 4 + 32); waves = ANY ? RT_LDS_ANY_WAVES : RT_LDS_TRACE_WAVES; }
  unsigned per_cu = (160u * 1024u) / lds; if (per_cu * BLOCK > waves * 256u) per_cu = waves * 256u / BLOCK; if (per_cu < 1) per_cu = 1;
  return (unsigned)s->n_cu * per_cu;
}
// HBM half of the child-pair / four-wide kernels' traversal stack: [depth][lane of the grid]. Sized once per scene for the largest grid any
// launch variant uses (rt_scene_create), so that no launch can fail on an allocation and leave stale hit records behind.
#define RT_TOP_BLOCK 512
static unsigned top_grid(const rt_scene* s, bool any = false) {
  (void)any;
  const unsigned fit = (160u * 1024u) / (16384u + (unsigned)RT_TOP_LDS_DEPTH * RT_TOP_BLOCK * 4u);
  return (unsigned)s->n_cu * std::max(1u, std::min(3u, fit));
}  // 48 KB of LDS per workgroup: three per CU, six waves per SIMD
static size_t tmin_stack_bytes(const rt_scene* s) {
  if (s->small || !s->use_pairs) return 0;
  const size_t a = s->stack_depth <= 32 ? (size_t)trace_grid<false, false, 128, 32>(s) * 128 * 32 * 4 : (size_t)trace_grid<false, false, 128, 64>(s) * 128 * 64 * 4;
  const size_t b = s->use_top ? (size_t)std::max(top_grid(s), top_grid(s, true)) * RT_TOP_BLOCK * (size_t)s->stack_depth * 4 : 0;
  return std::max(a, b);
}
static size_t deep_stack_bytes(const rt_scene* s) {
  if (!s->use_top) return 0;
  return (size_t)std::max(top_grid(s), top_grid(s, true)) * RT_TOP_BLOCK * (size_t)std::max(1, s->stack_depth - RT_TOP_LDS_DEPTH) * 4;
}
// the two knobs of the persistent traversal loops in one launch argument: lanes that must be idle before a wave refills (bits 0-7) and lanes that must wait at
// a leaf before the leaf phase runs (bits 8-15; RT_LEAF_MIN, see leaf_phase_now). On an instanced scene a
// leaf is a whole nested walk and gating pays even more (10 000 placements, closest hit: 310 / 227 / 166 / 140 / 125 / 122 ms at 1 / 8 / 20 / 32 / 48 / 64).
// does the scene trace through k_trace_inst (object instances over plain triangles, objects with pair records)?
static bool inst_loop_kernel(const rt_scene* s) { return s->has_instances && !s->has_masks && !s->has_spheres && s->use_pairs && s->d.obj_pairs; }
static unsigned trace_knobs(const rt_scene* s, bool any) {
  const unsigned refill = (unsigned)RT_REFILL_MIN;
  // (an instance leaf of the nested form is a whole walk: the holders wait for every walker. In k_trace_inst a leaf is a leaf again: 20 / 24 / 32 / 40 / 48 / 64 lanes
  // gave 780 / 779 / 781 / 764 / 739 / 684 Msamples/s on 10 000 placements of a 1280-triangle object)
  unsigned leaf = s->has_instances ? (inst_loop_kernel(s) ? 32u : 64u) : (unsigned)RT_LEAF_MIN;
  // Occlusion rays under an environment light are unbounded (t_max = inf) and long: with the order of RT_ANY_ORDER their leaf phases pay at 48 waiting lanes
  // (S4 shadow / environment MIS rays 360 / 686 ms at 20, 335 / 637 at 28 - 32, 321 / 606 at 48), where the bounded shadow rays of area-light scenes lose
  // (S2 31 -> 45 ms, S3 101 -> 112)
  if (any && s->d.n_infinite > 0 && !s->has_instances) leaf = 48u;
  return refill | (leaf << 8);
}
template <bool ANY, bool COUNT, int BLOCK, int DEPTH>
static void launch_trace_hbm(rt_scene* s, const TraceIO& io, const unsigned* queue, const unsigned* count_ptr, unsigned shard_cap, unsigned count_static,
                             unsigned long long* stats, int st_rays, int st_nodes, int st_tris, hipStream_t stream) {
  // HBM scenes of plain triangles: occlusion rays four-wide (k_trace_quad) where the records exist, otherwise child-pair steps with the top of the tree in LDS (k_trace_top)
  // or without (k_trace_pair); frames that count visits and trees without pair records walk one node per step (k_trace_big)
  const unsigned grid = trace_grid<ANY, false, BLOCK, DEPTH>(s);
  const unsigned refill_min = trace_knobs(s, ANY);
  if constexpr (ANY && !COUNT) {
    if (s->use_quads && s->quad_stack_depth <= DEPTH) {
      hipLaunchKernelGGL((k_trace_quad<BLOCK, DEPTH>), dim3(grid), dim3(BLOCK), 0, stream, s->d, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, s->tmin_stack.as<float>(), refill_min);
      return;
    }
  }
  if constexpr (!COUNT) {
    if (s->use_pairs && s->use_top && (ANY || s->top_for_closest)) {
      hipLaunchKernelGGL((k_trace_top<ANY, RT_TOP_BLOCK>), dim3(top_grid(s, ANY)), dim3(RT_TOP_BLOCK), 0, stream, s->d, io, queue, count_ptr, shard_cap, count_static, stats, st_rays,
                         s->tmin_stack.as<float>(), s->deep_stack.as<unsigned>(), refill_min);
      return;
    }
    if (s->use_pairs) {
      hipLaunchKernelGGL((k_trace_pair<ANY, BLOCK, DEPTH>), dim3(grid), dim3(BLOCK), 0, stream, s->d, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, s->tmin_stack.as<float>(), refill_min);
      return;
    }
  }
  hipLaunchKernelGGL((k_trace_big<ANY, COUNT, BLOCK, DEPTH>), dim3(grid), dim3(BLOCK), 0, stream, s->d, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, st_nodes, st_tris);
}
// does the scene's tree live in LDS for the rays of a frame (k_trace: the 256-node kernels, or one 1024-lane workgroup per CU for a mid-size scene)? RTX_MID_CLOSEST=0: the A/B
// control of the mid-size closest-hit walk (those rays through the HBM kernels)
static bool mid_closest_on() { static const bool on = !(env_is("RTX_MID_CLOSEST", '0')); return on; }
template <bool ANY, bool COUNT>
static void launch_trace_c(rt_scene* s, const TraceIO& io, const unsigned* queue, const unsigned* count_ptr, unsigned shard_cap, unsigned count_static,
                           unsigned long long* stats, int st_rays, int st_nodes, int st_tris, hipStream_t stream) {
#define RT_ARGS s, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, st_nodes, st_tris, stream
#define RT_KARGS s->d, io, queue, count_ptr, shard_cap, count_static, stats, st_rays
  if (s->general_prims) {  // quadrics, masked triangles, object instances: the GENERAL instantiations (plain leaves of such a scene still run the bare triangle loop)
    // (objects that hold quadrics or masked triangles - obj_general - are walked by the one-node-per-step kernel below: the only one whose object walk takes them)
    if (!COUNT && !s->obj_general) {
      const unsigned refill_min = trace_knobs(s, ANY);
      const bool all = s->has_masks;
#define RT_GEN_LAUNCH(KERNEL_ALL, KERNEL_NOMASK, GRID, BLK, ...) do { if (all) hipLaunchKernelGGL(KERNEL_ALL, dim3(GRID), dim3(BLK), 0, stream, __VA_ARGS__); \
                                                                      else hipLaunchKernelGGL(KERNEL_NOMASK, dim3(GRID), dim3(BLK), 0, stream, __VA_ARGS__); } while (0)
      // ... and neither masks nor quadrics (instances over plain triangles): the pair / four-wide kernels without the quadric test as well
#define RT_GEN_LAUNCH3(KERNEL_ALL, KERNEL_NOMASK, KERNEL_INST, GRID, BLK, ...) do { if (!all && !s->has_spheres) hipLaunchKernelGGL(KERNEL_INST, dim3(GRID), dim3(BLK), 0, stream, __VA_ARGS__); \
                                                                                   else RT_GEN_LAUNCH(KERNEL_ALL, KERNEL_NOMASK, GRID, BLK, __VA_ARGS__); } while (0)
      if (s->small) {  // the link walks: no stack, one instantiation whatever the tree's depth; grid = what fits a CU (LDS: scene + link rows; registers: RT_GEN_MIN_WAVES)
        const unsigned lds = (unsigned)((8 * RT_SMALL_NODES + 10 * RT_SMALL_TRIS) * 4 + ((ANY ? 1 : 8) * RT_SMALL_NODES + 8) * 4 + 64);
        unsigned per_cu = (160u * 1024u) / lds; const unsigned by_regs = all ? 2u : 4u; if (per_cu > by_regs) per_cu = by_regs;
        RT_GEN_LAUNCH((k_trace<ANY, false, 256, 16, RT_GEN_ALL>), (k_trace<ANY, false, 256, 16, RT_GEN_NO_MASKS>), ((unsigned)s->n_cu * per_cu), 256, RT_KARGS, st_nodes, st_tris);
        return;
      }
      if (inst_loop_kernel(s)) {  // object instances over plain triangles: the two-level walk as one loop (k_trace_inst)
        const unsigned depth = s->stack_depth <= 32 ? 32u : 64u;
        const unsigned lds = depth * 128u * 4u + 13u * 128u * 4u;
        const unsigned grid_i = (unsigned)s->n_cu * std::max(1u, std::min(16u, (160u * 1024u) / lds));
        const unsigned grid_cap = depth == 32u ? trace_grid<ANY, false, 128, 32>(s) : trace_grid<ANY, false, 128, 64>(s);  // (the deferred-tmin array is sized for this grid)
        const unsigned g = std::min(grid_i, grid_cap);
        if (depth == 32u) hipLaunchKernelGGL((k_trace_inst<ANY, 128, 32>), dim3(g), dim3(128), 0, stream, RT_KARGS, s->tmin_stack.as<float>(), refill_min);
        else hipLaunchKernelGGL((k_trace_inst<ANY, 128, 64>), dim3(g), dim3(128), 0, stream, RT_KARGS, s->tmin_stack.as<float>(), refill_min);
        return;
      }
      if constexpr (ANY) {
        if (s->use_quads && s->quad_stack_depth <= 32) {
          RT_GEN_LAUNCH3((k_trace_quad<128, 32, RT_GEN_ALL>), (k_trace_quad<128, 32, RT_GEN_NO_MASKS>), (k_trace_quad<128, 32, RT_GEN_INSTANCES_ONLY>), (trace_grid<ANY, false, 128, 32>(s)), 128, RT_KARGS, s->tmin_stack.as<float>(), refill_min);
          return;
        }
      }
      if (s->use_pairs && s->use_top && (ANY || s->top_for_closest)) {
        RT_GEN_LAUNCH((k_trace_top<ANY, RT_TOP_BLOCK, RT_GEN_ALL>), (k_trace_top<ANY, RT_TOP_BLOCK, RT_GEN_NO_MASKS>), (top_grid(s, ANY)), RT_TOP_BLOCK, RT_KARGS, s->tmin_stack.as<float>(), s->deep_stack.as<unsigned>(), refill_min);
        return;
      }
      if (s->use_pairs) {
        if (s->stack_depth <= 32) RT_GEN_LAUNCH3((k_trace_pair<ANY, 128, 32, RT_GEN_ALL>), (k_trace_pair<ANY, 128, 32, RT_GEN_NO_MASKS>), (k_trace_pair<ANY, 128, 32, RT_GEN_INSTANCES_ONLY>), (trace_grid<ANY, false, 128, 32>(s)), 128, RT_KARGS, s->tmin_stack.as<float>(), refill_min);
        else RT_GEN_LAUNCH3((k_trace_pair<ANY, 128, 64, RT_GEN_ALL>), (k_trace_pair<ANY, 128, 64, RT_GEN_NO_MASKS>), (k_trace_pair<ANY, 128, 64, RT_GEN_INSTANCES_ONLY>), (trace_grid<ANY, false, 128, 64>(s)), 128, RT_KARGS, s->tmin_stack.as<float>(), refill_min);
        return;
      }
#undef RT_GEN_LAUNCH3
#undef RT_GEN_LAUNCH
    }
    if (s->stack_depth > 64) hipLaunchKernelGGL((k_trace_big<ANY, COUNT, 64, 128, true>), dim3(trace_grid<ANY, false, 64, 128>(s)), dim3(64), 0, stream, RT_KARGS, st_nodes, st_tris);
    else if (s->stack_depth <= 32) hipLaunchKernelGGL((k_trace_big<ANY, COUNT, 128, 32, true>), dim3(trace_grid<ANY, false, 128, 32>(s)), dim3(128), 0, stream, RT_KARGS, st_nodes, st_tris);
    else hipLaunchKernelGGL((k_trace_big<ANY, COUNT, 128, 64, true>), dim3(trace_grid<ANY, false, 128, 64>(s)), dim3(128), 0, stream, RT_KARGS, st_nodes, st_tris);
    return;
  }
  if constexpr (!COUNT) {
    if (s->mid && (ANY || mid_closest_on())) {
      hipLaunchKernelGGL((k_trace<ANY, false, 1024, 16, 0, 1>), dim3((unsigned)s->n_cu), dim3(1024), 0, stream, RT_KARGS, st_nodes, st_tris);
      return;
    }
    if (s->small) {  // the link walks: no stack, whatever the tree's depth
      hipLaunchKernelGGL((k_trace<ANY, false, 256, 16>), dim3(trace_grid<ANY, true, 256, 16>(s, true)), dim3(256), 0, stream, RT_KARGS, st_nodes, st_tris);
      return;
    }
  }
  if (s->small) {  // a frame that counts the reference's walk: the stack walk over the LDS copy
    if (s->stack_depth <= 16) hipLaunchKernelGGL((k_trace<ANY, true, 256, 16>), dim3(trace_grid<ANY, true, 256, 16>(s)), dim3(256), 0, stream, RT_KARGS, st_nodes, st_tris);
    else if (s->stack_depth <= 32) hipLaunchKernelGGL((k_trace<ANY, true, 256, 32>), dim3(trace_grid<ANY, true, 256, 32>(s)), dim3(256), 0, stream, RT_KARGS, st_nodes, st_tris);
    else hipLaunchKernelGGL((k_trace<ANY, true, 128, 64>), dim3(trace_grid<ANY, true, 128, 64>(s)), dim3(128), 0, stream, RT_KARGS, st_nodes, st_tris);
  } else {
    if (s->stack_depth <= 32) launch_trace_hbm<ANY, COUNT, 128, 32>(RT_ARGS);
    else launch_trace_hbm<ANY, COUNT, 128, 64>(RT_ARGS);
  }
#undef RT_KARGS
#undef RT_ARGS
}
template <bool ANY>
static void launch_trace(rt_scene* s, bool count, const TraceIO& io, const unsigned* queue, const unsigned* count_ptr, unsigned shard_cap, unsigned count_static,
                         unsigned long long* stats, int st_rays, int st_nodes, int st_tris, hipStream_t stream) {
  if (count) launch_trace_c<ANY, true>(s, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, st_nodes, st_tris, stream);
  else launch_trace_c<ANY, false>(s, io, queue, count_ptr, shard_cap, count_static, stats, st_rays, st_nodes, st_tris, stream);
}
// planar arrays of the batch entry points
static TraceIO trace_io_planar(const float4* ro, const float4* rd, float4* hits, unsigned* occ) {
  TraceIO io{}; io.ray_o = ro; io.ray_d = rd; io.ray_stride = 1; io.hits = hits; io.hit_stride = 1; io.hit_b2 = 0; io.occluded = occ; io.occ_stride = 1;
  io.shadow_masks = 1;  // rt_trace_any is Scene::intersect_p
  return io;
}

static int trace_batch(rt_scene* s, const float* rays, uint64_t n, bool any, float* hits, uint32_t* occluded, uint64_t counters[2]) {
  if (!s || !rays || n == 0 || n > 0x7fffffffull) return fail(RT_ERR_INVALID, "bad trace batch");
  HIP_TRY(hipSetDevice(s->device));
  DevBuf ro, rd, out, st;
  HIP_TRY(ro.ensure(n * 16)); HIP_TRY(rd.ensure(n * 16)); HIP_TRY(out.ensure(n * 16)); HIP_TRY(st.ensure(ST_COUNT * 8));
  std::vector<float> o(n * 4), d(n * 4);
  for (uint64_t i = 0; i < n; ++i) {
    const float* r = rays + 8 * i;
    o[4 * i] = r[0]; o[4 * i + 1] = r[1]; o[4 * i + 2] = r[2]; o[4 * i + 3] = r[3];
    d[4 * i] = r[4]; d[4 * i + 1] = r[5]; d[4 * i + 2] = r[6]; d[4 * i + 3] = 0.0f;
  }
  HIP_TRY(hipMemcpy(ro.p, o.data(), n * 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rd.p, d.data(), n * 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(st.p, 0, ST_COUNT * 8));
  const TraceIO io = trace_io_planar(ro.as<float4>(), rd.as<float4>(), any ? nullptr : out.as<float4>(), any ? out.as<unsigned>() : nullptr);
  if (any) launch_trace<true>(s, counters != nullptr, io, nullptr, nullptr, 0, (unsigned)n, st.as<unsigned long long>(), ST_RAYS_SHADOW, ST_NODES_SHADOW, ST_TRIS_SHADOW, nullptr);
  else launch_trace<false>(s, counters != nullptr, io, nullptr, nullptr, 0, (unsigned)n, st.as<unsigned long long>(), ST_RAYS_CLOSEST, ST_NODES_CLOSEST, ST_TRIS_CLOSEST, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  if (any) HIP_TRY(hipMemcpy(occluded, out.p, n * 4, hipMemcpyDeviceToHost));
  else HIP_TRY(hipMemcpy(hits, out.p, n * 16, hipMemcpyDeviceToHost));
  if (counters) {
    unsigned long long h[ST_COUNT];
    HIP_TRY(hipMemcpy(h, st.p, sizeof(h), hipMemcpyDeviceToHost));
    counters[0] = any ? h[ST_NODES_SHADOW] : h[ST_NODES_CLOSEST];
    counters[1] = any ? h[ST_TRIS_SHADOW] : h[ST_TRIS_CLOSEST];
  }
  return RT_OK;
}
extern "C" int rt_trace_closest(rt_scene* s, const float* rays, uint64_t n, float* hits, uint64_t counters[2]) { return trace_batch(s, rays, n, false, hits, nullptr, counters); }
extern "C" int rt_trace_any(rt_scene* s, const float* rays, uint64_t n, uint32_t* occluded, uint64_t counters[2]) { return trace_batch(s, rays, n, true, nullptr, occluded, counters); }

extern "C" int rt_trace_closest_device(rt_scene* s, const void* d_rays, uint64_t n, void* d_hits, int reps, void* stream_, float* ms_per_launch) {
  if (!s || !d_rays || !d_hits || n == 0 || reps <= 0) return fail(RT_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t stream = (hipStream_t)stream_;
  // d_rays: n x (float4 o|tmax, float4 d) planar: first n float4 origins, then n float4 directions
  const float4* ro = (const float4*)d_rays; const float4* rd = ro + n;
  hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, stream));
  const TraceIO io = trace_io_planar(ro, rd, (float4*)d_hits, nullptr);
  for (int i = 0; i < reps; ++i) launch_trace<false>(s, false, io, nullptr, nullptr, 0, (unsigned)n, nullptr, 0, 0, 0, stream);
  HIP_TRY(hipEventRecord(e1, stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_launch) *ms_per_launch = ms / (float)reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------- sampler tables
static unsigned next_pow2(unsigned v) { unsigned p = 1; while (p < v) p <<= 1; return p; }
// lanes per K0 block: the per-lane permutation (2 B * (spp + 2)) of a block should fill ~32 KB of LDS
static unsigned sampler_lanes_per_block(unsigned spp) {
  unsigned l = (32u * 1024u) / ((spp + 2u) * 2u);
  if (l >= 64u) return 64u;
  if (l >= 32u) return 32u;
  if (l >= 16u) return 16u;
  if (l >= 8u) return 8u;
  return 4u;
}

// PCG32 jump-ahead constants: state_n = A^n * state_0 + S_n * inc with S_n = 1 + A + ... + A^(n-1) (mod 2^64)
static void pcg_advance_constants(unsigned long long n, unsigned long long& mul, unsigned long long& sum) {
  unsigned long long cur_mul = 0x5851f42d4c957f2dULL, cur_sum = 1ull;
  mul = 1ull; sum = 0ull;
  while (n) {
    if (n & 1ull) { sum = sum * cur_mul + cur_sum; mul *= cur_mul; }
    cur_sum = (cur_mul + 1ull) * cur_sum; cur_mul *= cur_mul;
    n >>= 1;
  }
}
// Launch plan of K0 for (spp, dims): stream segments for k_sampler_draws and the reciprocal table.
static int sampler_plan_prepare(SamplerPlan& pl, unsigned spp_, unsigned dims_) {
  unsigned &spp = pl.spp, &dims = pl.dims, &seg_len = pl.seg_len, &n_segs = pl.n_segs;
  DevBuf &segs = pl.segs, &magic = pl.magic, &dirty = pl.dirty;
  {
    if (spp == spp_ && dims == dims_ && segs.p) return RT_OK;
    seg_len = spp_ < 256u ? spp_ : 256u;
    const unsigned per_half = spp_ / seg_len;
    std::vector<SamplerSeg> h;
    unsigned long long pos = 0;
    for (unsigned t = 0; t < 2u * dims_; ++t) {
      const unsigned n_scr = t < dims_ ? 1u : 2u;
      for (unsigned half = 0; half < 2u; ++half)
        for (unsigned g = 0; g < per_half; ++g) {
          SamplerSeg sg{};
          const bool first = half == 0u && g == 0u;
          const unsigned long long start = first ? pos : pos + n_scr + (unsigned long long)half * spp_ + (unsigned long long)g * seg_len;
          pcg_advance_constants(start, sg.mul, sg.sum);
          sg.table = t; sg.half = half; sg.i0 = g * seg_len; sg.n_scr = first ? n_scr : 0u;
          h.push_back(sg);
        }
      pos += n_scr + 2ull * spp_;
    }
    std::vector<unsigned> mg(spp_ + 1, 0u);
    for (unsigned b = 2; b <= spp_; ++b) mg[b] = (unsigned)((1ull << 32) / b);
    int rc;
    if ((rc = upload(segs, h.data(), h.size() * sizeof(SamplerSeg))) != RT_OK) return rc;
    if ((rc = upload(magic, mg.data(), mg.size() * 4)) != RT_OK) return rc;
    HIP_TRY(dirty.ensure((3 + RT_DIRTY_CAP) * 4 + 128));
    HIP_TRY(hipMemset(dirty.p, 0, (3 + RT_DIRTY_CAP) * 4 + 128));
    spp = spp_; dims = dims_; n_segs = (unsigned)h.size();
    return RT_OK;
  }
}
// lanes per K0b block: 16384 / spp lanes hold 32 KB of permutations (one wave at most, four lanes at least)
static unsigned shuffle_lanes_per_block(unsigned spp) { unsigned l = 65536u / spp; return l > 64u ? 64u : (l < 4u ? 4u : l); }
// The tables of a batch in the ORDER A FRAME NEEDS THEM (round 4). ZeroTwoSequence hands out dimension after dimension: the camera sample takes 2-D tables 0 and 1
// (film, lens) and 1-D table 0 (time - which this camera never reads); bounce 0 takes 1-D table 1 (light pick) and the remaining 2-D tables (light point,
// scattering direction); bounces 1, 2, ... the remaining 1-D tables; everything after comes from the per-sample RNG. One PCG32 stream runs through all tables
// of a pixel, so the DRAWS of all tables are made first (k_sampler_draws finds the retries that shift later tables; 3.8 ms of a 28.6 ms batch) - but the
// shuffles, the expensive part, are independent per table: group 0 = what k_raygen reads, group 1 = what bounce 0's shade reads, group 2 = the rest; an event after
// each group lets the frame's first batch start its camera rays after 11 ms instead of 28.6, with the other shuffles under its first kernels. 1-D table 0 is not
// shuffled at all inside a frame (nothing reads it); rt_sampler_tables builds every table.
struct TableGroups { unsigned long long packed[3]; unsigned n[3]; int n_groups; };  // a group's table ids, 4 bits each: 2 * dims <= 16 tables fill 64 bits (32 bits held 8)
static TableGroups table_groups_all(unsigned dims) {
  TableGroups g{}; g.n_groups = 1;
  for (unsigned t = 0; t < 2u * dims; ++t) { g.packed[0] |= (unsigned long long)t << (4u * g.n[0]); g.n[0] += 1; }
  return g;
}
static TableGroups table_groups_frame(unsigned dims, bool pinhole) {
  TableGroups g{}; g.n_groups = 3;
  auto add = [&](int k, unsigned t) { g.packed[k] |= (unsigned long long)t << (4u * g.n[k]); g.n[k] += 1; };
  add(0, dims);                                                      // 2-D table 0: the camera sample's film position
  // 2-D table 1 is the lens sample: get_camera_sample consumes it (camera/mod.rs), a pinhole camera never looks at it (perspective.rs: lens_radius > 0) - like 1-D table 0,
  // the time sample nothing reads, its shuffle is not replayed inside a frame (round 6: a seventh of K0's replays; the draws still advance the pixel's RNG stream)
  if (!pinhole) add(0, dims + 1u);
  if (dims > 1u) add(1, 1u);                                         // 1-D table 1: bounce 0's light pick
  for (unsigned d = 2; d < dims; ++d) add(1, dims + d);              // 2-D tables 2 ...: bounce 0's light point and scattering direction
  for (unsigned d = 2; d < dims; ++d) add(2, d);                     // 1-D tables 2 ...: the light picks of bounces 1, 2, ...
  return g;
}
static int launch_sampler_tables(SamplerPlan& pl, const FrameParams& fp, unsigned n_pixels, unsigned long long explicit_pixel0, int use_explicit,
                                 unsigned* scrambles, unsigned short* perms, hipStream_t stream, const TableGroups& groups, hipEvent_t* group_done) {
  const unsigned spp = pl.spp, dims = pl.dims;
  const unsigned lpb = shuffle_lanes_per_block(spp);
  const size_t lds = (size_t)lpb * spp * 2;
  HIP_TRY(pl.partners.ensure((size_t)n_pixels * 2u * dims * spp * 2u));
  unsigned short* partners = pl.partners.as<unsigned short>();
  HIP_TRY(hipMemsetAsync(pl.dirty.p, 0, 4, stream));
  // Which kernel replays the shuffles. k_sampler_shuffle_par (one wave per chain, exact; 64 <= spp <= 1024) takes 25.0 ms for a batch of S1's tables (524 288 pixels x 8
  // tables) on its own against the chain kernel's 28.6 (the replay 15.4 against 23.2 ms; the draws kernel pays 4.4 ms for the chain-major partner layout it needs) -
  // but a frame builds all tables except its first batch's UNDER path kernels, where the chain kernel (one latency-bound wave per CU) takes almost nothing from them
  // and the parallel one competes for issue slots and LDS. Measured in round 4, frame times: at 256 / 512 spp the parallel replay loses (S2 167.3 -> 169.5 ms, S3 323.5 ->
  // 324.6); at 1024 spp, with a short leading batch (lead_pixels in rt_render), it wins for EVERY batch (S1 679.5 -> 667.5 ms, S4 5280 -> 5242).
  // DEFAULT therefore: parallel replay at spp == 1024, chain kernel otherwise. RTX_K0_PARALLEL=1 forces the parallel replay (64 <= spp <= 1024), =0 the chain kernel.
  static const bool par_on = env_is("RTX_K0_PARALLEL", '1');
  static const bool par_never = env_is("RTX_K0_PARALLEL", '0');
  // RTX_K0_FORCE_RESORT=1 (test knob, read per call): the parallel replay is handed descending ranks, so every wave takes the branch that re-sorts its groups
  const unsigned force_resort = (env_is("RTX_K0_FORCE_RESORT", '1')) ? 1u : 0u;
  const bool par = (par_on || (!par_never && spp == 1024u)) && spp >= 64u && spp <= 1024u;
  hipLaunchKernelGGL(k_sampler_draws, dim3((n_pixels + 255u) / 256u, pl.n_segs), dim3(256), 0, stream, fp, n_pixels, spp, dims, pl.seg_len, explicit_pixel0, use_explicit,
                     pl.segs.as<SamplerSeg>(), pl.magic.as<unsigned>(), scrambles, partners, pl.dirty.as<unsigned>(), par ? 1 : 0);
  hipLaunchKernelGGL(k_sampler_redo, dim3(RT_DIRTY_CAP / 64u), dim3(64), 0, stream, fp, n_pixels, spp, dims, explicit_pixel0, use_explicit, pl.dirty.as<unsigned>(), pl.magic.as<unsigned>(), scrambles, partners, par ? 1 : 0);
  unsigned* const resorted = pl.dirty.as<unsigned>() + 1 + RT_DIRTY_CAP + 1;  // waves whose groups did not come out sorted (diagnostic; the result is exact either way)
  for (int g = 0; g < groups.n_groups; ++g) {
    const unsigned long long tables = groups.packed[g]; const unsigned nt = groups.n[g];
    if (nt) {
      if (par) {
        const dim3 grid((n_pixels + RT_SHUF_PIX - 1) / RT_SHUF_PIX, nt);
        switch (spp) {
          case 64: hipLaunchKernelGGL(k_sampler_shuffle_par<1>, grid, dim3(256), 0, stream, n_pixels, partners, perms, resorted, tables, force_resort); break;
          case 128: hipLaunchKernelGGL(k_sampler_shuffle_par<2>, grid, dim3(256), 0, stream, n_pixels, partners, perms, resorted, tables, force_resort); break;
          case 256: hipLaunchKernelGGL(k_sampler_shuffle_par<4>, grid, dim3(256), 0, stream, n_pixels, partners, perms, resorted, tables, force_resort); break;
          case 512: hipLaunchKernelGGL(k_sampler_shuffle_par<8>, grid, dim3(256), 0, stream, n_pixels, partners, perms, resorted, tables, force_resort); break;
          default: hipLaunchKernelGGL(k_sampler_shuffle_par<16>, grid, dim3(256), 0, stream, n_pixels, partners, perms, resorted, tables, force_resort); break;
        }
      } else hipLaunchKernelGGL(k_sampler_shuffle, dim3((n_pixels + lpb - 1) / lpb, nt), dim3(lpb), lds, stream, n_pixels, spp, partners, perms, tables);
    }
    if (group_done) (void)hipEventRecord(group_done[g], stream);
  }
  return RT_OK;
}
static int sampler_set_lds_limits(unsigned spp) {
  const size_t lds = (size_t)sampler_lanes_per_block(spp) * (spp + 2) * 2;
  HIP_TRY(hipFuncSetAttribute((const void*)k_sampler_tables, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIP_TRY(hipFuncSetAttribute((const void*)k_sampler_shuffle, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)shuffle_lanes_per_block(spp) * spp * 2)));
  return RT_OK;
}

static int sampler_tables_host(int32_t spp_, int32_t dims, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms, bool plain);
extern "C" int rt_sampler_tables(int32_t spp, int32_t dims, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms) {
  return sampler_tables_host(spp, dims, pixel0, n_pixels, scrambles, perms, false);
}
extern "C" int rt_sampler_tables_plain(int32_t spp, int32_t dims, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms) {
  return sampler_tables_host(spp, dims, pixel0, n_pixels, scrambles, perms, true);
}
static int sampler_tables_host(int32_t spp_, int32_t dims, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms, bool plain) {
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  if (spp_ <= 0 || dims <= 0 || dims > 8 || n_pixels == 0) return fail(RT_ERR_INVALID, "bad sampler arguments");
  const unsigned spp = next_pow2((unsigned)spp_);
  if (spp > 16384) return fail(RT_ERR_INVALID, "spp > 16384 unsupported");
  DevBuf sc, pm;
  HIP_TRY(sc.ensure(n_pixels * 3 * dims * 4)); HIP_TRY(pm.ensure(n_pixels * 2 * dims * spp * 2));
  FrameParams fp{};
  SamplerPlan plan; int rc;
  if ((rc = sampler_plan_prepare(plan, spp, (unsigned)dims)) != RT_OK) return rc;
  if ((rc = sampler_set_lds_limits(spp)) != RT_OK) return rc;
  if (plain) {
    const unsigned lpb = sampler_lanes_per_block(spp);
    hipLaunchKernelGGL(k_sampler_tables, dim3((unsigned)((n_pixels + lpb - 1) / lpb)), dim3(lpb), (size_t)lpb * (spp + 2) * 2, nullptr, fp, (unsigned)n_pixels, spp, (unsigned)dims,
                       (unsigned long long)pixel0, 1, sc.as<unsigned>(), pm.as<unsigned short>());
  } else if ((rc = launch_sampler_tables(plan, fp, (unsigned)n_pixels, (unsigned long long)pixel0, 1, sc.as<unsigned>(), pm.as<unsigned short>(), nullptr, table_groups_all((unsigned)dims), nullptr)) != RT_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  { unsigned ovf[2] = {0, 0}; HIP_TRY(hipMemcpy(ovf, plan.dirty.as<unsigned>() + 1 + RT_DIRTY_CAP, 8, hipMemcpyDeviceToHost)); if (ovf[0]) return fail(RT_ERR_INVALID, "sampler retry list overflow");
    if (getenv("RTX_K0_REPORT")) { fprintf(stderr, "k_sampler_shuffle_par: %u wave(s) re-sorted their groups\n", ovf[1]);
      unsigned long long st[9]; (void)hipMemcpy(st, (char*)plan.dirty.p + (3 + RT_DIRTY_CAP) * 4 + 4, sizeof(st), hipMemcpyDeviceToHost);
      fprintf(stderr, "  stamps:"); for (int k = 0; k < 9; ++k) fprintf(stderr, " %llu", st[k]); fprintf(stderr, "\n"); } }
  // device layout is pixel-minor ([k][pixel], [table][sample][pixel]); the ABI returns pixel-major
  std::vector<uint32_t> hs(n_pixels * 3 * dims); std::vector<uint16_t> hp(n_pixels * 2 * dims * spp);
  HIP_TRY(hipMemcpy(hs.data(), sc.p, hs.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hp.data(), pm.p, hp.size() * 2, hipMemcpyDeviceToHost));
  for (uint64_t p = 0; p < n_pixels; ++p) {
    for (uint64_t k = 0; k < 3ull * dims; ++k) scrambles[p * 3 * dims + k] = hs[k * n_pixels + p];
    for (uint64_t t = 0; t < 2ull * dims; ++t)
      for (uint64_t i = 0; i < spp; ++i) perms[(p * 2 * dims + t) * spp + i] = hp[(t * spp + i) * n_pixels + p];
  }
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------- render
struct KTimer {  // HIP-event kernel timing on the render stream; events come from a per-scene pool
  bool on; hipStream_t st; std::vector<hipEvent_t>* pool; size_t used = 0;
  std::vector<double*> dst;
  hipEvent_t get() {
    if (used == pool->size()) { hipEvent_t e; (void)hipEventCreate(&e); pool->push_back(e); }
    return (*pool)[used++];
  }
  void begin(double* where, hipStream_t on_stream = nullptr) { if (!on) return; (void)hipEventRecord(get(), on_stream ? on_stream : st); dst.push_back(where); }
  void end(hipStream_t on_stream = nullptr) { if (!on) return; (void)hipEventRecord(get(), on_stream ? on_stream : st); }
  void collect() {
    for (size_t i = 0; i < dst.size(); ++i) {
      float ms = 0; (void)hipEventSynchronize((*pool)[2 * i + 1]); (void)hipEventElapsedTime(&ms, (*pool)[2 * i], (*pool)[2 * i + 1]);
      *dst[i] += ms;
    }
    dst.clear(); used = 0;
  }
};

// one shade launch of front-end MODE (rtx_shade.hip: the k_shade instantiations are a translation unit of their own, compiled beside this one)
template <int MODE>
static void launch_shade(bool general, bool lean, bool bounced, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p, bool qlights = false, int lds = 0) {
  rtx_launch_shade(MODE, general, lean, bounced, grid, block, stream, d, fp, p, qlights, lds);
}

extern "C" int rt_render(rt_scene* s, const rt_camera* cam, const rt_film_desc* film, const rt_sampler_desc* smp, const rt_path_desc* path,
                         const rt_shard* shard, uint32_t flags, void* stream_, float* film_xyzw, rt_stats* stats_out) {
  if (!s || !cam || !film || !smp || !path || !film_xyzw) return fail(RT_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> render_lock(s->render_mutex);
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned spp = next_pow2((unsigned)(smp->spp > 0 ? smp->spp : 1));
  const unsigned dims = (unsigned)smp->dimensions;
  if (dims < 2 || dims > 8) return fail(RT_ERR_INVALID, "sampler dimensions must be in [2, 8]");
  if (spp > 16384) return fail(RT_ERR_INVALID, "spp > 16384 unsupported");
  unsigned spp_log2 = 0; while ((1u << spp_log2) < spp) ++spp_log2;
  const int rank = shard ? shard->rank : 0, world = shard ? shard->world_size : 1;
  if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad shard");
  const int W = film->sample_bounds[2] - film->sample_bounds[0], H = film->sample_bounds[3] - film->sample_bounds[1];
  const int cw = film->cropped_pixel_bounds[2] - film->cropped_pixel_bounds[0], ch = film->cropped_pixel_bounds[3] - film->cropped_pixel_bounds[1];
  if (W <= 0 || H <= 0 || cw <= 0 || ch <= 0) return fail(RT_ERR_INVALID, "empty film");
  if (film->filter_radius[0] > 8.0f || film->filter_radius[1] > 8.0f) return fail(RT_ERR_INVALID, "filter radius > 8");
  auto t_begin = std::chrono::steady_clock::now();

  FrameParams fp{};
  memcpy(fp.r2c, cam->raster_to_camera, 64); memcpy(fp.c2w, cam->camera_to_world, 64);
  fp.dx_camera = f3{cam->dx_camera[0], cam->dx_camera[1], cam->dx_camera[2]}; fp.dy_camera = f3{cam->dy_camera[0], cam->dy_camera[1], cam->dy_camera[2]};
  fp.lens_radius = cam->lens_radius; fp.focal_distance = cam->focal_distance;
  fp.crop_x0 = film->cropped_pixel_bounds[0]; fp.crop_y0 = film->cropped_pixel_bounds[1]; fp.crop_x1 = film->cropped_pixel_bounds[2]; fp.crop_y1 = film->cropped_pixel_bounds[3];
  fp.sb_x0 = film->sample_bounds[0]; fp.sb_y0 = film->sample_bounds[1]; fp.sb_x1 = film->sample_bounds[2]; fp.sb_y1 = film->sample_bounds[3];
  fp.radius_x = film->filter_radius[0]; fp.radius_y = film->filter_radius[1]; fp.max_sample_luminance = film->max_sample_luminance;
  fp.max_depth = (int)(uint8_t)path->max_depth; fp.rr_threshold = path->rr_threshold;
  fp.pb_x0 = path->pixel_bounds[0]; fp.pb_y0 = path->pixel_bounds[1]; fp.pb_x1 = path->pixel_bounds[2]; fp.pb_y1 = path->pixel_bounds[3];
  fp.rank = rank; fp.world = world;
  const int band = RT_SHARD_ROWS(H, world);  // rows per shard band (a power of two: 4 on a sharded frame)
  fp.shard_log2 = 0; while ((1 << fp.shard_log2) < band) fp.shard_log2 += 1;
  fp.w_recip = W > 1 ? (unsigned)((1ull << 32) / (unsigned long long)W) : 0u;

  // owned sample rows (tile rows of 16, interleaved over ranks)
  unsigned long long owned_rows = 0;
  for (int row = 0; row < H; ++row) if (((row >> fp.shard_log2) % world) == rank) owned_rows++;
  const unsigned long long owned_pixels = owned_rows * (unsigned long long)W;

  rt_stats stats{};
  KTimer tm{(flags & RT_FLAG_TIME_KERNELS) != 0, stream, &s->event_pool};

  // integrator.preprocess (renderer.rs:30)
  tm.begin(&stats.ms_lightdist);
  int rc = build_light_distribution(s, path->light_strategy, stream);
  if (rc != RT_OK) return rc;
  tm.end();

  if (flags & RT_FLAG_REF_STREAM) {  // the reference's own sampler stream: one lane per 16 x 16 tile (rtx_ref.hip)
    if (world != 1) return fail(RT_ERR_UNSUPPORTED, "the reference-stream frame renders on one device");
    if (s->general_prims) return fail(RT_ERR_UNSUPPORTED, "the reference-stream frame takes scenes of plain triangles");
    if (s->stack_depth > 64) return fail(RT_ERR_UNSUPPORTED, "BVH deeper than the reference's 64-entry stack");
    RefParams rp{};
    rp.spp = spp; rp.dims = dims; rp.tile = 16; rp.ntx = (W + 15) / 16; rp.nty = (H + 15) / 16;
    const size_t n_tiles = (size_t)rp.ntx * rp.nty;
    HIP_TRY(s->ref_samples.ensure(n_tiles * 3 * dims * spp * 4)); HIP_TRY(s->ref_stack.ensure(n_tiles * 64 * 4));
    HIP_TRY(s->film_acc.ensure((size_t)cw * ch * 16)); HIP_TRY(s->filter_table.ensure(1024)); HIP_TRY(s->stats.ensure((size_t)ST_COUNT * 8));
    if (!(flags & RT_FLAG_FILM_ON_DEVICE)) HIP_TRY(s->film_out.ensure((size_t)cw * ch * 16));
    float4* const d_out = (flags & RT_FLAG_FILM_ON_DEVICE) ? (float4*)film_xyzw : s->film_out.as<float4>();
    HIP_TRY(hipMemcpyAsync(s->filter_table.p, film->filter_table, 1024, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(s->film_acc.p, 0, (size_t)cw * ch * 16, stream));
    HIP_TRY(hipMemsetAsync(s->stats.p, 0, ST_COUNT * 8, stream));
    rp.samples = s->ref_samples.as<float>(); rp.stack = s->ref_stack.as<int>(); rp.film_acc = s->film_acc.as<float4>();
    rp.filter_table = s->filter_table.as<float>(); rp.stats = s->stats.as<unsigned long long>();
    rtx_launch_render_ref(s->d, fp, rp, stream);
    const unsigned long long n = (unsigned long long)cw * ch;
    hipLaunchKernelGGL(k_film_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s->film_acc.as<float4>(), d_out, n);
    HIP_TRY(hipGetLastError());
    if (!(flags & RT_FLAG_FILM_ON_DEVICE)) HIP_TRY(hipMemcpyAsync(film_xyzw, d_out, (size_t)cw * ch * 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long h5[5];
    HIP_TRY(hipMemcpy(h5, s->stats.p, sizeof(h5), hipMemcpyDeviceToHost));
    stats.camera_rays = h5[0]; stats.rays_closest = h5[1]; stats.rays_shadow = h5[2]; stats.rays_mis = h5[3]; stats.paths_scrubbed = h5[4];
    stats.n_passes = 1;
    stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats_out) *stats_out = stats;
    return RT_OK;
  }
  // pixel_bounds (path.rs:30) that do not crop the sample bounds: every generated sample is traced
  const bool all_in_bounds = path->pixel_bounds[0] <= film->sample_bounds[0] && path->pixel_bounds[2] >= film->sample_bounds[2] &&
                             path->pixel_bounds[1] <= film->sample_bounds[1] && path->pixel_bounds[3] >= film->sample_bounds[3];
  // batch / pass sizing. A batch is a range of owned pixels whose sampler tables (2*dims u16 per sample) are built
  // at once; it is rendered in passes of n_samples consecutive samples of all its pixels, ~2^29 paths (211 GB of path state) per pass.
  // measurement knobs; 2^23 -> 2^26 paths per pass: -9 % (S1), -32 % (S2); 2^26 -> 2^28 with 2^19-pixel batches: -1 % (S1), -7 % (S2), -1 % (S3):
  // fewer, larger launches - late bounces hold few rays - and 288 GB of HBM hold the path state of such a pass with room to spare. On a
  // device with less free memory (a busy GPU, several live scenes, a smaller part) the pass is halved until its workspace fits what
  // hipMemGetInfo reports, and halved again if an allocation fails all the same.
  auto env_log2 = [](const char* name, int dflt) { const char* e = getenv(name); return e ? std::min(30, std::max(16, atoi(e))) : dflt; };
  static const int tp_log2 = env_log2("RTX_PASS_LOG2", 29);  // (round 4: 2^28 -> 2^29 paths per pass, 211 GB of workspace: S1 815.9 -> 808.5 ms, S4 5764 -> 5728, S2 167.4 -> 164.9)
  static const int bp_log2 = env_log2("RTX_BATCH_LOG2", 19);
  enum { B_GEN0, B_GEN1, B_HIT, B_LACC, B_PFILM, B_SH, B_MI, B_QSH, B_QMI, B_QMA, B_OCCSH, B_OCCMI, B_COUNT };  // two generations of travelling path records + their hits (by queue slot), radiance and film position (by path id), ray records and the three ray queues
  const bool has_infinite = s->d.n_infinite > 0;
  const size_t counter_words = (size_t)(fp.max_depth + 2) * RT_NQ * RT_QSHARDS * RT_CNT_STRIDE;  // one block of {out, shadow, mis, mis-any} shard counts per bounce + raygen's
  // material binning before the generic shade kernel: per bounce {hist, cursor}[RT_BIN_MAX + 1] + the 8 count words of the sorted queue
  const bool use_bins = !s->lambert_materials && s->n_code_classes > 1;  // one class: every vertex runs the same code, the queue order is kept
  // (moving the travelling records into sorted order so that the shade launches stream them was measured in round 3: S3 shade 121.6 -> 135.5 ms, S4 3158 -> 3235 - a bin keeps
  // the queue's order in rounds of 256 entries, so the gather through the sorted list already reads runs of neighbouring slots. Removed.)
  const bool gshade = s->masked_emitters;  // quadric / instance hits, quadric or masked emitters: the GENERAL instantiations of the shade kernels
  const bool lean_shade = s->lean_shade;
  const int lds_shade = (s->lds_records_q && s->d.route_quadric_hits != 0) ? 1 : ((lean_shade || gshade) ? (s->lds_mats ? 2 : 0) : (s->lds_tables ? 3 : 0));  // the LEAN / QLIGHTS forms' tables in LDS
  const bool qlights = s->d.route_quadric_hits != 0 && use_bins;  // QLIGHTS forms on the front-end ranges, quadric hits in a generic bin of their own
  const unsigned n_bins = std::min<unsigned>(s->n_code_classes, RT_BIN_MAX - 1) + 1u + (qlights ? 1u : 0u);
  const size_t bin_stride = (RT_BIN_MAX + 1) + (size_t)(RT_BIN_MAX + 1) * RT_CNT_STRIDE + (size_t)RT_QSHARDS * RT_CNT_STRIDE + 10;  // hist, cursors (spread), the sorted queue's counts (laid out as shard counters)  // + {begin, end} of the four class ranges and of the miss bin
  // RTX_SHADE_SPLIT (measurement knob): 0 = every class through the generic front-end, 1 = Lambert classes apart, default = Lambert and two-lobe classes apart
  const int split_mode = env_int("RTX_SHADE_SPLIT", 2);
  const unsigned n_first = split_mode >= 1 ? s->n_lambert_classes : 0u, n_second = split_mode >= 2 ? s->n_small_classes : 0u;
  const unsigned n_third = split_mode >= 2 ? s->n_wide_classes : 0u;
  const bool split_classes = use_bins && (n_first + n_second + n_third) > 0 && n_first + n_second + n_third < RT_BIN_MAX;
  const unsigned pgrid_q = (unsigned)s->n_cu * 8u;
  const unsigned long long table_bytes_per_pixel = 2ull * dims * spp * 2ull;

  unsigned long long batch_pixels = 0, chunk_pixels = 0, cap = 0, lead_pixels = 0; unsigned pass_samples = 0, shard_cap = 0; bool multi_batch = false; size_t n_slots = 0;
  static const char* lead_env = getenv("RTX_LEAD_BATCH");  // measurement knob (round 4), see lead_pixels below: 1 always, 0 never, unset: at 1024 spp
  const bool lead_on = lead_env ? lead_env[0] == '1' : spp == 1024u;
  for (int shrink = 0;; ++shrink) {
    if (tp_log2 - shrink < 14) return fail(RT_ERR_OOM, "not enough free device memory for the smallest pass (2^14 paths)");
    const unsigned long long target_paths = 1ull << (tp_log2 - shrink);
    batch_pixels = std::max<unsigned long long>(1, std::min<unsigned long long>(owned_pixels, std::min(1ull << bp_log2, target_paths)));  // a rank may own no rows
    while (batch_pixels > 4096 && batch_pixels * table_bytes_per_pixel > (16ull << 30)) batch_pixels >>= 1;  // <= 16 GiB of tables per buffer
    // A short LEADING batch (RTX_LEAD_BATCH; on at 1024 spp since the end of round 4): the first batch's sampler tables are the only ones no path kernel runs over, so a
    // first batch of an eighth of the others lets the path kernels start early. Measured alone it gained nothing (the NEXT batch's tables are then built under that
    // short batch's kernels on the low-priority stream, where they take 3 - 4x their time alone: S1 883 -> 881 ms, S3 328 -> 353); together with the parallel replay
    // of the shuffles at 1024 spp it does (S1 679.5 -> 667.5 ms, S4 5280 -> 5242).
    lead_pixels = 0;
    if (lead_on && owned_pixels >= (1ull << 16) && owned_pixels * spp >= (1ull << 27)) lead_pixels = std::max<unsigned long long>(1ull << 15, (batch_pixels / 8) & ~4095ull);
    if (lead_pixels >= owned_pixels || lead_pixels >= batch_pixels) lead_pixels = 0;
    // a shard without a leading batch that fits one batch (a 256 x 256 window; 1/8 of a frame on an 8-GPU run) is still cut in two, so that the second half's
    // sampler tables are built under the first half's path kernels; only worth it when there is enough work to hide them under (the rule used to test
    // lead_on instead of lead_pixels, so 2^15 <= owned_pixels < 2^16 at 1024 spp rendered as one batch)
    if (lead_pixels == 0 && owned_pixels <= batch_pixels && owned_pixels >= (1ull << 15) && owned_pixels * spp >= (1ull << 27)) batch_pixels = (owned_pixels + 1) / 2;
    pass_samples = (unsigned)std::max<unsigned long long>(1, target_paths / batch_pixels);
    if (pass_samples > spp) pass_samples = spp;
    chunk_pixels = batch_pixels;
    cap = batch_pixels * pass_samples;
    if (cap > 0x7fffffffull) return fail(RT_ERR_INVALID, "pass too large");
    // a shard receives the appends of the blocks with blockIdx % RT_QSHARDS == shard; a grid-stride loop hands
    // each block at most ceil(n / (grid * 256)) iterations, so cap / RT_QSHARDS plus one iteration per block bounds it
    shard_cap = (unsigned)(cap / RT_QSHARDS) + 256u * (pgrid_q / RT_QSHARDS + 1u) + 256u;
    const size_t szq = (size_t)shard_cap * RT_QSHARDS * 4;
    n_slots = (size_t)shard_cap * RT_QSHARDS;  // slots of a sharded queue (>= cap: bounce 0 of a fully traced pass uses slot = path id)
    multi_batch = owned_pixels > chunk_pixels || lead_pixels > 0;
    struct Want { DevBuf* buf; size_t bytes; };
    std::vector<Want> want = {
        {&s->ws[B_GEN0], n_slots * 64}, {&s->ws[B_GEN1], n_slots * 64}, {&s->ws[B_HIT], n_slots * 16}, {&s->ws[B_LACC], cap * 16}, {&s->ws[B_PFILM], cap * 8},
        {&s->ws[B_SH], cap * 48}, {&s->ws[B_MI], cap * 100}, {&s->ws[B_OCCSH], cap}, {&s->ws[B_OCCMI], cap},
        {&s->ws[B_QSH], szq}, {&s->ws[B_QMI], szq}, {&s->ws[B_QMA], has_infinite ? szq : 16},
        {&s->counters, counter_words * 4}, {&s->stats, (size_t)ST_COUNT * 8}, {&s->film_acc, (size_t)cw * ch * 16}, {&s->own_acc, (size_t)chunk_pixels * 16},
        {&s->filter_table, 1024}, {&s->scrambles[0], (size_t)chunk_pixels * 3 * dims * 4}, {&s->perms[0], (size_t)(chunk_pixels * table_bytes_per_pixel)},
        {&s->sampler_plan.partners, (size_t)(chunk_pixels * table_bytes_per_pixel)}};
    if (use_bins) { want.push_back({&s->bin_words, (size_t)(fp.max_depth + 1) * bin_stride * 4}); want.push_back({&s->bin_sorted, (size_t)cap * 4}); want.push_back({&s->bin_at, (size_t)cap * 2}); }
    if (multi_batch) { want.push_back({&s->scrambles[1], (size_t)chunk_pixels * 3 * dims * 4}); want.push_back({&s->perms[1], (size_t)(chunk_pixels * table_bytes_per_pixel)}); }
    if (!(flags & RT_FLAG_FILM_ON_DEVICE)) want.push_back({&s->film_out, (size_t)cw * ch * 16});
    size_t grow = 0;  // bytes the buffers have to grow by (a buffer that is too small is freed and allocated anew)
    for (const Want& w : want) if (!w.buf->p || w.buf->bytes < w.bytes) grow += w.bytes;
    size_t held = 0;
    for (const Want& w : want) if (w.buf->p && w.buf->bytes < w.bytes) held += w.buf->bytes;
    size_t mem_free = 0, mem_total = 0;
    HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
    if (grow > mem_free + held - std::min<size_t>(mem_free + held, (size_t)256 << 20)) continue;  // keep 256 MiB clear of the workspace
    bool ok = true;
    for (const Want& w : want) if (w.buf->ensure(w.bytes) != hipSuccess) { ok = false; break; }
    if (ok) break;
    (void)hipGetLastError();  // the failed hipMalloc; try again with half the pass
  }
  if (!s->aux_stream) {
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = numerically greatest = lowest priority
    HIP_TRY(hipStreamCreateWithPriority(&s->aux_stream, hipStreamNonBlocking, lo));
    for (int i = 0; i < 2; ++i) { for (int g = 0; g < 3; ++g) HIP_TRY(hipEventCreateWithFlags(&s->ev_tables[i][g], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->ev_batch_done[i], hipEventDisableTiming)); }
    HIP_TRY(hipEventCreateWithFlags(&s->ev_frame_begin, hipEventDisableTiming));
  }
  float4* const d_out = (flags & RT_FLAG_FILM_ON_DEVICE) ? (float4*)film_xyzw : s->film_out.as<float4>();
  HIP_TRY(hipMemcpyAsync(s->filter_table.p, film->filter_table, 1024, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(s->film_acc.p, 0, (size_t)cw * ch * 16, stream));
  HIP_TRY(hipMemsetAsync(s->stats.p, 0, ST_COUNT * 8, stream));

  PassState ps{};
  ps.spp = spp; ps.spp_log2 = spp_log2; ps.dims = dims;
  auto gen_of = [&](int b) { PathGen g; char* p = (char*)s->ws[b].p; g.o = (float4*)p; g.d = (float4*)(p + n_slots * 16); g.beta = (float4*)(p + n_slots * 32); g.st = (uint4*)(p + n_slots * 48); return g; };
  const PathGen gen0 = gen_of(B_GEN0), gen1 = gen_of(B_GEN1);
  ps.hit = s->ws[B_HIT].as<float4>(); ps.lacc = s->ws[B_LACC].as<float4>(); ps.pfilm = s->ws[B_PFILM].as<float2>();
  { char* p = (char*)s->ws[B_SH].p; ps.sh.o = (float4*)p; ps.sh.d = (float4*)(p + cap * 16); ps.sh.add = (float4*)(p + cap * 32);
    char* q = (char*)s->ws[B_MI].p; ps.mi.o = (float4*)q; ps.mi.d = (float4*)(q + cap * 16); ps.mi.hit = (float4*)(q + cap * 32); ps.mi.a = (float4*)(q + cap * 48);
    ps.mi.b = (float4*)(q + cap * 64); ps.mi.c = (float4*)(q + cap * 80); ps.mi.flags = (unsigned*)(q + cap * 96); }
  ps.occ_sh = s->ws[B_OCCSH].as<unsigned char>(); ps.occ_mi = s->ws[B_OCCMI].as<unsigned char>();
  ps.q_shadow = s->ws[B_QSH].as<unsigned>(); ps.q_mis = s->ws[B_QMI].as<unsigned>(); ps.q_misany = s->ws[B_QMA].as<unsigned>();
  ps.own_acc = s->own_acc.as<float4>(); ps.shard_cap = shard_cap;
  ps.stats = s->stats.as<unsigned long long>();

  if ((rc = sampler_plan_prepare(s->sampler_plan, spp, dims)) != RT_OK) return rc;
  if ((rc = sampler_set_lds_limits(spp)) != RT_OK) return rc;
  // the four kinds of trace launch of a bounce, on the records of the pass
  TraceIO io_path{}, io_shadow{}, io_mis{}, io_mis_any{};
  {
    io_path.ray_stride = 1; io_path.hits = ps.hit; io_path.hit_stride = 1; io_path.hit_b2 = 1; io_path.queue_is_slots = 1;  // (ray_o / ray_d: the bounce's generation)
    io_shadow.ray_o = ps.sh.o; io_shadow.ray_d = ps.sh.d; io_shadow.ray_stride = 1; io_shadow.occluded = (unsigned*)ps.occ_sh; io_shadow.occ_stride = 0;
    io_shadow.shadow_masks = 1; io_shadow.lacc = ps.lacc; io_shadow.lacc_stride = 1; io_shadow.direct_add = ps.sh.add; io_shadow.add_stride = 1;
    io_mis.ray_o = ps.mi.o; io_mis.ray_d = ps.mi.d; io_mis.ray_stride = 1; io_mis.hits = ps.mi.hit; io_mis.hit_stride = 1; io_mis.hit_b2 = 0;
    io_mis_any = io_mis; io_mis_any.shadow_masks = 0; io_mis_any.hits = nullptr; io_mis_any.occluded = (unsigned*)ps.occ_mi; io_mis_any.occ_stride = 0;
  }
  const bool count = (flags & RT_FLAG_COUNT_TRAVERSAL) != 0;
  const unsigned pgrid = (unsigned)s->n_cu * 8u;
  // workgroup size of the shade launches (measurement knob RTX_SHADE_BLOCK = 64 | 128 | 256): the queue appends of a workgroup meet at three barriers per
  // iteration, so its waves run in lockstep; smaller workgroups trade that for more atomics on the shard counters. Same number of lanes in the grid.
  const unsigned sblock = 256u, sgrid = pgrid;
  unsigned long long* dstats = s->stats.as<unsigned long long>();

  // K0 of a batch goes to the low-priority aux stream: it is a latency-bound chain of LDS swaps that leaves the
  // VALUs idle, so it runs underneath the previous batch's path kernels. ev_tables[k]: tables in buffer k are
  // ready; ev_batch_done[k]: the path kernels that read buffer k have finished (it may be overwritten).
  // RTX_K0_OVERLAP=0 (measurement knob) keeps K0 on the render stream
  const char* ov = getenv("RTX_K0_OVERLAP");
  hipStream_t aux = (ov && ov[0] == '0') ? stream : s->aux_stream;
  HIP_TRY(hipEventRecord(s->ev_frame_begin, stream));
  HIP_TRY(hipStreamWaitEvent(aux, s->ev_frame_begin, 0));
  int tables_rc = RT_OK;
  std::vector<std::pair<unsigned long long, unsigned long long>> batches;  // (first owned pixel, pixels)
  for (unsigned long long first = 0; first < owned_pixels;) {
    const unsigned long long npx = (first == 0 && lead_pixels) ? lead_pixels : std::min(chunk_pixels, owned_pixels - first);
    batches.push_back({first, npx}); first += npx;
  }
  TableGroups tgroups = table_groups_frame(dims, !(fp.lens_radius > 0.0f));
  auto launch_tables = [&](size_t b, int buf) {
    FrameParams f2 = fp; f2.chunk_first = batches[b].first;
    tm.begin(&stats.ms_sampler, aux);
    const int trc = launch_sampler_tables(s->sampler_plan, f2, (unsigned)batches[b].second, 0ull, 0, s->scrambles[buf].as<unsigned>(), s->perms[buf].as<unsigned short>(), aux, tgroups, s->ev_tables[buf]);
    if (trc != RT_OK && tables_rc == RT_OK) tables_rc = trc;
    tm.end(aux);
  };
  if (!batches.empty()) launch_tables(0, 0);  // a rank may own no rows
  for (size_t batch_no = 0; batch_no < batches.size(); ++batch_no) {
    const unsigned long long first = batches[batch_no].first, npx = batches[batch_no].second;
    const int buf = (int)(batch_no & 1ull);
    fp.chunk_first = first;
    ps.n_pixels = (unsigned)npx; ps.n_pixels_recip = npx > 1 ? (unsigned)((1ull << 32) / npx) : 0xffffffffu;
    ps.scrambles = s->scrambles[buf].as<unsigned>(); ps.perms = s->perms[buf].as<unsigned short>();
    if (batch_no + 1 < batches.size()) {  // next batch's tables into the other buffer, once its previous readers are done
      if (batch_no >= 1) HIP_TRY(hipStreamWaitEvent(aux, s->ev_batch_done[buf ^ 1], 0));
      launch_tables(batch_no + 1, buf ^ 1);
    }
    const unsigned batch_pass_samples = (unsigned)std::min<unsigned long long>(spp, std::max<unsigned long long>(pass_samples, cap / npx));  // (a short batch: more samples per pass, same paths)
    if (tables_rc != RT_OK) { (void)hipDeviceSynchronize(); return tables_rc; }
    HIP_TRY(hipStreamWaitEvent(stream, s->ev_tables[buf][0], 0));  // the tables the camera samples read; the others are waited for where they are first read
    for (unsigned s0 = 0; s0 < spp; s0 += batch_pass_samples) {
      ps.s0 = s0; ps.n_samples = std::min(batch_pass_samples, spp - s0); ps.cap = (unsigned)(npx * ps.n_samples);
      ps.q_in = nullptr; ps.in = gen1; ps.out = gen0;  // raygen writes bounce 0's records
      HIP_TRY(hipMemsetAsync(s->counters.p, 0, counter_words * 4, stream));
      if (use_bins) HIP_TRY(hipMemsetAsync(s->bin_words.p, 0, (size_t)(fp.max_depth + 1) * bin_stride * 4, stream));
      unsigned* const cb = s->counters.as<unsigned>();
      ps.cnt_in = nullptr; ps.cnt_out = cb;  // raygen appends to block 0's `out` queue
      ps.all_in_bounds = all_in_bounds ? 1 : 0;
      const int fresh_planes = all_in_bounds ? ((s->lambert_only && s->lds_records) ? RT_FRESH_RECORDS_LDS : RT_FRESH_RECORDS) : 0;  // which records k_raygen leaves out (PassState::fresh)
      ps.fresh = fresh_planes;
      // a frame that counts node visits keeps the reference's closest-hit walk for every MIS ray, unless it is asked to count what a production frame walks
      ps.mis_any = (has_infinite && (!count || (flags & RT_FLAG_COUNT_AS_RENDERED))) ? 1 : 0;
      static const bool reach_off = env_is("RTX_MIS_REACH", '0');  // measurement knob
      ps.skip_unreachable_mis = (!reach_off && (!count || (flags & RT_FLAG_COUNT_AS_RENDERED))) ? 1 : 0;
      static const bool tail_off = env_is("RTX_DEAD_TAIL", '0');  // measurement knob: cast the rays nothing reads as well
      ps.skip_dead_tail = (!tail_off && (!count || (flags & RT_FLAG_COUNT_AS_RENDERED))) ? 1 : 0;
      tm.begin(&stats.ms_raygen);
      hipLaunchKernelGGL(k_raygen, dim3(pgrid), dim3(256), 0, stream, fp, ps);
      tm.end();
      for (int bounce = 0; bounce <= fp.max_depth; ++bounce) {
        ps.cnt_in = cb + (size_t)bounce * RT_NQ * RT_QSHARDS * RT_CNT_STRIDE; ps.cnt_out = cb + (size_t)(bounce + 1) * RT_NQ * RT_QSHARDS * RT_CNT_STRIDE;
        std::swap(ps.in, ps.out);  // what the previous stage appended is this bounce's input
        if (bounce == 0 && all_in_bounds) ps.cnt_in = nullptr;  // identity: entry i is slot i is path i
        ps.fresh = bounce == 0 ? fresh_planes : 0;                // ... whose throughput / state records are rebuilt, not read
        io_path.ray_o = ps.in.o; io_path.ray_d = ps.in.d;
        tm.begin(&stats.ms_trace_closest);
        // the rays of the bounce sit at their queue slots: the kernels walk the entries by the shard counts alone (a non-NULL `queue` only says "sharded")
        launch_trace<false>(s, count, io_path, ps.cnt_in, ps.cnt_in, ps.shard_cap, ps.cap, dstats, ST_RAYS_CLOSEST, ST_NODES_CLOSEST, ST_TRIS_CLOSEST, stream);
        tm.end();
        if (bounce <= 1) HIP_TRY(hipStreamWaitEvent(stream, s->ev_tables[buf][bounce + 1], 0));  // table groups 1 / 2: first read by the shade launches of bounce 0 / 1
#define RT_SHADE(MODE, P) stats.launches_shade += 1, launch_shade<MODE>(gshade, lean_shade, bounce >= 1, sgrid, sblock, stream, s->d, fp, P, qlights, lds_shade)
        if (s->lambert_only) {
          tm.begin(&stats.ms_shade_lambert_const);
          rtx_launch_shade_const(s->lds_records ? 1 : (s->lds_tables ? 3 : 0), sgrid, sblock, stream, s->d, fp, ps);  // LDSREC 3: a large mesh with few lights / materials (S2)
          tm.end(); stats.launches_shade += 1;
        }
        else if (s->lambert_materials) { tm.begin(&stats.ms_shade_lambert); RT_SHADE(3, ps); tm.end(); }
        else if (!use_bins) { tm.begin(&stats.ms_shade_generic); RT_SHADE(0, ps); tm.end(); }
        else {
          tm.begin(&stats.ms_shade_bin);
          unsigned* bw = s->bin_words.as<unsigned>() + (size_t)bounce * bin_stride;
          unsigned* hist = bw; unsigned* cursor = bw + (RT_BIN_MAX + 1); unsigned* sorted_cnt = cursor + (size_t)(RT_BIN_MAX + 1) * RT_CNT_STRIDE;
          hipLaunchKernelGGL(k_bin_count, dim3(pgrid), dim3(256), 0, stream, s->d, ps, n_bins, hist, s->bin_at.as<unsigned short>());
          unsigned* ranges = sorted_cnt + (size_t)RT_QSHARDS * RT_CNT_STRIDE;
#define RT_BIN_ARGS dim3(pgrid), dim3(256), 0, stream, s->d, ps, n_bins, hist, cursor, s->bin_sorted.as<unsigned>(), sorted_cnt, split_classes ? n_first : 0u, split_classes ? n_first + n_second : 0u, \
                    split_classes ? n_first + n_second + n_third : 0u, ranges, s->bin_at.as<unsigned short>()
          hipLaunchKernelGGL(k_bin_scatter, RT_BIN_ARGS);
#undef RT_BIN_ARGS
          tm.end();
          PassState pb = ps; pb.q_in = s->bin_sorted.as<unsigned>(); pb.cnt_in = sorted_cnt;  // all entries in shard 0: QView::get(i) = ids[i]
          // classes of the register-resident front-ends, then the generic one, then the rays that left the scene
          if (split_classes && n_first) { pb.range = ranges; tm.begin(&stats.ms_shade_lambert); RT_SHADE(3, pb); tm.end(); }
          if (split_classes && n_second) { pb.range = ranges + 2; tm.begin(&stats.ms_shade_two_lobe); RT_SHADE(5, pb); tm.end(); }
          if (split_classes && n_third) { pb.range = ranges + 8; tm.begin(&stats.ms_shade_two_lobe); RT_SHADE(6, pb); tm.end(); }
          pb.range = ranges + 4; tm.begin(&stats.ms_shade_generic); RT_SHADE(0, pb); tm.end();
          pb.range = ranges + 6; tm.begin(&stats.ms_shade_miss); hipLaunchKernelGGL(k_shade_miss, dim3(pgrid), dim3(256), 0, stream, s->d, pb); tm.end();
        }
#undef RT_SHADE
        tm.begin(&stats.ms_trace_any);
        launch_trace<true>(s, count, io_shadow, ps.q_shadow, ps.cnt_out + RT_QSHARDS * RT_CNT_STRIDE, ps.shard_cap, 0, dstats, ST_RAYS_SHADOW, ST_NODES_SHADOW, ST_TRIS_SHADOW, stream);
        tm.end();
        tm.begin(&stats.ms_trace_mis);
        launch_trace<false>(s, count, io_mis, ps.q_mis, ps.cnt_out + 2 * RT_QSHARDS * RT_CNT_STRIDE, ps.shard_cap, 0, dstats, ST_RAYS_MIS, ST_NODES_MIS, ST_TRIS_MIS, stream);
        tm.end();
        if (ps.mis_any) {
          tm.begin(&stats.ms_trace_mis_any);
          launch_trace<true>(s, count, io_mis_any, ps.q_misany, ps.cnt_out + 3 * RT_QSHARDS * RT_CNT_STRIDE, ps.shard_cap, 0, dstats, ST_RAYS_MISANY, ST_NODES_MISANY, ST_TRIS_MISANY, stream);
          tm.end();
        }
        tm.begin(&stats.ms_resolve);
        if (s->has_spheres || s->has_instances) hipLaunchKernelGGL(k_resolve<true>, dim3(pgrid), dim3(256), 0, stream, s->d, ps);  // (a hit id inside an instance is not a primitive index)
        else hipLaunchKernelGGL(k_resolve<false>, dim3(pgrid), dim3(256), 0, stream, s->d, ps);
        tm.end();
        stats.launches_trace_closest += 2; stats.launches_trace_path += 1; stats.launches_trace_mis += 1; stats.launches_trace_shadow += 1; stats.launches_trace_mis_any += ps.mis_any ? 1 : 0;
      }
      tm.begin(&stats.ms_film);
      hipLaunchKernelGGL(k_film_accumulate, dim3(pgrid), dim3(256), 0, stream, fp, ps, s->filter_table.as<float>(), s->film_acc.as<float4>());
      tm.end();
      stats.n_passes += 1;
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(s->ev_batch_done[buf], stream));
  }
  tm.begin(&stats.ms_film);
  {
    unsigned long long n = (unsigned long long)cw * ch;
    hipLaunchKernelGGL(k_film_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s->film_acc.as<float4>(), d_out, n);
  }
  tm.end();
  HIP_TRY(hipGetLastError());
  if (!(flags & RT_FLAG_FILM_ON_DEVICE)) HIP_TRY(hipMemcpyAsync(film_xyzw, d_out, (size_t)cw * ch * 16, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  tm.collect();
  stats.ms_shade = stats.ms_shade_lambert_const + stats.ms_shade_lambert + stats.ms_shade_two_lobe + stats.ms_shade_generic + stats.ms_shade_bin + stats.ms_shade_miss;
  unsigned long long h[ST_COUNT];
  HIP_TRY(hipMemcpy(h, s->stats.p, sizeof(h), hipMemcpyDeviceToHost));
  { unsigned ovf = 0; HIP_TRY(hipMemcpy(&ovf, s->sampler_plan.dirty.as<unsigned>() + 1 + RT_DIRTY_CAP, 4, hipMemcpyDeviceToHost));
    if (ovf) { (void)hipMemset(s->sampler_plan.dirty.p, 0, (2 + RT_DIRTY_CAP) * 4); return fail(RT_ERR_INVALID, "sampler retry list overflow"); } }
  stats.camera_rays = h[ST_CAMERA];  // counted by k_raygen: samples inside pixel_bounds and the film's sample rows
  stats.rays_mis_any = h[ST_RAYS_MISANY]; stats.nodes_mis_any = h[ST_NODES_MISANY]; stats.tris_mis_any = h[ST_TRIS_MISANY];
  stats.rays_closest = h[ST_RAYS_CLOSEST] + h[ST_TAIL_UNCAST]; stats.rays_tail_not_cast = h[ST_TAIL_UNCAST]; stats.rays_shadow = h[ST_RAYS_SHADOW]; stats.rays_mis = h[ST_RAYS_MIS] + h[ST_RAYS_MISANY] + h[ST_MIS_UNREACHED];
  stats.rays_mis_not_cast = h[ST_MIS_UNREACHED];
  stats.nodes_closest = h[ST_NODES_CLOSEST]; stats.nodes_shadow = h[ST_NODES_SHADOW]; stats.nodes_mis = h[ST_NODES_MIS] + h[ST_NODES_MISANY];
  stats.tris_closest = h[ST_TRIS_CLOSEST]; stats.tris_shadow = h[ST_TRIS_SHADOW]; stats.tris_mis = h[ST_TRIS_MIS] + h[ST_TRIS_MISANY];
  stats.ms_trace_mis += stats.ms_trace_mis_any;  // ms_trace_mis stays the time of all MIS launches
  stats.paths_scrubbed = h[ST_SCRUBBED];
  for (int k = 0; k < 32; ++k) stats.shade_section_cycles[k] = h[ST_STAMP + k];
  stats.vertices_lambert_const = h[ST_SHADED]; stats.vertices_lambert = h[ST_SHADED + 1]; stats.vertices_two_lobe = h[ST_SHADED + 2]; stats.vertices_generic = h[ST_SHADED + 3];
  if (h[ST_UNBUILT_VOXEL]) return fail(RT_ERR_INVALID, "a path looked up a light-distribution voxel that holds no surface (voxel marking bug)");
  stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (stats_out) *stats_out = stats;
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------- several GPUs, one process
// renderer::render hands tiles to worker threads from a shared queue and merges each finished tile into the film (rc/renderer.rs:47-71,
// rc/film.rs:177-194). Here the workers are GPUs: the scene is replicated, the frame is cut into chunks of interleaved bands of RT_SHARD_ROWS rows
// (rt_shard), one host thread per device takes chunks from a shared counter, renders each with rt_render into a film on its own device and sends
// the rows that chunk can have touched - its tile rows plus the filter's reach - to the first device over xGMI (hipMemcpyPeerAsync), where they are
// added into the frame in chunk order. Only those rows cross the links; nothing is exchanged while paths are traced.
__global__ void k_film_add(float4* __restrict__ dst, const float4* __restrict__ src, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = dst[i], b = src[i];
  dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// every counter and every timer of b added into a
static void stats_add(rt_stats& a, const rt_stats& b) {
  a.camera_rays += b.camera_rays; a.rays_closest += b.rays_closest; a.rays_shadow += b.rays_shadow; a.rays_mis += b.rays_mis; a.rays_mis_not_cast += b.rays_mis_not_cast; a.rays_tail_not_cast += b.rays_tail_not_cast;
  a.nodes_closest += b.nodes_closest; a.nodes_shadow += b.nodes_shadow; a.nodes_mis += b.nodes_mis;
  a.tris_closest += b.tris_closest; a.tris_shadow += b.tris_shadow; a.tris_mis += b.tris_mis; a.paths_scrubbed += b.paths_scrubbed;
  a.ms_total += b.ms_total; a.ms_sampler += b.ms_sampler; a.ms_raygen += b.ms_raygen; a.ms_trace_closest += b.ms_trace_closest; a.ms_trace_any += b.ms_trace_any;
  a.ms_trace_mis += b.ms_trace_mis; a.ms_shade += b.ms_shade; a.ms_resolve += b.ms_resolve; a.ms_film += b.ms_film; a.ms_lightdist += b.ms_lightdist;
  a.launches_trace_closest += b.launches_trace_closest; a.n_passes += b.n_passes;
  a.launches_trace_path += b.launches_trace_path; a.launches_trace_shadow += b.launches_trace_shadow; a.launches_trace_mis += b.launches_trace_mis; a.launches_trace_mis_any += b.launches_trace_mis_any; a.launches_shade += b.launches_shade;
  a.vertices_lambert_const += b.vertices_lambert_const; a.vertices_lambert += b.vertices_lambert; a.vertices_two_lobe += b.vertices_two_lobe; a.vertices_generic += b.vertices_generic;
  a.ms_shade_lambert_const += b.ms_shade_lambert_const; a.ms_shade_lambert += b.ms_shade_lambert; a.ms_shade_two_lobe += b.ms_shade_two_lobe;
  a.ms_shade_generic += b.ms_shade_generic; a.ms_shade_bin += b.ms_shade_bin; a.ms_shade_miss += b.ms_shade_miss;
  for (int k = 0; k < 32; ++k) a.shade_section_cycles[k] += b.shade_section_cycles[k];
  a.rays_mis_any += b.rays_mis_any; a.nodes_mis_any += b.nodes_mis_any; a.tris_mis_any += b.tris_mis_any; a.ms_trace_mis_any += b.ms_trace_mis_any;
}
struct rt_multi {
  std::vector<int> devices; std::vector<rt_scene*> replicas; std::vector<hipStream_t> streams; std::vector<DevBuf*> chunk_film;
  DevBuf acc; std::vector<DevBuf*> staging;  // on devices[0]: the frame, and one buffer of rows per chunk
  hipStream_t add_stream = nullptr;          // on devices[0]: the additions into the frame, in chunk order, each behind its chunk's copy event only
  std::vector<hipEvent_t> chunk_done;        // per chunk: its rows have arrived in staging (recorded on the worker's stream, on the worker's device)
  std::vector<int> chunk_done_dev;
  std::string warnings;  // conditions that do not fail a call but are worth a line (left in rt_last_error by rt_multi_create)
  ~rt_multi() {
    for (size_t i = 0; i < replicas.size(); ++i) {
      (void)hipSetDevice(devices[i]);
      if (i < streams.size() && streams[i]) (void)hipStreamDestroy(streams[i]);
      if (i < chunk_film.size()) delete chunk_film[i];
      if (replicas[i]) rt_scene_destroy(replicas[i]);
    }
    for (size_t c = 0; c < chunk_done.size(); ++c) { (void)hipSetDevice(chunk_done_dev[c]); (void)hipEventDestroy(chunk_done[c]); }
    if (!devices.empty()) (void)hipSetDevice(devices[0]);
    if (add_stream) (void)hipStreamDestroy(add_stream);
    for (DevBuf* b : staging) delete b;
  }
};
extern "C" int rt_multi_create(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices, rt_multi** out) {
  if (!desc || !devices || n_devices < 1 || !out) return fail(RT_ERR_INVALID, "bad rt_multi_create arguments");
  int n_visible = 0;
  if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  rt_multi* m = new rt_multi();
  for (int i = 0; i < n_devices; ++i) {
    if (devices[i] < 0 || devices[i] >= n_visible) { delete m; return fail(RT_ERR_INVALID, "device index out of range"); }
    m->devices.push_back(devices[i]); m->replicas.push_back(nullptr); m->streams.push_back(nullptr); m->chunk_film.push_back(new DevBuf());
  }
  for (int i = 0; i < n_devices; ++i) {  // a device may be named more than once (several workers on one GPU, each with its own replica)
    const int rc = rt_scene_create(desc, devices[i], &m->replicas[i]);
    if (rc != RT_OK) { const std::string e = g_err; delete m; return fail(rc, e); }
    if (hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess) { delete m; return fail(RT_ERR_HIP, "stream creation failed"); }
    if (devices[i] != devices[0]) {  // let the workers write into the first device's memory directly where the fabric allows it (else the copy is staged by the runtime)
      int can = 0; (void)hipDeviceCanAccessPeer(&can, devices[i], devices[0]);
      hipError_t e = can ? hipDeviceEnablePeerAccess(devices[0], 0) : hipErrorPeerAccessUnsupported;
      if (e != hipSuccess) (void)hipGetLastError();
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)  // not fatal - hipMemcpyPeerAsync then stages through the host - but slow: say so (rt_last_error after a successful create)
        m->warnings += "device " + std::to_string(devices[i]) + " cannot write into device " + std::to_string(devices[0]) + " directly (" + hipGetErrorString(e) + "): film rows are staged by the runtime; ";
    }
  }
  g_err = m->warnings;
  *out = m;
  return RT_OK;
}
extern "C" void rt_multi_destroy(rt_multi* m) { delete m; }

extern "C" int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_film_desc* film, const rt_sampler_desc* smp, const rt_path_desc* path,
                               int32_t chunks_per_device, uint32_t flags, float* film_xyzw, rt_stats* total, rt_stats* per_device) {
  if (!m || !cam || !film || !smp || !path || !film_xyzw) return fail(RT_ERR_INVALID, "null argument");
  const int n_dev = (int)m->devices.size();
  const int n_chunks = n_dev * std::max(1, std::min(64, (int)chunks_per_device));
  const int cw = film->cropped_pixel_bounds[2] - film->cropped_pixel_bounds[0], ch = film->cropped_pixel_bounds[3] - film->cropped_pixel_bounds[1];
  const int H = film->sample_bounds[3] - film->sample_bounds[1];
  if (cw <= 0 || ch <= 0 || H <= 0) return fail(RT_ERR_INVALID, "empty film");
  const auto t_begin = std::chrono::steady_clock::now();
  // film rows chunk c can have written: its tile rows, widened by the filter's reach (FilmTile::add_sample splats ceil(y - 0.5 - r) .. floor(y - 0.5 + r),
  // film.rs:303-321; with the box filter that is the sample's own pixel, plus the one above when the sample sits exactly on the edge)
  const int halo = (int)std::ceil(film->filter_radius[1] - 0.5f) + 1;
  const int band = RT_SHARD_ROWS(H, n_chunks), n_tile_rows = (H + band - 1) / band;  // (the bands rt_render cuts for world_size = n_chunks)
  auto bands = [&](int c) {
    std::vector<std::pair<int, int>> b;  // [first, end) film rows, merged
    for (int t = c; t < n_tile_rows; t += n_chunks) {
      const int y0 = film->sample_bounds[1] + band * t - halo - film->cropped_pixel_bounds[1], y1 = film->sample_bounds[1] + std::min(band * t + band, H) + halo - film->cropped_pixel_bounds[1];
      const int a = std::max(0, y0), e = std::min(ch, y1);
      if (a >= e) continue;
      if (!b.empty() && a <= b.back().second) b.back().second = std::max(b.back().second, e); else b.push_back({a, e});
    }
    return b;
  };
  const size_t frame_bytes = (size_t)cw * ch * 16;
  HIP_TRY(hipSetDevice(m->devices[0]));
  HIP_TRY(m->acc.ensure(frame_bytes));
  HIP_TRY(hipMemset(m->acc.p, 0, frame_bytes));
  HIP_TRY(hipStreamSynchronize(nullptr));  // the additions run on add_stream (non-blocking: not ordered behind the null stream): the clear is complete before the first of them
  while ((int)m->staging.size() < n_chunks) m->staging.push_back(new DevBuf());
  std::vector<std::vector<std::pair<int, int>>> chunk_bands(n_chunks);
  for (int c = 0; c < n_chunks; ++c) {
    chunk_bands[c] = bands(c);
    size_t rows = 0; for (auto& b : chunk_bands[c]) rows += (size_t)(b.second - b.first);
    HIP_TRY(m->staging[c]->ensure(std::max<size_t>(rows * cw * 16, 16)));
  }
  if (!m->add_stream) HIP_TRY(hipStreamCreateWithFlags(&m->add_stream, hipStreamNonBlocking));
  std::atomic<int> next{0}; std::mutex err_mtx; int first_rc = RT_OK; std::string first_err;
  std::vector<rt_stats> dev_stats(n_dev, rt_stats{});
  // Film::merge_film_tile without a mutex and without a barrier: chunk c's rows are added into the frame on devices[0] by a stream of their own (not the null
  // stream: round 4 added everything there after the LAST device had finished) that waits for chunk c's copy event only - so a chunk's addition runs while the
  // other devices still trace. The additions are ISSUED in chunk order by this thread (a pixel two chunks touched - a filter wider than a pixel, a sample exactly
  // on an edge - sums in a fixed order, whatever order the devices finish in); chunk_state[c]: 0 not yet, 1 rows on their way (event recorded), -1 failed.
  std::vector<std::atomic<int>> chunk_state(n_chunks);
  for (auto& a : chunk_state) a.store(0);
  while ((int)m->chunk_done.size() < n_chunks) { m->chunk_done.push_back(nullptr); m->chunk_done_dev.push_back(-1); }
  auto worker = [&](int k) {
    auto bail = [&](int rc, const std::string& msg) { std::lock_guard<std::mutex> g(err_mtx); if (first_rc == RT_OK) { first_rc = rc; first_err = msg; } };
    if (hipSetDevice(m->devices[k]) != hipSuccess || m->chunk_film[k]->ensure(frame_bytes) != hipSuccess) bail(RT_ERR_HIP, "device set-up failed");
    for (;;) {  // the tile queue of renderer.rs:68-71, in chunks
      const int c = next.fetch_add(1);
      if (c >= n_chunks) break;
      bool failed; { std::lock_guard<std::mutex> g(err_mtx); failed = first_rc != RT_OK; }
      if (failed) { chunk_state[c].store(-1); continue; }  // (every chunk's state is set, so the issuing thread never waits for a chunk nobody renders)
      const rt_shard sh{c, n_chunks}; rt_stats st{};
      const int rc = rt_render(m->replicas[k], cam, film, smp, path, &sh, (flags & ~RT_FLAG_FILM_ON_DEVICE) | RT_FLAG_FILM_ON_DEVICE, m->streams[k], m->chunk_film[k]->as<float>(), &st);
      if (rc != RT_OK) { bail(rc, g_err); chunk_state[c].store(-1); continue; }
      size_t off = 0; bool ok = true;
      for (auto& b : chunk_bands[c]) {
        const size_t bytes = (size_t)(b.second - b.first) * cw * 16;
        if (hipMemcpyPeerAsync((char*)m->staging[c]->p + off, m->devices[0], m->chunk_film[k]->as<char>() + (size_t)b.first * cw * 16, m->devices[k], bytes, m->streams[k]) != hipSuccess) { ok = false; break; }
        off += bytes;
      }
      if (ok && (m->chunk_done[c] == nullptr || m->chunk_done_dev[c] != m->devices[k])) {  // an event lives on the device it was created on: chunk c's on the device that renders it this frame
        if (m->chunk_done[c]) { (void)hipSetDevice(m->chunk_done_dev[c]); (void)hipEventDestroy(m->chunk_done[c]); (void)hipSetDevice(m->devices[k]); m->chunk_done[c] = nullptr; }
        ok = hipEventCreateWithFlags(&m->chunk_done[c], hipEventDisableTiming) == hipSuccess; m->chunk_done_dev[c] = m->devices[k];
      }
      if (ok) ok = hipEventRecord(m->chunk_done[c], m->streams[k]) == hipSuccess;
      if (!ok) { bail(RT_ERR_HIP, "peer copy failed"); chunk_state[c].store(-1); continue; }
      chunk_state[c].store(1);  // (the next chunk's kernels on this stream are ordered behind the copies: chunk_film[k] is not overwritten under them)
      stats_add(dev_stats[k], st);
    }
    if (hipStreamSynchronize(m->streams[k]) != hipSuccess) bail(RT_ERR_HIP, "peer copy failed");
  };
  std::vector<std::thread> threads;
  for (int k = 0; k < n_dev; ++k) threads.emplace_back(worker, k);
  bool add_failed = hipSetDevice(m->devices[0]) != hipSuccess;  // (no early return while the workers run)
  for (int c = 0; c < n_chunks; ++c) {
    int stt; while ((stt = chunk_state[c].load()) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (stt < 0 || add_failed) continue;
    if (hipStreamWaitEvent(m->add_stream, m->chunk_done[c], 0) != hipSuccess) { add_failed = true; continue; }
    size_t off = 0;
    for (auto& b : chunk_bands[c]) {
      const unsigned long long n = (unsigned long long)(b.second - b.first) * cw;
      hipLaunchKernelGGL(k_film_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->add_stream, m->acc.as<float4>() + (size_t)b.first * cw, (const float4*)((char*)m->staging[c]->p + off), n);
      off += n * 16;
    }
  }
  for (auto& t : threads) t.join();
  const auto t_gather = std::chrono::steady_clock::now();  // the last device has finished (its rows are in staging): what is left is the tail of the additions
  if (first_rc != RT_OK) { (void)hipStreamSynchronize(m->add_stream); return fail(first_rc, first_err); }
  if (add_failed) return fail(RT_ERR_HIP, "waiting for a chunk's rows failed");
  HIP_TRY(hipStreamSynchronize(m->add_stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(film_xyzw, m->acc.p, frame_bytes, (flags & RT_FLAG_FILM_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
  if (per_device) for (int k = 0; k < n_dev; ++k) per_device[k] = dev_stats[k];
  if (total) {
    rt_stats t{};
    for (const rt_stats& a : dev_stats) stats_add(t, a);  // times are summed over devices (device-milliseconds), ms_total below is the call's wall time
    t.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    t.ms_gather = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gather).count();
    *total = t;
  }
  return RT_OK;
}
