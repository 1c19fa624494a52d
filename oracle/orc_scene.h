// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Scene layer: triangles (rc/shapes/mesh.rs, rc/shapes/mod.rs), GeometricPrimitive
// (rc/primitive.rs), SAH BVH build + traversal (rc/bvh/mod.rs, rc/bounds.rs), interactions
// (rc/interaction.rs, rc/ray.rs), lights (rc/light/*.rs), textures (rc/texture/*.rs, rc/mipmap.rs),
// materials (rc/material/*.rs) and light distributions (rc/lightdistrib.rs, rc/sampling/*).
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>
#include "orc_bsdf.h"
#include "orc_math.h"
#include "orc_sampler.h"

namespace orc {

// ---------------------------------------------------------------- rays / interactions
struct Ray {  // rc/ray.rs:10-15
  V3 o, d;
  float t_max = kInf;
  bool has_diff = false;
  V3 rx_o, ry_o, rx_d, ry_d;
};
inline Ray ray_new(V3 o, V3 d) { Ray r; r.o = o; r.d = d; r.t_max = kInf; r.has_diff = false; return r; }      // :18-28
inline Ray ray_segment(V3 o, V3 d, float tmax) { Ray r = ray_new(o, d); r.t_max = tmax; return r; }            // :30-40

struct Interaction {  // rc/interaction.rs:17-26
  V3 p, p_error, wo, n;
};
inline Interaction interaction_new(V3 p, V3 p_error, V3 wo, V3 n) { return {p, p_error, normalize(wo), n}; }  // :38-45 (normalizes wo)
inline Ray spawn_ray(const Interaction& it, V3 dir) { return ray_new(offset_ray_origin(it.p, it.p_error, it.n, dir), dir); }  // :56-60
inline Ray spawn_ray_to_interaction(const Interaction& a, const Interaction& b) {  // :69-74
  V3 origin = offset_ray_origin(a.p, a.p_error, a.n, b.p - a.p);
  V3 target = offset_ray_origin(b.p, b.p_error, b.n, origin - b.p);
  V3 d = target - origin;
  return ray_segment(origin, d, 1.0f - 1e-4f);
}
}  // namespace orc
#include "orc_sphere.h"  // needs Ray and Interaction
namespace orc {

struct SurfaceInteraction {  // rc/interaction.rs:78-104
  Interaction hit;
  P2 uv;
  V3 dpdu, dpdv;
  V3 dpdx{0, 0, 0}, dpdy{0, 0, 0};
  float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
  int prim = -1;  // index into ordered primitive list
  int sub = -1;   // hit inside an object instance (prim names the TransformedPrimitive): index into the OBJECT's ordered primitive list
  struct { V3 n, dpdu, dpdv; } shading;
};

// rc/interaction.rs:245-314
inline void compute_differential(SurfaceInteraction& si, const Ray& ray) {
  si.dudx = si.dudy = si.dvdx = si.dvdy = 0.0f;
  si.dpdx = si.dpdy = v3(0, 0, 0);
  if (!ray.has_diff) return;
  const V3 n = si.hit.n, p = si.hit.p;
  float d = dot(n, v3(p.x, p.y, p.z));
  float tx = -(dot(n, ray.rx_o) - d) / dot(n, ray.rx_d);
  float ty = -(dot(n, ray.ry_o) - d) / dot(n, ray.ry_d);
  if (std::isinf(tx) || tx != tx || std::isinf(ty) || ty != ty) return;
  V3 px = ray.rx_o + tx * ray.rx_d;
  V3 py = ray.ry_o + ty * ray.ry_d;
  si.dpdx = px - p;
  si.dpdy = py - p;
  int dim0, dim1;
  if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { dim0 = 1; dim1 = 2; }
  else if (fabsf(n.y) > fabsf(n.z)) { dim0 = 0; dim1 = 2; }
  else { dim0 = 0; dim1 = 1; }
  float A[2][2] = {{si.dpdu[dim0], si.dpdv[dim0]}, {si.dpdu[dim1], si.dpdv[dim1]}};
  float Bx0 = px[dim0] - p[dim0], Bx1 = px[dim1] - p[dim1];
  float By0 = py[dim0] - p[dim0], By1 = py[dim1] - p[dim1];
  float a, b;
  if (solve_linear_system2x2(A, Bx0, Bx1, &a, &b)) { si.dudx = a; si.dvdx = b; } else { si.dudx = 0; si.dvdx = 0; }
  if (solve_linear_system2x2(A, By0, By1, &a, &b)) { si.dudy = a; si.dvdy = b; } else { si.dudy = 0; si.dvdy = 0; }
}

// ---------------------------------------------------------------- textures
struct MipMap;  // orc_mipmap.h
enum TexKind { TEX_CONST = 0, TEX_SCALE = 1, TEX_MIX = 2, TEX_IMAGE = 3, TEX_CHECKER = 4, TEX_UV = 5, TEX_FBM = 6 };
struct Texture {
  int kind = TEX_CONST;
  RGB value{0, 0, 0};          // constant (float textures use .r)   rc/texture/constant.rs:35-38
  int tex1 = -1, tex2 = -1, amount = -1;  // scale / mix             rc/texture/scale.rs:23-25, mix.rs:24
  int mip = -1;                // imagemap                           rc/texture/imagemap.rs:232-235
  float su = 1, sv = 1, du = 0, dv = 0;   // UVMapping2D             rc/texture/mod.rs:38-61
  // checkerboard (rc/texture/checkerboard.rs): tex1, tex2, UV mapping, amount = AAMethod (0 None, 1 ClosedForm)
  // uv (rc/texture/uv.rs): UV mapping.  fbm (rc/texture/fbm.rs): value.r = omega, amount = octaves, identity texture space
  bool is_float = false;
};

float noise_perlin(float x, float y, float z);                                   // rc/noise.rs:8-43
float noise_fbm(V3 p, V3 dpdx, V3 dpdy, float omega, uint32_t max_octaves);     // rc/noise.rs:46-66

// ---------------------------------------------------------------- materials
enum MatKind { MAT_MATTE = 0, MAT_PLASTIC, MAT_METAL, MAT_MIRROR, MAT_GLASS, MAT_UBER, MAT_SUBSTRATE, MAT_MIX, MAT_TRANSLUCENT, MAT_DISNEY, MAT_NONE };
// MAT_DISNEY (rc/material/disney.rs) reuses the slots: kd = color, ks = metallic, eta = eta, roughness = roughness, kr = speculartint,
// urough = anisotropic, kt = sheen, sigma = sheentint, vrough = clearcoat, k = clearcoatgloss, opacity = spectrans,
// reflect = scatterdistance, transmit = flatness, amount = difftrans, m1 = thin (0 / 1, not a texture id)
struct Material {
  int kind = MAT_MATTE;
  // texture ids; meaning depends on kind (see build_bsdf)
  int kd = -1, ks = -1, kr = -1, kt = -1, sigma = -1, roughness = -1, urough = -1, vrough = -1;
  int eta = -1, k = -1, opacity = -1, reflect = -1, transmit = -1, amount = -1;
  int m1 = -1, m2 = -1;  // mix
  int bump = -1;         // "bumpmap" float texture (every material but mix), material/mod.rs:50-92
  bool remap_roughness = true;
};

// ---------------------------------------------------------------- lights
enum LightKind { LIGHT_DIFFUSE_AREA = 0, LIGHT_POINT = 1, LIGHT_DISTANT = 2, LIGHT_INFINITE = 3 };
struct Distribution1D {  // rc/sampling/distribution1d.rs
  std::vector<float> func, cdf;
  float func_int = 0;
  Distribution1D() {}
  explicit Distribution1D(const float* f, size_t n) { init(f, n); }
  void init(const float* f, size_t n) {  // :11-42
    func.assign(f, f + n);
    cdf.assign(n + 1, 0.0f);
    for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
    func_int = cdf[n];
    if (func_int == 0.0f) for (size_t i = 1; i < n + 1; ++i) cdf[i] = (float)i / (float)n;
    else for (size_t i = 1; i < n + 1; ++i) cdf[i] /= func_int;
  }
  size_t count() const { return func.size(); }
  // :48-68 → (x, pdf, offset)
  void sample_continuous(float u, float* x, float* pdf, size_t* off) const {
    size_t offset = find_interval(cdf.size(), [&](size_t i) { return cdf[i] <= u; });
    float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
    *pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
    *x = ((float)offset + du) / (float)count();
    *off = offset;
  }
  // :70-79
  void sample_discrete(float u, size_t* off, float* pdf) const {
    size_t offset = find_interval(cdf.size(), [&](size_t i) { return cdf[i] <= u; });
    *pdf = func_int > 0.0f ? func[offset] / (func_int * (float)count()) : 0.0f;
    *off = offset;
  }
};
struct Distribution2D {  // rc/sampling/distribution2d.rs
  std::vector<Distribution1D> cond;
  Distribution1D marginal;
  void init(const float* func, size_t nu, size_t nv) {  // :11-27
    cond.resize(nv);
    std::vector<float> mf(nv);
    for (size_t v = 0; v < nv; ++v) { cond[v].init(func + v * nu, nu); mf[v] = cond[v].func_int; }
    marginal.init(mf.data(), nv);
  }
  P2 sample_continuous(P2 u, float* pdf) const {  // :29-34
    float d1, pdf1, d0, pdf0; size_t v, dummy;
    marginal.sample_continuous(u.y, &d1, &pdf1, &v);
    cond[v].sample_continuous(u.x, &d0, &pdf0, &dummy);
    *pdf = pdf0 * pdf1;
    return P2{d0, d1};
  }
  float pdf(P2 p) const {  // :36-49
    size_t nu = cond[0].count(), nv = marginal.count();
    size_t iu = clamp_t<size_t>((size_t)f2u_sat(p.x * (float)nu), 0, nu - 1);
    size_t iv = clamp_t<size_t>((size_t)f2u_sat(p.y * (float)nv), 0, nv - 1);
    return cond[iv].func[iu] / marginal.func_int;
  }
};

struct Light {
  int kind = LIGHT_DIFFUSE_AREA;
  // diffuse area (rc/light/diffuse.rs)
  int tri = -1;          // input-order triangle index
  RGB l_emit{0, 0, 0};
  bool two_sided = false;
  float area = 0;
  // point (rc/light/point.rs) / distant (rc/light/distant.rs)
  V3 pos{0, 0, 0};       // point: position; distant: normalized direction
  RGB intensity{0, 0, 0};
  V3 w_center{0, 0, 0}; float w_radius = 0;  // distant/infinite preprocess
  // infinite (rc/light/infinite.rs)
  int mip = -1;
  M44 l2w = m44_identity(), w2l = m44_identity();
  std::shared_ptr<Distribution2D> distribution;
};

// ---------------------------------------------------------------- BVH node (rc/bvh/mod.rs:582-598)
struct LinearNode {
  B3 bounds;
  uint32_t offset;   // leaf: primitives_offset ; interior: second_child_offset
  uint16_t n_prims;  // 0 => interior
  uint8_t axis;
  uint8_t pad;
};

struct TraceCounters {
  uint64_t rays_closest = 0, rays_any = 0, nodes = 0, tris = 0, tri_hits = 0;
  void add(const TraceCounters& o) { rays_closest += o.rays_closest; rays_any += o.rays_any; nodes += o.nodes; tris += o.tris; tri_hits += o.tri_hits; }
};

struct TriHit { float t, b0, b1, b2; };

struct Scene {
  // ---- input geometry: triangle soup (world space, as TriangleMesh::new leaves it, mesh.rs:61)
  std::vector<V3> P, N, S;
  std::vector<P2> UV;
  std::vector<int32_t> idx;         // 3 per tri
  std::vector<int32_t> tri_material;
  std::vector<int32_t> tri_light;   // index into lights or -1
  std::vector<uint8_t> tri_flags;   // bit0: reverse_orientation ^ swaps_handedness; bit1: mesh has N; bit2: has UV; bit3: has S
  std::vector<int32_t> tri_alpha;   // 2 per triangle: float-texture ids of mesh.alpha_mask / mesh.shadow_alpha_mask or -1 (mesh.rs:38-39); empty = none
  // ---- analytic spheres (rc/shapes/sphere.rs): primitive ids n_tris() .. n_prims() - 1; tri_material / tri_light / tri_flags carry their entries too
  // (bit0 of the flags: reverse_orientation ^ swaps_handedness, as for triangles)
  std::vector<Sphere> spheres;
  // ---- object instances (TransformedPrimitive, rc/primitive.rs:79-118; api.rs:1053-1090): primitive ids after the spheres. An object is a triangle
  // mesh in OBJECT space with its own BVH (the aggregate object_instance builds over more than one primitive; a single primitive is wrapped as it is);
  // its tri_material entries index THIS scene's materials; its `lights` are the area lights of emitting shapes inside the definition, which no light list
  // holds (api.rs:954-964): looked up by isect_le, never sampled.
  struct Instance { int object; M44 o2w, w2o; };
  std::vector<std::shared_ptr<Scene>> objects;
  std::vector<Instance> instances;
  size_t n_prims() const { return n_tris() + spheres.size() + instances.size(); }
  bool is_sphere(int prim) const { return (size_t)prim >= n_tris() && (size_t)prim < n_tris() + spheres.size(); }
  bool is_instance(int prim) const { return (size_t)prim >= n_tris() + spheres.size(); }
  const Instance& instance_of(int prim) const { return instances[(size_t)prim - n_tris() - spheres.size()]; }
  // the scene and input-order primitive id that carry a hit's material / flags: the object's for a hit inside an instance
  const Scene& owner_of(const SurfaceInteraction& si, int* tri) const {
    const int top = ordered[si.prim];
    if (!is_instance(top)) { *tri = top; return *this; }
    const Scene& o = *objects[instance_of(top).object];
    *tri = o.ordered[si.sub];
    return o;
  }
  int material_of(const SurfaceInteraction& si) const { int tri; const Scene& o = owner_of(si, &tri); return o.tri_material[tri]; }
  // The id a hit record carries (rt_trace_closest, include/rtx_hip.h): the leaf-order index of a top-level primitive, or - inside instance k (in
  // add_instance order) - n_prims() + (primitives of the objects of instances 0 .. k-1) + the object's leaf-order index
  int64_t hit_id(int prim, int sub_) const {
    const int top = ordered[prim];
    if (!is_instance(top)) return prim;
    const size_t k = (size_t)top - n_tris() - spheres.size();
    int64_t base = (int64_t)n_prims();
    for (size_t j = 0; j < k; ++j) base += (int64_t)objects[instances[j].object]->n_prims();
    return base + sub_;
  }
  const Sphere& sphere_of(int prim) const { return spheres[(size_t)prim - n_tris()]; }
  float shape_area(int prim) const { return is_sphere(prim) ? quadric_area(sphere_of(prim)) : tri_area(prim); }
  // Shape::intersect of primitive `prim`, hit test only: triangles give barycentrics, spheres t alone
  bool prim_test(int prim, const Ray& ray, TriHit* h) const {
    if (!is_sphere(prim)) return tri_test(prim, ray, h);
    SphereHit sh;
    if (!quadric_intersect(sphere_of(prim), ray, false, &sh)) return false;
    h->t = sh.t; h->b0 = h->b1 = h->b2 = 0.0f;
    return true;
  }
  void prim_fill_interaction(int prim, const Ray& ray, const TriHit& h, SurfaceInteraction* si) const;
  std::vector<Texture> textures;
  std::vector<std::shared_ptr<MipMap>> mips;
  std::vector<Material> materials;
  std::vector<Light> lights;
  std::vector<int> infinite_lights;
  // ---- BVH
  int max_prims_per_node = 4;
  std::vector<LinearNode> nodes;
  std::vector<int32_t> ordered;  // ordered_prims: leaf order -> input triangle index
  B3 world_bounds() const { return nodes.empty() ? b3_empty() : nodes[0].bounds; }

  size_t n_tris() const { return idx.size() / 3; }
  void tri_verts(int tri, V3* p0, V3* p1, V3* p2) const { *p0 = P[idx[3 * tri]]; *p1 = P[idx[3 * tri + 1]]; *p2 = P[idx[3 * tri + 2]]; }
  void tri_uvs(int tri, P2 uv[3]) const {  // mesh.rs:201-211
    if (tri_flags[tri] & 4) { uv[0] = UV[idx[3 * tri]]; uv[1] = UV[idx[3 * tri + 1]]; uv[2] = UV[idx[3 * tri + 2]]; }
    else { uv[0] = P2{0, 0}; uv[1] = P2{1, 0}; uv[2] = P2{1, 1}; }
  }
  B3 tri_world_bounds(int tri) const {  // mesh.rs:603-608
    V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
    return b3_union_p(b3_from_points(p0, p1), p2);
  }
  float tri_area(int tri) const {  // mesh.rs:588-594
    V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
    return 0.5f * length(cross(p1 - p0, p2 - p0));
  }

  // ---- build / traversal (orc_scene.cpp)
  void build_bvh();
  bool tri_test(int tri, const Ray& ray, TriHit* h) const;          // mesh.rs:215-319 / 428-532 (shared hit test)
  bool tri_alpha_rejects(int tri, const Ray& ray, const TriHit& h, bool shadow_ray) const;  // mesh.rs:353-370 (intersect), 534-582 (intersect_p)
  void tri_fill_interaction(int tri, const Ray& ray, const TriHit& h, SurfaceInteraction* si) const;  // mesh.rs:321-425
  bool intersect_raw(Ray& ray, int* prim_out, TriHit* hit_out, TraceCounters* tc, int* sub_out = nullptr) const;  // bvh/mod.rs:366-433
  bool object_intersect_raw(Ray& ray, int* prim_out, TriHit* hit_out, TraceCounters* tc) const;  // the primitive a TransformedPrimitive wraps: the BVH, or the single primitive
  bool object_intersect_p(const Ray& ray, TraceCounters* tc) const;
  bool intersect(Ray& ray, SurfaceInteraction* si, TraceCounters* tc) const;
  bool intersect_p(const Ray& ray, TraceCounters* tc) const;        // bvh/mod.rs:435-501
  // ---- lights
  void preprocess_lights();
  RGB area_light_l(const Light& l, const Interaction& it, V3 w) const {  // diffuse.rs:91-97
    if (l.two_sided || dot(it.n, w) > 0.0f) return l.l_emit;
    return rgb(0, 0, 0);
  }
  RGB isect_le(const SurfaceInteraction& si, V3 w) const {  // interaction.rs:149-154
    int tri = ordered[si.prim];
    if (is_instance(tri)) {  // TransformedPrimitive hands the object's hit on: its primitive's area light, if any (never one of THIS scene's lights)
      const Scene& o = *objects[instance_of(tri).object];
      const int oli = o.tri_light[o.ordered[si.sub]];
      return oli < 0 ? rgb(0, 0, 0) : area_light_l(o.lights[oli], si.hit, w);
    }
    int li = tri_light[tri];
    if (li < 0) return rgb(0, 0, 0);
    return area_light_l(lights[li], si.hit, w);
  }
  void tri_sample(int tri, P2 u, Interaction* it, float* pdf) const;              // mesh.rs:610-634
  void shape_sample_si(int tri, const Interaction& ref, P2 u, Interaction* it, float* pdf) const;  // shapes/mod.rs:39-53
  float shape_pdf_wi(int tri, const Interaction& ref, V3 wi, TraceCounters* tc) const;             // shapes/mod.rs:59-68
  // Light trait
  struct LiSample { RGB li; V3 wi; float pdf; Interaction p0, p1; };
  LiSample light_sample_li(const Light& l, const Interaction& it, P2 u) const;
  float light_pdf_li(const Light& l, const Interaction& it, V3 wi, TraceCounters* tc) const;
  RGB light_le(const Light& l, const Ray& ray) const;
  static bool is_delta(const Light& l) { return l.kind == LIGHT_POINT || l.kind == LIGHT_DISTANT; }  // light/mod.rs:38-40
  // ---- textures / materials
  RGB tex_eval(int id, const SurfaceInteraction& si) const;
  float tex_eval_f(int id, const SurfaceInteraction& si) const { return tex_eval(id, si).r; }
  void build_bsdf(int mat, SurfaceInteraction& si, Bsdf* bsdf, int depth = 0) const;  // may bump-map si's shading geometry
  void bump(int tex, SurfaceInteraction& si) const;
};

// ---------------------------------------------------------------- light distributions (rc/lightdistrib.rs)
struct LightDistribution {
  const Scene* scene = nullptr;
  bool uniform = true;
  Distribution1D uniform_distrib;
  uint32_t n_voxels[3] = {1, 1, 1};
  // Dense, lazily filled table instead of the reference's lock-free hash (:183-297). The
  // distribution is a pure function of the voxel index, so the result is identical.
  std::vector<std::atomic<Distribution1D*>> table;
  std::mutex mtx;
  ~LightDistribution() { for (auto& p : table) delete p.load(); }
  void init(const Scene* s, const char* strategy, uint32_t max_voxels = 64);  // path.rs:86-94
  void voxel_of(V3 p, int pi[3]) const;                                       // :187-198
  Distribution1D* compute_distribution(const int pi[3]) const;                // :101-179
  const Distribution1D* lookup(V3 p);
};

}  // namespace orc
