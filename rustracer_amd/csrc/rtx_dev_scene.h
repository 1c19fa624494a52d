// rtx_dev_scene.h — device scene layout in HBM, watertight triangle test, BVH traversal and
// SurfaceInteraction construction. Restates rc/bvh/mod.rs:366-501, rc/bounds.rs:127-157,
// rc/shapes/mesh.rs:215-426,428-586,610-634, rc/interaction.rs, rc/shapes/mod.rs:39-68.
#pragma once
#include "rtx_dev_math.h"

namespace rtx {

// ---- HBM layout (DESIGN.md "Data layout") -------------------------------------------------
// nodes : n_nodes x 32 B  = 2 x float4 {bmin.xyz, bmax.x} {bmax.yz, offset, n_prims|axis<<16}
// tri_p : n_tris  x 48 B  = 3 x float4 {p0.xyz, material} {p1.xyz, light} {p2.xyz, flags}   (leaf order;
//         the w lanes carry rt_tri_meta so that shading needs no second gather; the algorithmic
//         size of a triangle stays the 36 B of its nine coordinates)
struct DTexture { int kind; float v[3]; int tex1, tex2, amount, image; float su, sv, du, dv; };
struct DImage {  // levels of a MIPMap: float4 texels in tiles of 4 x 2 (one 128-byte line), 2^tshift tiles per tile row; off = first texel of a level
  int n_levels; int w[16], h[16]; unsigned long long off[16];
  const float4* texels; int trilinear; float max_aniso; int wrap; int tshift[16];
};
struct DMaterial { int kind; int slot[16]; int remap; int bump; int code_class; };  // code_class: materials that run the same shading code share one (host: material_code_classes)
struct DLight {
  int kind; int prim; float rgb[3]; int two_sided; float vec[3]; float area; float world_radius; int image;
  float l2w[12], w2l[12];
  // Distribution2D of an infinite light (distribution2d.rs:11-50). The conditional rows are stored as (cdf[i], func[i]) PAIRS, nu + 1 per row (the last
  // func is padding): the entries sample_continuous reads last - cdf[o], cdf[o + 1], func[o] - and the func[v][u] of Distribution2D::pdf sit in the sector the
  // bracketed search has just fetched, one dependent round trip to a random row less per sample.
  int nu, nv; const float* cf; const float* func_int; const float* mfunc; const float* mcdf; float mfunc_int;
  // guide tables of the CDF searches (rt_scene_create): guide[row][k] = the number of entries of the row's CDF that are <= k / 2^glog, k = 0 .. 2^glog - so
  // the bisection for u starts inside [guide[k], guide[k + 1]], k = floor(u 2^glog), ~16 entries instead of 2049 (same answer: the CDF is non-decreasing)
  const unsigned short* guide; const unsigned short* mguide; int glog, mglog;
  // constants of a triangle emitter, evaluated once on the device (k_light_consts) by the expressions DiffuseAreaLight::sample_li's call chain uses per sample:
  // normalize(cross(p1 - p0, p2 - p0)) (Triangle::sample, mesh.rs:617, before orientation) and 1 / area (shapes/mod.rs:45)
  float nrm[3]; float inv_area;
};
struct DSphere;  // rtx_dev_sphere.h
// An object instance (rt_instance): primitive_to_world and its inverse, where the object's tree and primitives sit in the scene's arrays, and the first
// hit id of this instance (n_top_prims + the primitives of the instances before it)
struct DInstance { float o2w[16], w2o[16]; unsigned node_base, n_nodes, prim_base, n_prims, id_base, pad[3]; };
struct DScene {
  const float4* nodes; unsigned n_nodes;
  const float4* pairs;  // n_nodes x 64 B child-pair records of the interior nodes (NULL for LDS-resident scenes), see k_trace_pair
  int obj_pairs;        // ... the objects' trees have records too (behind the top level's, child codes local to the object): nested_pair_walk
  int obj_general;      // some instanced object holds a quadric or a masked triangle: objects are walked by the general one-node-per-step walk (instance_intersect)
  const float4* top_pairs; unsigned n_top;  // pair records of the first levels of the tree, child codes re-pointed at LDS slots (k_trace_top); n_top <= RT_TOP_MAX
  const float4* quads;  // n_nodes x 128 B grandchild records of the interior nodes (NULL when not built), see k_trace_quad
  // LDS-resident scenes of plain triangles: [8][n_nodes] u16 - for each sign octant of a ray's direction (bit 0 / 1 / 2: d.x / d.y / d.z negative) the node the
  // reference's walk takes up once it is done with node i and everything below it (n_nodes: the walk is over). With these links the walk needs no stack
  // (closest_small, occluded_small in rtx_kernels.h); built by rt_scene_create from the nodes
  const unsigned short* skip8;
  // round 5: the walks' links as one word per (octant, node): (where the walk goes when the node's box passes and it is interior) << 16 | (where it carries on after the node and
  // everything below it), then 8 start nodes. link8: over the nodes the walks TEST (interior nodes whose test rarely fails are passed over, rt_scene_create); link8_full: over all nodes
  const unsigned* link8; const unsigned* link8_full;
  const float4* tri_p; unsigned n_tris;
  const unsigned short* prim_class;  // [n_tris] the code class of the primitive's material (DMaterial::code_class), bit 15: a quadric - the key of the vertex queue's binning
  const float4* tri_rec;  // per-triangle shade records (8 x float4, see tri_fill_interaction_inl), built on the device at rt_scene_create
  const float* tri_n; const float* tri_uv; const float* tri_s;
  const int2* tri_alpha;  // {alpha, shadowalpha} texture ids of the triangles whose flags carry bit 4 / bit 5 (NULL: no mask in the scene)
  const DInstance* instances; unsigned n_instances, n_top_prims;  // object instances (RT_FLAG_INSTANCE primitives); primitives [0, n_top_prims) are the top level's
  const DSphere* spheres; // analytic spheres: a primitive whose flags carry bit 6 (RT_PRIM_SPHERE) holds its world box in p0 / p1 and its index as the bits of p2.x
  const DTexture* textures; const DImage* images; const DMaterial* materials; const DLight* lights;
  int n_lights; int n_infinite; int infinite_ids[4];
  int n_lights_all;  // sampled lights + the emitters no light list holds: records in `lights`
  int n_materials, n_textures, n_images;
  int needs_differentials;  // some texture reads dudx.. / dpdx.. (image maps, closed-form checkerboards, fbm)
  f3 wb_min, wb_max;
  // light distribution (rc/lightdistrib.rs): dense voxel table or a single uniform distribution
  int ld_uniform; int nvox[3];
  // tables of the BUILT voxels only (those that can hold a surface point): [slot][n_lights], [slot][n_lights+1], [slot]; ld_slot[voxel] = slot or -1.
  // Memory is O(built voxels x lights) - a dense 64^3 table of 10^5 emitters would not fit any GPU
  const float* ld_func; const float* ld_cdf; const float* ld_int; const int* ld_slot;
  const float4* ld_dense8;  // ... the same records indexed by VOXEL (an unbuilt voxel: func_int = -1): no slot to fetch first; NULL where the grid is too large for it
  const float4* ld_rows8;  // n_lights <= 3: [slot] {func_int, func[0..2]} {cdf[0..3]} - a voxel's whole distribution in two 16-byte loads in flight together (round 5); else NULL
  const unsigned short* ld_guide; int ld_glog;  // guide tables of the rows' CDF searches, [slot][2^ld_glog + 1] (see DLight::guide); ld_glog < 0: none (few lights)
  // This record in device memory. A kernel takes its DScene by value (fields in SGPRs); an out-of-line device function that wants the scene is handed
  // `*sc.self` instead of `sc`: a reference to the kernel argument would force a 300-byte private copy of it into every lane's scratch, read back
  // lane by lane at each use (k_shade<3> on S4 moved 4 KB per vertex between L2 and memory, most of it scratch).
  int route_quadric_hits;  // k_bin_count: vertices on analytic quadrics go to a bin of their own in the generic range (scenes shaded by the QLIGHTS forms of k_shade)
  const DScene* self;
#ifdef RT_ABLATE
  int dbg;  // measurement builds only (make ABLATE=1): bits switch parts of the shade kernel off to see what they cost; images are wrong
#endif
};
#ifdef RT_ABLATE
#define RT_DBG(sc, bit) (((sc).dbg & (bit)) != 0)
#else
#define RT_DBG(sc, bit) false
#endif

struct Ray { f3 o, d; float t_max; };
struct TriHit { float t, b0, b1, b2; };

// Shared hit test of Triangle::intersect / intersect_p (mesh.rs:215-319, 428-532).
RT_DEV bool tri_test(f3 p0, f3 p1, f3 p2, const Ray& ray, TriHit& h);

// Ray-only part of the watertight test (mesh.rs:229-247): permutation and shear constants depend on the ray
// alone, so the traversal computes them once per ray instead of once per triangle (same values, same order).
struct RayPre { int kx, ky, kz; float sx, sy, sz; };
RT_DEV RayPre ray_pre(const Ray& ray) {
  RayPre r;
  r.kz = max_dimension(abs3(ray.d));
  r.kx = r.kz + 1; if (r.kx == 3) r.kx = 0;
  r.ky = r.kx + 1; if (r.ky == 3) r.ky = 0;
  f3 d = permute(ray.d, r.kx, r.ky, r.kz);
  r.sx = -d.x / d.z; r.sy = -d.y / d.z; r.sz = 1.0f / d.z;
  return r;
}
// ... from the translated, permuted vertices on (mesh.rs:248-319): shear, edge functions, the rejection tests, t and the barycentrics, the error bound on t
// EARLY_SIGN: leave at the edge-sign test when every lane that runs the test fails it (the occlusion walk of an LDS-resident scene tests a primitive for the
// two or three lanes that hold a leaf in a given step; most of them miss the triangle outright)
template <bool EARLY_SIGN = false>
RT_DEV bool tri_test_permuted(f3 p0t, f3 p1t, f3 p2t, float sx, float sy, float sz, float t_max, TriHit& h);
RT_DEV bool tri_test_pre(f3 p0, f3 p1, f3 p2, const Ray& ray, const RayPre& rp, TriHit& h) {
  f3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
  p0t = permute(p0t, rp.kx, rp.ky, rp.kz); p1t = permute(p1t, rp.kx, rp.ky, rp.kz); p2t = permute(p2t, rp.kx, rp.ky, rp.kz);
  return tri_test_permuted(p0t, p1t, p2t, rp.sx, rp.sy, rp.sz, ray.t_max, h);
}
template <bool EARLY_SIGN>
RT_DEV bool tri_test_permuted(f3 p0t, f3 p1t, f3 p2t, const float sx, const float sy, const float sz, const float t_max, TriHit& h) {
  p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
  p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
  p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
  float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
  float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
  float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
  if (__builtin_expect(e0 == 0.0f || e1 == 0.0f || e2 == 0.0f, 0)) {  // cold f64 path, mesh.rs:260-270
    double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
    e0 = (float)(p2typ1tx - p2txp1ty);
    double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
    e1 = (float)(p0typ2tx - p0txp2ty);
    double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
    e2 = (float)(p1typ0tx - p1txp0ty);
  }
  // The reference's three rejection tests (edge signs, det == 0, t range; mesh.rs:272-296) are pure comparisons: they are
  // evaluated together and leave through one branch instead of three.
  const bool sign_fail = ((e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f)) & ((e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f));
  if (EARLY_SIGN && sign_fail) return false;
  float det = e0 + e1 + e2;
  p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
  float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  const float tmd = t_max * det;
  const bool range_fail = ((det < 0.0f) & ((t_scaled >= 0.0f) | (t_scaled < tmd))) | ((det > 0.0f) & ((t_scaled <= 0.0f) | (t_scaled > tmd)));
  if (sign_fail | (det == 0.0f) | range_fail) return false;
  float inv_det = 1.0f / det;
  float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
  float t = t_scaled * inv_det;
  float maxzt = max_component(abs3(mk3(p0t.z, p1t.z, p2t.z)));
  float delta_z = gamma_n(3) * maxzt;
  float maxxt = max_component(abs3(mk3(p0t.x, p1t.x, p2t.x)));
  float maxyt = max_component(abs3(mk3(p0t.y, p1t.y, p2t.y)));
  float delta_x = gamma_n(5) * (maxxt + maxzt);
  float delta_y = gamma_n(5) * (maxyt + maxzt);
  float delta_e = 2.0f * (gamma_n(2) * maxxt * maxyt + delta_y * maxxt + delta_x * maxyt);
  float max_e = max_component(abs3(mk3(e0, e1, e2)));
  float delta_t = 3.0f * (gamma_n(3) * max_e * maxzt + delta_e * maxzt + delta_z * max_e) * fabsf(inv_det);
  if (t <= delta_t) return false;
  h.t = t; h.b0 = b0; h.b1 = b1; h.b2 = b2;
  return true;
}

RT_DEV bool tri_test(f3 p0, f3 p1, f3 p2, const Ray& ray, TriHit& h) { RayPre rp = ray_pre(ray); return tri_test_pre(p0, p1, p2, ray, rp, h); }
RT_DEVN bool tri_test_call(f3 p0, f3 p1, f3 p2, const Ray& ray, TriHit& h) { return tri_test(p0, p1, p2, ray, h); }

// Bounds3::intersect_p_fast (bounds.rs:127-157): no (1 + 2 gamma3) widening, as in the reference.
RT_DEV bool slab_test(float4 n0, float4 n1, const Ray& ray, f3 inv_dir, int neg_x, int neg_y, int neg_z) {
  // n0 = {min.x, min.y, min.z, max.x}, n1 = {max.y, max.z, ..}
  // The reference leaves at the first failing clause; every clause is a pure comparison, so evaluating all of them and
  // combining the predicates gives the same answer without the divergent branches (which cost more than the arithmetic).
  float bx0 = neg_x ? n0.w : n0.x, bx1 = neg_x ? n0.x : n0.w;
  float by0 = neg_y ? n1.x : n0.y, by1 = neg_y ? n0.y : n1.x;
  float bz0 = neg_z ? n1.y : n0.z, bz1 = neg_z ? n0.z : n1.y;
  float tmin = (bx0 - ray.o.x) * inv_dir.x;
  float tmax = (bx1 - ray.o.x) * inv_dir.x;
  float tymin = (by0 - ray.o.y) * inv_dir.y;
  float tymax = (by1 - ray.o.y) * inv_dir.y;
  const bool miss_xy = (tmin > tymax) | (tymin > tmax);
  tmin = tymin > tmin ? tymin : tmin;
  tmax = tymax < tmax ? tymax : tmax;
  float tzmin = (bz0 - ray.o.z) * inv_dir.z;
  float tzmax = (bz1 - ray.o.z) * inv_dir.z;
  const bool miss_z = (tmin > tzmax) | (tzmin > tmax);
  tmin = tzmin > tmin ? tzmin : tmin;
  tmax = tzmax < tmax ? tzmax : tmax;
  return !miss_xy & !miss_z & (tmin < ray.t_max) & (tmax > 0.0f);
}

// The same decision for a ray whose reciprocal direction has no infinite component ("finite ray": no component of d is zero or so small that 1 / d
// overflows) - all but a few hundred rays of a frame. The only way a product (bound - o) * inv_dir can be NaN is 0 * inf, so for a finite ray every t is an
// ordered value (possibly +-inf by overflow) and the reference's sign selects and compare-selects are plain minima and maxima:
//   * per axis the near / far pick by the direction's sign is min / max of the two products (multiplying by a positive number keeps the order of
//     min <= max, by a negative one reverses it; rounding is monotone) - the SAME two products, so tmin / tmax carry the same bits (up to the sign of a zero,
//     which only ever meets comparisons);
//   * `tymin > tmin ? tymin : tmin` is max, `tymax < tmax ? tymax : tmax` is min, and the two early exits (tmin > tymax || tymin > tmax, then
//     tmin > tzmax || tzmin > tmax) reject exactly when the largest near value exceeds the smallest far value (each axis has near <= far).
// 17 VALU instructions instead of 33 (3 packed subtractions, 3 packed products, 6 min / max, max3, min3, 3 compares); no sign masks. A ray with an infinite
// reciprocal takes slab_test: there 0 * inf = NaN can occur (origin exactly on a bound of that axis) and the reference's selects are not minima.
RT_DEV bool inv_dir_finite(f3 inv_dir) { return (fabsf(inv_dir.x) < kInf) & (fabsf(inv_dir.y) < kInf) & (fabsf(inv_dir.z) < kInf); }
typedef float v2f __attribute__((ext_vector_type(2)));
// (origin and reciprocal BY VALUE: read through a reference to a lane-state struct, the scalar that feeds a two-wide operation is widened to a two-element load
// of the struct before inlining, and the struct then stays in scratch)
RT_DEV void slab_interval_finite(float4 n0, float4 n1, f3 o, f3 inv_dir, float& tmin, float& tmax) {
  // (min, max) of an axis as one two-wide value: v_pk_add_f32 / v_pk_mul_f32 do both bounds in one instruction each (the same IEEE operations)
  const v2f bx = {n0.x, n0.w}, by = {n0.y, n1.x}, bz = {n0.z, n1.y};
  const v2f tx = (bx - o.x) * inv_dir.x, ty = (by - o.y) * inv_dir.y, tz = (bz - o.z) * inv_dir.z;
  tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fminf(tz.x, tz.y));
  tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
}
RT_DEV void slab_interval_finite_scalar(float4 n0, float4 n1, f3 o, f3 inv_dir, float& tmin, float& tmax) {  // the same without two-wide values (k_trace_quad: four boxes of two-wide values leave the lane's state in scratch)
  const float tx0 = (n0.x - o.x) * inv_dir.x, tx1 = (n0.w - o.x) * inv_dir.x;
  const float ty0 = (n0.y - o.y) * inv_dir.y, ty1 = (n1.x - o.y) * inv_dir.y;
  const float tz0 = (n0.z - o.z) * inv_dir.z, tz1 = (n1.y - o.z) * inv_dir.z;
  tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fminf(tz0, tz1));
  tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fmaxf(tz0, tz1));
}
RT_DEV bool slab_test_finite(float4 n0, float4 n1, f3 o, float t_max, f3 inv_dir) {
  float tmin, tmax; slab_interval_finite(n0, n1, o, inv_dir, tmin, tmax);
  return (tmin <= tmax) & (tmin < t_max) & (tmax > 0.0f);
}
// FINITE: the caller has established inv_dir_finite for every lane that gets here
template <bool FINITE>
RT_DEV bool slab_test_t(float4 n0, float4 n1, const Ray& ray, f3 inv_dir, int neg_x, int neg_y, int neg_z) {
  if (FINITE) return slab_test_finite(n0, n1, ray.o, ray.t_max, inv_dir);
  return slab_test(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z);
}

RT_DEV void load_tri(const float4* tri_p, int prim, f3& p0, f3& p1, f3& p2) {
  float4 a = tri_p[3 * prim], b = tri_p[3 * prim + 1], c = tri_p[3 * prim + 2];
  p0 = mk3(a.x, a.y, a.z); p1 = mk3(b.x, b.y, b.z); p2 = mk3(c.x, c.y, c.z);
}
RT_DEV void load_tri_rec(const float4* tri_rec, int prim, f3& p0, f3& p1, f3& p2) {
  float4 a = tri_rec[8 * (size_t)prim], b = tri_rec[8 * (size_t)prim + 1], c = tri_rec[8 * (size_t)prim + 2];
  p0 = mk3(a.x, a.y, a.z); p1 = mk3(b.x, b.y, b.z); p2 = mk3(c.x, c.y, c.z);
}
RT_DEV int tri_material(const float4* tri_p, int prim) { return __float_as_int(tri_p[3 * prim].w); }
RT_DEV int tri_light(const float4* tri_p, int prim) { return __float_as_int(tri_p[3 * prim + 1].w); }
RT_DEV unsigned tri_flags(const float4* tri_p, int prim) { return __float_as_uint(tri_p[3 * prim + 2].w); }
#define RT_FLAG_SPHERE 64u
#define RT_FLAG_INSTANCE 128u
// the same three words read from the shade record (one line with everything else a vertex needs of its triangle)
RT_DEV int rec_material(const float4* tri_rec, int prim) { return __float_as_int(tri_rec[8 * (size_t)prim].w); }
RT_DEV int rec_light(const float4* tri_rec, int prim) { return __float_as_int(tri_rec[8 * (size_t)prim + 1].w); }
RT_DEV unsigned rec_flags(const float4* tri_rec, int prim) { return __float_as_uint(tri_rec[8 * (size_t)prim + 2].w); }
RT_DEV unsigned prim_sphere_index(const float4* tri_p, int prim) { return __float_as_uint(tri_p[3 * prim + 2].x); }

// ---------------------------------------------------------------- interactions (rc/interaction.rs)
struct Interaction { f3 p, p_error, wo, n; };
struct LightPoint { f3 p, p_error, n; };  // the Interaction a light sample returns, without the fields nothing reads (wo, time): LiSample stays within the 16 registers a call returns
struct SurfaceInteraction {
  Interaction hit;
  f2 uv;
  f3 dpdu, dpdv;
  float dudx, dvdx, dudy, dvdy;
  f3 dpdx, dpdy;
  f3 sh_n, sh_dpdu, sh_dpdv;
  f3 ssb;  // normalize(sh_dpdu): the first axis of Bsdf::new's frame (bsdf/mod.rs:77-91), set by the triangle fill (a constant of most triangles: tri_rec); the generic front-end recomputes it (bump maps rewrite sh_dpdu)
  int prim;
};
RT_DEV Ray spawn_ray(const Interaction& it, f3 dir) {  // :56-60
  Ray r; r.o = offset_ray_origin(it.p, it.p_error, it.n, dir); r.d = dir; r.t_max = kInf; return r;
}
template <class Target>
RT_DEV Ray spawn_ray_to_interaction(const Interaction& a, const Target& b) {  // :69-74
  Ray r;
  f3 origin = offset_ray_origin(a.p, a.p_error, a.n, b.p - a.p);
  f3 target = offset_ray_origin(b.p, b.p_error, b.n, origin - b.p);
  r.o = origin; r.d = target - origin; r.t_max = 1.0f - 1e-4f;
  return r;
}

// Triangle::intersect tail, mesh.rs:321-425, in three parts: what depends on the triangle alone (tri_geo), the shading frame (tri_frame: also a constant of
// the triangle unless the mesh carries per-vertex normals or tangents) and what depends on the hit (point, error bounds, uv, wo). The first two are evaluated
// once per triangle by k_tri_records into DScene::tri_rec - by these same functions, so a vertex shaded from the record sees the bits it would compute itself.
struct TriGeo { f3 dpdu, dpdv, n; };  // n = normalize(cross(dp02, dp12)), before orientation
RT_DEV TriGeo tri_geo(f3 p0, f3 p1, f3 p2, f2 uv0, f2 uv1, f2 uv2) {
  TriGeo g;
  f3 dpdu = mk3(0, 0, 0), dpdv = mk3(0, 0, 0);
  float duv02x = uv0.x - uv2.x, duv02y = uv0.y - uv2.y, duv12x = uv1.x - uv2.x, duv12y = uv1.y - uv2.y;
  f3 dp02 = p0 - p2, dp12 = p1 - p2;
  float determinant = duv02x * duv12y - duv02y * duv12x;
  bool degenerate_uv = fabsf(determinant) < 1e-8f;
  if (!degenerate_uv) {
    float inv_det = 1.0f / determinant;
    dpdu = (duv12y * dp02 - duv02y * dp12) * inv_det;
    dpdv = (-duv12x * dp02 + duv02x * dp12) * inv_det;
  }
  if (degenerate_uv || len2(cross(dpdu, dpdv)) == 0.0f) coordinate_system(normalize(cross(p2 - p0, p1 - p0)), dpdu, dpdv);
  g.dpdu = dpdu; g.dpdv = dpdv;
  g.n = normalize(cross(dp02, dp12));  // :385
  return g;
}
struct TriFrame { f3 n, sh_n, ss, ts; };  // hit.n, shading.n, shading.dpdu, shading.dpdv
// ns / ss_in: the interpolated, normalised vertex normal / tangent where the mesh has them (flags & 2, flags & 8)
RT_DEV TriFrame tri_frame(const TriGeo& g, unsigned flags, f3 ns_in, f3 ss_in) {
  TriFrame f;
  f.n = g.n;
  f3 ns = (flags & 2u) ? ns_in : g.n;
  f3 ss = (flags & 8u) ? ss_in : normalize(g.dpdu);
  f3 ts = cross(ss, ns);
  if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
  else coordinate_system(ns, ss, ts);
  f.sh_n = ns; f.ss = ss; f.ts = ts;
  if (flags & 2u) f.n = face_forward(f.n, f.sh_n);  // :417-422
  else if (flags & 1u) { f.n = -f.n; f.sh_n = f.n; }
  return f;
}
// Per-triangle shade record (DScene::tri_rec, 8 x float4 = one 128-byte line): {p0 | material} {p1 | light} {p2 | flags} - the traversal record - then
// {hit.n | ssb.x} {shading.dpdu | ssb.y} {shading.dpdv | ssb.z} {dpdu | 0} {dpdv | 0}; ssb = normalize(shading.dpdu), the first axis of Bsdf::new's frame
// (bsdf/mod.rs:77-91). The frame entries are those of a triangle WITHOUT per-vertex normals and tangents (RT_REC_CONST_FRAME); with them only dpdu / dpdv
// and n (row 3: g.n before orientation) are constants.
#define RT_REC_CONST_FRAME(flags) (((flags) & (2u | 8u)) == 0u)
// WO_SIGNS_ONLY: the caller's BSDF reads wo only through sign and zero tests of its dot products (the Lambert front-ends: same_hemisphere, the reflect /
// transmit side, wo.z == 0) - positive scaling does not change those, so the two normalisations of -ray_d (2 square roots, 6 divisions per vertex) are left out.
template <bool WO_SIGNS_ONLY = false>
RT_DEV void tri_fill_interaction_inl(const DScene& sc, int prim, f3 ray_d, const TriHit& h, SurfaceInteraction& si) {
  const float4* __restrict__ rec = sc.tri_rec + 8 * (size_t)prim;
  const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
  const f3 p0 = mk3(r0.x, r0.y, r0.z), p1 = mk3(r1.x, r1.y, r1.z), p2 = mk3(r2.x, r2.y, r2.z);
  const unsigned flags = __float_as_uint(r2.w);
  const float b0 = h.b0, b1 = h.b1, b2 = h.b2;
  f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);  // :201-211
  if ((flags & 4u) && !RT_DBG(sc, 32)) {  // (RT_DBG 32, measurement builds: what the shade stage would gain if the uvs came with the record's line)
    const float* u = sc.tri_uv + 6 * (size_t)prim;
    uv0 = mk2(u[0], u[1]); uv1 = mk2(u[2], u[3]); uv2 = mk2(u[4], u[5]);
  }
  float x_abs_sum = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
  float y_abs_sum = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
  float z_abs_sum = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
  si.hit.p_error = gamma_n(7) * mk3(x_abs_sum, y_abs_sum, z_abs_sum);
  si.hit.p = p0 * b0 + p1 * b1 + p2 * b2;
  si.uv = mk2(uv0.x * b0 + uv1.x * b1 + uv2.x * b2, uv0.y * b0 + uv1.y * b1 + uv2.y * b2);
  si.hit.wo = WO_SIGNS_ONLY ? -ray_d : normalize(normalize(-ray_d));  // SurfaceInteraction::new + Interaction::new both normalise (interaction.rs:42,123)
  const bool half_rec = RT_DBG(sc, 64);  // (RT_DBG 64, measurement builds: only the first 64 bytes of the record are read - what a 64-byte record would cost in traffic)
  const float4 r6 = half_rec ? r0 : rec[6], r7 = half_rec ? r1 : rec[7];
  si.dpdu = mk3(r6.x, r6.y, r6.z); si.dpdv = mk3(r7.x, r7.y, r7.z);
  if (RT_REC_CONST_FRAME(flags)) {
    const float4 r4 = half_rec ? make_float4(r3.y, r3.z, r3.x, r3.y) : rec[4], r5 = half_rec ? make_float4(r3.z, r3.x, r3.y, r3.z) : rec[5];
    si.hit.n = mk3(r3.x, r3.y, r3.z); si.sh_n = si.hit.n;
    si.sh_dpdu = mk3(r4.x, r4.y, r4.z); si.sh_dpdv = mk3(r5.x, r5.y, r5.z);
    si.ssb = mk3(r3.w, r4.w, r5.w);
  } else {
    TriGeo g; g.dpdu = si.dpdu; g.dpdv = si.dpdv; g.n = mk3(r3.x, r3.y, r3.z);
    f3 ns = g.n, ss = g.n;
    if (flags & 2u) {
      const float* q = sc.tri_n + 9 * (size_t)prim;
      ns = normalize(mk3(q[0], q[1], q[2]) * b0 + mk3(q[3], q[4], q[5]) * b1 + mk3(q[6], q[7], q[8]) * b2);
    }
    if (flags & 8u) {
      const float* q = sc.tri_s + 9 * (size_t)prim;
      ss = normalize(mk3(q[0], q[1], q[2]) * b0 + mk3(q[3], q[4], q[5]) * b1 + mk3(q[6], q[7], q[8]) * b2);
    }
    const TriFrame f = tri_frame(g, flags, ns, ss);
    si.hit.n = f.n; si.sh_n = f.sh_n; si.sh_dpdu = f.ss; si.sh_dpdv = f.ts;
    si.ssb = normalize(si.sh_dpdu);
  }
  si.dudx = si.dvdx = si.dudy = si.dvdy = 0.0f;
  si.dpdx = si.dpdy = mk3(0, 0, 0);
  si.prim = prim;
}

RT_DEVN void tri_fill_interaction(const DScene& sc, int prim, f3 ray_d, const TriHit& h, SurfaceInteraction& si) { tri_fill_interaction_inl(sc, prim, ray_d, h, si); }

// Geometric normal + hit point of a known hit, enough for Light::l / pdf_wi (no shading frame).
RT_DEV void tri_hit_point_normal_inl(const DScene& sc, int prim, const TriHit& h, f3& p, f3& n) {
  const float4* __restrict__ rec = sc.tri_rec + 8 * (size_t)prim;
  const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
  const unsigned flags = __float_as_uint(r2.w);
  p = mk3(r0.x, r0.y, r0.z) * h.b0 + mk3(r1.x, r1.y, r1.z) * h.b1 + mk3(r2.x, r2.y, r2.z) * h.b2;
  n = mk3(r3.x, r3.y, r3.z);  // normalize(cross(p0 - p2, p1 - p2)); already oriented (flags & 1) when the mesh has no vertex normals
  if (flags & 2u) {
    const float* q = sc.tri_n + 9 * (size_t)prim;
    f3 ns = normalize(mk3(q[0], q[1], q[2]) * h.b0 + mk3(q[3], q[4], q[5]) * h.b1 + mk3(q[6], q[7], q[8]) * h.b2);
    n = face_forward(n, ns);
  } else if (!RT_REC_CONST_FRAME(flags) && (flags & 1u)) n = -n;  // tangents without normals: row 3 holds the unoriented normal
}

RT_DEVN void tri_hit_point_normal(const DScene& sc, int prim, const TriHit& h, f3& p, f3& n) { tri_hit_point_normal_inl(sc, prim, h, p, n); }

// compute_differential (interaction.rs:245-314) for a ray that carries differentials. Inline: called out of line, `si` would have to live in scratch.
RT_DEV void compute_differential(SurfaceInteraction& si, f3 rx_o, f3 ry_o, f3 rx_d, f3 ry_d) {
  const f3 n = si.hit.n, p = si.hit.p;
  float d = dot(n, mk3(p.x, p.y, p.z));
  float tx = -(dot(n, rx_o) - d) / dot(n, rx_d);
  float ty = -(dot(n, ry_o) - d) / dot(n, ry_d);
  if (isinf(tx) || tx != tx || isinf(ty) || ty != ty) return;
  f3 px = rx_o + tx * rx_d, py = ry_o + ty * ry_d;
  si.dpdx = px - p; si.dpdy = py - p;  // interaction.rs:268-269
  int dim0, dim1;
  if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { dim0 = 1; dim1 = 2; }
  else if (fabsf(n.y) > fabsf(n.z)) { dim0 = 0; dim1 = 2; }
  else { dim0 = 0; dim1 = 1; }
  float A00 = comp(si.dpdu, dim0), A01 = comp(si.dpdv, dim0), A10 = comp(si.dpdu, dim1), A11 = comp(si.dpdv, dim1);
  float Bx0 = comp(px, dim0) - comp(p, dim0), Bx1 = comp(px, dim1) - comp(p, dim1);
  float By0 = comp(py, dim0) - comp(p, dim0), By1 = comp(py, dim1) - comp(p, dim1);
  float det = A00 * A11 - A01 * A10;  // solve_linear_system2x2, transform.rs:382-394
  if (fabsf(det) < 1e-10f) return;
  float x0 = (A11 * Bx0 - A01 * Bx1) / det, x1 = (A00 * Bx1 - A10 * Bx0) / det;
  if (!(x0 != x0 || x1 != x1)) { si.dudx = x0; si.dvdx = x1; }
  float y0 = (A11 * By0 - A01 * By1) / det, y1 = (A00 * By1 - A10 * By0) / det;
  if (!(y0 != y0 || y1 != y1)) { si.dudy = y0; si.dvdy = y1; }
}

RT_DEVN void compute_differential_call(SurfaceInteraction& si, f3 rx_o, f3 ry_o, f3 rx_d, f3 ry_d) { compute_differential(si, rx_o, ry_o, rx_d, ry_d); }  // generic shade kernel: its SurfaceInteraction lives in scratch anyway

// ---------------------------------------------------------------- BVH traversal
// One lane = one ray. The to-visit stack (64 entries, bvh/mod.rs:374) lives in LDS, laid out
// [depth][lane] so that a wave's push/pop of one depth touches 64 consecutive banks.
// NodeSrc abstracts where node/triangle records come from: HBM, or an LDS copy of a small scene.
struct GlobalSrc {
  const float4* nodes; const float4* tri_p;
  RT_DEV void node(int i, float4& a, float4& b) const { a = nodes[2 * i]; b = nodes[2 * i + 1]; }
  RT_DEV void tri(int i, f3& p0, f3& p1, f3& p2) const { load_tri(tri_p, i, p0, p1, p2); }
  RT_DEV void tri_flags(int i, f3& p0, f3& p1, f3& p2, unsigned& flags) const {
    float4 a = tri_p[3 * i], b = tri_p[3 * i + 1], c = tri_p[3 * i + 2];
    p0 = mk3(a.x, a.y, a.z); p1 = mk3(b.x, b.y, b.z); p2 = mk3(c.x, c.y, c.z); flags = __float_as_uint(c.w);
  }
};

// Closest-hit uses the "while-while" form: every lane first descends (slab tests, pushes) until it holds a
// leaf, then the wave tests leaf triangles together. Any-hit keeps the reference's single loop (it leaves at
// the first accepted triangle, so there is little leaf work to batch). In both, the per-ray sequence of node
// visits and triangle tests is exactly the reference's (counts and tie-breaking unchanged); only the SIMD
// interleaving differs.
// A primitive of a GENERAL scene that is not a plain triangle test: an analytic quadric (Sphere::intersect / intersect_p, the hit record carries t where a
// triangle's carries b2) or a triangle whose mesh has an alpha / shadow-alpha mask (an accepted test is dropped where the mask is 0, mesh.rs:353-370,
// 534-582). Object instances are not handled here (they need a stack: general_leaf in rtx_kernels.h). c2 = the third float4 of the primitive's record
// (p2 | flags). Defined in rtx_dev_shading.h; returns whether the ray hits, h filled.
struct DScene;
// GENERAL of the trace templates: 0 = plain triangles; 1 = every general primitive; 2 = general primitives of a scene WITHOUT alpha masks - the mask evaluator
// (a texture evaluation: 179 VGPRs) is then not instantiated and the kernels keep the quadric test's 134 (three waves per SIMD instead of two).
#define RT_GEN_ALL 1
#define RT_GEN_NO_MASKS 2
#define RT_GEN_INSTANCES_ONLY 3  // neither masks nor quadrics: object instances over plain triangles (133 / 99 VGPRs: the quadric test's 134 is not instantiated either)
template <bool MASKS, bool QUADRICS> RT_DEV bool general_prim_test(const DScene& sc_self, int prim, f3 p0, f3 p1, f3 p2, unsigned flags, const Ray& ray, const RayPre& rp, bool shadow_masks, TriHit& h);
#define RT_FLAG_GENERAL_TRI 48u  // alpha (16) | shadow alpha (32)

// GENERAL: leaves may hold quadrics and alpha-masked triangles (gen = the scene record in device memory and whether shadowalpha masks apply); the plain
// instantiation is the code it was.
struct GeneralCtx { const DScene* self; bool shadow_masks; int prim_base = 0; };  // prim_base: the walk's primitive 0 in the scene's arrays (an instanced object's walk)
template <int GENERAL, class Src>
RT_DEV bool leaf_prim_test(const Src& src, const GeneralCtx& gen, int prim, const Ray& ray, const RayPre& rp, TriHit& h) {
  f3 p0, p1, p2;
  if (GENERAL) {
    unsigned flags; src.tri_flags(prim, p0, p1, p2, flags);
    if (flags & (RT_FLAG_SPHERE | RT_FLAG_GENERAL_TRI)) return general_prim_test<GENERAL == RT_GEN_ALL, GENERAL != RT_GEN_INSTANCES_ONLY>(*gen.self, gen.prim_base + prim, p0, p1, p2, flags, ray, rp, gen.shadow_masks, h);
    return tri_test_pre(p0, p1, p2, ray, rp, h);
  }
  src.tri(prim, p0, p1, p2);
  return tri_test_pre(p0, p1, p2, ray, rp, h);
}
// FINITE: every lane's ray has a finite reciprocal direction (inv_dir_finite; the caller's wave-uniform choice) - the node test is slab_test_finite
template <bool ANY, bool COUNT, class Src, class StackT, int GENERAL = 0, bool FINITE = false>
RT_DEV bool traverse(const Src& src, Ray ray, StackT* stack, int stack_stride, int& prim_out, TriHit& hit_out, unsigned& n_nodes, unsigned& n_tris, GeneralCtx gen = GeneralCtx{nullptr, false}) {
  bool found = false;
  int sp = 0, cur = 0;
  f3 inv_dir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  const int neg_x = inv_dir.x < 0.0f, neg_y = inv_dir.y < 0.0f, neg_z = inv_dir.z < 0.0f;
  const RayPre rp = ray_pre(ray);
  if (ANY) {
    for (;;) {
      float4 n0, n1;
      src.node(cur, n0, n1);
      if (COUNT) n_nodes += 1;
      if (slab_test_t<FINITE>(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z)) {
        const unsigned packed = __float_as_uint(n1.w);
        const int n_prims = (int)(packed & 0xffffu);
        const int offset = __float_as_int(n1.z);
        if (n_prims > 0) {
          for (int i = 0; i < n_prims; ++i) {
            if (COUNT) n_tris += 1;
            TriHit h;
            if (leaf_prim_test<GENERAL>(src, gen, offset + i, ray, rp, h)) return true;
          }
          if (sp == 0) break;
          cur = (int)stack[(--sp) * stack_stride];
        } else {
          const int axis = (int)((packed >> 16) & 0xffu);
          const int neg = axis == 0 ? neg_x : (axis == 1 ? neg_y : neg_z);
          if (neg) { stack[(sp++) * stack_stride] = (StackT)(cur + 1); cur = offset; }
          else { stack[(sp++) * stack_stride] = (StackT)offset; cur = cur + 1; }
        }
      } else {
        if (sp == 0) break;
        cur = (int)stack[(--sp) * stack_stride];
      }
    }
    return false;
  }
  for (;;) {
    int leaf_off = 0, leaf_n = 0;
    bool done = false;
    for (;;) {  // descend to the next leaf whose box the ray enters
      float4 n0, n1;
      src.node(cur, n0, n1);
      if (COUNT) n_nodes += 1;
      if (slab_test_t<FINITE>(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z)) {
        const unsigned packed = __float_as_uint(n1.w);
        const int n_prims = (int)(packed & 0xffffu);
        const int offset = __float_as_int(n1.z);
        if (n_prims > 0) { leaf_off = offset; leaf_n = n_prims; break; }
        const int axis = (int)((packed >> 16) & 0xffu);
        const int neg = axis == 0 ? neg_x : (axis == 1 ? neg_y : neg_z);
        if (neg) { stack[(sp++) * stack_stride] = (StackT)(cur + 1); cur = offset; }
        else { stack[(sp++) * stack_stride] = (StackT)offset; cur = cur + 1; }
      } else {
        if (sp == 0) { done = true; break; }
        cur = (int)stack[(--sp) * stack_stride];
      }
    }
    if (done) break;
    for (int i = 0; i < leaf_n; ++i) {
      if (COUNT) n_tris += 1;
      TriHit h;
      if (leaf_prim_test<GENERAL>(src, gen, leaf_off + i, ray, rp, h)) {
        ray.t_max = h.t; found = true; prim_out = leaf_off + i; hit_out = h;  // `.or(result)`: later accepted hits replace
      }
    }
    if (sp == 0) break;
    cur = (int)stack[(--sp) * stack_stride];
  }
  return found;
}

// Leaf phase gating of the persistent traversal loops (k_trace_pair, k_trace_top, k_trace_quad). One iteration of those loops runs the interior step for the
// lanes at interior nodes and then the leaf step for the lanes at leaves, each at the full cost of its memory round trip however few lanes take it - and at
// any moment only ~15 % of the rays are at a leaf, so the leaf step ran every iteration for a handful of lanes. It is now held back until RT_LEAF_MIN lanes
// of the wave wait at a leaf, or no lane is left at an interior node. Per-ray steps and their order are untouched - only when they run: same hit records.
// Measured (threshold 1 / 8 / 16 / 24 / 32 / 48 / 64 = "only when every lane waits"): S2 frame 217.8 / 194.8 / 187.3 / 189.1 / 197.9 / 239.0 / 274.6 ms,
// S4 9461 / 8465 / 8134 / 7971 / 8009 / 8002 / 8380 ms, S3 448 / 432 / 419 / 411 / 411 / 430 / 443 ms. The LDS-resident kernel of S1 is not gated: as a
// per-lane state machine (which gating needs) its walk costs more VALU instructions than the while-while form saves (closest hit 357 -> 444 ms at best).
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 20
#endif
RT_DEV bool leaf_phase_now(bool active, bool at_leaf, unsigned leaf_min) {  // called by every executing lane of the wave; wave-uniform answer. leaf_min: a launch parameter (rt_render picks it per ray class and scene)
  if (leaf_min <= 1u) return true;
  const unsigned nl = (unsigned)__popcll(__ballot(active && at_leaf)), ni = (unsigned)__popcll(__ballot(active && !at_leaf));
  return nl >= leaf_min || ni == 0u;
}

// traverse() with the wave-level exit of its descend loop made explicit: the lanes walk down one node per round until LEAF_MIN of them hold a leaf or none
// is walking, then the holders test their triangles. LEAF_MIN = 64 is traverse()'s closest-hit form, LEAF_MIN = 1 its any-hit form. On the LDS-resident
// S1 (VALU-bound, no memory latency to amortise) closest hit takes 357 ms at 64, 349 / 344 / 345 at 12 / 16 / 20 and 380 at 40; shadow rays only lose
// (174 ms at 1, 193 - 217 at 8 - 40), so they stay on traverse()'s single loop.
#ifndef RT_LDS_LEAF_MIN_CLOSEST
#define RT_LDS_LEAF_MIN_CLOSEST 16
#endif
#ifndef RT_LDS_LEAF_MIN_ANY
#define RT_LDS_LEAF_MIN_ANY 1
#endif

}  // namespace rtx
