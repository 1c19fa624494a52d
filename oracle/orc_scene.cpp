// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
#include "orc_scene.h"
#include <algorithm>
#include <string>
#include "orc_mipmap.h"

namespace orc {

// ============================================================================ BVH build
// rc/bvh/mod.rs:80-358.  SplitMethod::SAH only (Middle has the `mid` bug, quirk 2).
namespace {
struct PrimInfo { int prim_number; V3 centroid; B3 bounds; };  // :522-536
struct BuildNode {
  B3 bounds; int axis = 0; int first_prim = 0, n_prims = 0;
  std::unique_ptr<BuildNode> child[2];
};
struct Bucket { int count = 0; B3 bounds = b3_empty(); };

// itertools 0.10.3 `partition` (two-cursor, unstable): swaps a failing front element with the
// next passing element found scanning from the back.
template <class Pred>
size_t it_partition(PrimInfo* a, size_t n, Pred pred) {
  size_t split = 0, front = 0, back = n;
  while (front < back) {
    PrimInfo* f = &a[front++];
    if (!pred(*f)) {
      bool found = false;
      while (front < back) {
        PrimInfo* b = &a[--back];
        if (pred(*b)) { std::swap(*f, *b); found = true; break; }
      }
      if (!found) return split;
    }
    split += 1;
  }
  return split;
}

struct Builder {
  const Scene* sc; std::vector<PrimInfo>& info; std::vector<int32_t>& ordered; int max_prims; size_t total_nodes = 0;
  std::unique_ptr<BuildNode> leaf(size_t start, size_t end, const B3& bounds) {
    auto n = std::make_unique<BuildNode>();
    n->bounds = bounds; n->first_prim = (int)ordered.size(); n->n_prims = (int)(end - start);
    for (size_t i = start; i < end; ++i) ordered.push_back(info[i].prim_number);
    return n;
  }
  std::unique_ptr<BuildNode> build(size_t start, size_t end) {  // :137-312
    total_nodes += 1;
    size_t n_primitives = end - start;
    B3 bounds = b3_empty();
    for (size_t i = start; i < end; ++i) bounds = b3_union(bounds, info[i].bounds);
    if (n_primitives == 1) return leaf(start, end, bounds);
    B3 cb = b3_empty();
    for (size_t i = start; i < end; ++i) cb = b3_union_p(cb, info[i].centroid);
    int dim = b3_maximum_extent(cb);
    if (cb.mn[dim] == cb.mx[dim]) return leaf(start, end, bounds);
    size_t mid;
    if (n_primitives <= 2) {  // :204-212
      mid = (start + end) / 2;
      if (start != end - 1 && info[end - 1].centroid[dim] < info[start].centroid[dim]) std::swap(info[start], info[end - 1]);
    } else {
      const int NB = 12;
      Bucket buckets[NB];
      auto bucket_of = [&](const PrimInfo& pi) {
        int b = (int)f2u_sat((float)NB * b3_offset(cb, pi.centroid)[dim]);
        if (b == NB) b = NB - 1;
        return b;
      };
      for (size_t i = start; i < end; ++i) {
        int b = bucket_of(info[i]);
        buckets[b].count += 1;
        buckets[b].bounds = b3_union(buckets[b].bounds, info[i].bounds);
      }
      float cost[NB - 1];
      for (int i = 0; i < NB - 1; ++i) {  // :233-251
        B3 b0 = b3_empty(), b1 = b3_empty();
        int c0 = 0, c1 = 0;
        for (int j = 0; j <= i; ++j) { b0 = b3_union(b0, buckets[j].bounds); c0 += buckets[j].count; }
        for (int j = i + 1; j < NB; ++j) { b1 = b3_union(b1, buckets[j].bounds); c1 += buckets[j].count; }
        cost[i] = 1.0f + ((float)c0 * b3_surface_area(b0) + (float)c1 * b3_surface_area(b1)) / b3_surface_area(bounds);
      }
      float min_cost = cost[0]; int min_b = 0;
      for (int i = 1; i < NB - 1; ++i) if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
      float leaf_cost = (float)n_primitives;
      if ((int)n_primitives > max_prims || min_cost < leaf_cost) {
        mid = start + it_partition(&info[start], end - start, [&](const PrimInfo& pi) { return bucket_of(pi) <= min_b; });
      } else {
        return leaf(start, end, bounds);
      }
    }
    auto n = std::make_unique<BuildNode>();
    // right subtree first (quirk 1, :290-309)
    auto right = build(mid, end);
    auto left = build(start, mid);
    n->axis = dim;
    n->bounds = b3_union(left->bounds, right->bounds);  // :553
    n->child[0] = std::move(left); n->child[1] = std::move(right);
    return n;
  }
};
size_t flatten(const BuildNode* node, std::vector<LinearNode>& out) {  // :314-358
  size_t offset = out.size();
  LinearNode ln{};
  ln.bounds = node->bounds;
  if (node->n_prims > 0) {
    ln.offset = (uint32_t)node->first_prim; ln.n_prims = (uint16_t)node->n_prims; ln.axis = 0;
    out.push_back(ln);
  } else {
    ln.offset = 0; ln.n_prims = 0; ln.axis = (uint8_t)node->axis;
    out.push_back(ln);
    flatten(node->child[0].get(), out);
    size_t second = flatten(node->child[1].get(), out);
    out[offset].offset = (uint32_t)second;
  }
  return offset;
}
}  // namespace

void Scene::build_bvh() {
  nodes.clear(); ordered.clear();
  size_t n = n_prims();
  if (n == 0) return;
  std::vector<PrimInfo> info(n);
  for (size_t i = 0; i < n; ++i) {
    B3 bb;
    if (is_instance((int)i)) {  // TransformedPrimitive::world_bounds (primitive.rs:86-88): Transform * Bounds3f, the 8 corners (transform.rs:342-378)
      const Instance& in = instance_of((int)i);
      Scene& o = *objects[in.object];
      if (o.nodes.empty()) { o.max_prims_per_node = max_prims_per_node; o.build_bvh(); }
      const B3 ob = o.n_prims() == 1 ? (o.is_sphere(0) ? quadric_world_bounds(o.sphere_of(0)) : o.tri_world_bounds(0)) : o.nodes[0].bounds;
      bb = b3_from_points(xf_point(in.o2w, v3(ob.mn.x, ob.mn.y, ob.mn.z)), xf_point(in.o2w, v3(ob.mx.x, ob.mn.y, ob.mn.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mn.x, ob.mx.y, ob.mn.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mn.x, ob.mn.y, ob.mx.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mn.x, ob.mx.y, ob.mx.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mx.x, ob.mx.y, ob.mn.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mx.x, ob.mn.y, ob.mx.z)));
      bb = b3_union_p(bb, xf_point(in.o2w, v3(ob.mx.x, ob.mx.y, ob.mx.z)));
    } else bb = is_sphere((int)i) ? quadric_world_bounds(sphere_of((int)i)) : tri_world_bounds((int)i);
    info[i].prim_number = (int)i;
    info[i].bounds = bb;
    info[i].centroid = 0.5f * bb.mn + 0.5f * bb.mx;  // :532
  }
  ordered.reserve(n);
  Builder b{this, info, ordered, max_prims_per_node};
  auto root = b.build(0, n);
  nodes.reserve(b.total_nodes);
  flatten(root.get(), nodes);
}

// ============================================================================ triangle
// Shared hit test of Triangle::intersect (mesh.rs:215-319) and intersect_p (:428-532).
bool Scene::tri_test(int tri, const Ray& ray, TriHit* h) const {
  V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
  V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
  int kz = max_dimension(vabs(ray.d));
  int kx = kz + 1; if (kx == 3) kx = 0;
  int ky = kx + 1; if (ky == 3) ky = 0;
  V3 d = permute(ray.d, kx, ky, kz);
  p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
  float sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0f / d.z;
  p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
  p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
  p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
  float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
  float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
  float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
  if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // :260-270 double-precision fallback
    double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
    e0 = (float)(p2typ1tx - p2txp1ty);
    double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
    e1 = (float)(p0typ2tx - p0txp2ty);
    double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
    e2 = (float)(p1typ0tx - p1txp0ty);
  }
  if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
  float det = e0 + e1 + e2;
  if (det == 0.0f) return false;
  p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
  float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < ray.t_max * det)) || (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)))
    return false;
  float inv_det = 1.0f / det;
  float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
  float t = t_scaled * inv_det;
  float maxzt = max_component(vabs(v3(p0t.z, p1t.z, p2t.z)));
  float delta_z = gamma_n(3) * maxzt;
  float maxxt = max_component(vabs(v3(p0t.x, p1t.x, p2t.x)));
  float maxyt = max_component(vabs(v3(p0t.y, p1t.y, p2t.y)));
  float delta_x = gamma_n(5) * (maxxt + maxzt);
  float delta_y = gamma_n(5) * (maxyt + maxzt);
  float delta_e = 2.0f * (gamma_n(2) * maxxt * maxyt + delta_y * maxxt + delta_x * maxyt);
  float max_e = max_component(vabs(v3(e0, e1, e2)));
  float delta_t = 3.0f * (gamma_n(3) * max_e * maxzt + delta_e * maxzt + delta_z * max_e) * fabsf(inv_det);
  if (t <= delta_t) return false;
  h->t = t; h->b0 = b0; h->b1 = b1; h->b2 = b2;
  return true;
}

// The alpha tests of Triangle::intersect (mesh.rs:353-370: alpha_mask) and Triangle::intersect_p (:534-582: alpha_mask, then shadow_alpha_mask): the
// float texture is evaluated on a local SurfaceInteraction::new(p_hit, zero error, uv_hit, -ray.d, dpdu, dpdv, ...) that carries no ray differentials;
// a value of exactly 0 turns the accepted hit test into a miss.
bool Scene::tri_alpha_rejects(int tri, const Ray& ray, const TriHit& h, bool shadow_ray) const {
  if (tri_alpha.empty() || is_sphere(tri)) return false;  // (Sphere has no alpha mask in the reference)
  const int alpha = tri_alpha[2 * (size_t)tri], shadow_alpha = tri_alpha[2 * (size_t)tri + 1];
  if (alpha < 0 && !(shadow_ray && shadow_alpha >= 0)) return false;
  V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
  P2 uv[3]; tri_uvs(tri, uv);
  V3 dpdu = v3(0, 0, 0), dpdv = v3(0, 0, 0);
  float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y, duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
  V3 dp02 = p0 - p2, dp12 = p1 - p2;
  float determinant = duv02x * duv12y - duv02y * duv12x;
  bool degenerate_uv = fabsf(determinant) < 1e-8f;
  if (!degenerate_uv) {
    float inv_det = 1.0f / determinant;
    if (shadow_ray) {  // intersect_p DIVIDES by inv_det (mesh.rs:549-550) where intersect multiplies (:331-332): a reference quirk, kept.
      dpdu = (duv12y * dp02 - duv02y * dp12) / inv_det;   // No texture class reads dpdu / dpdv or the normal built from them, so it cannot
      dpdv = (-duv12x * dp02 + duv02x * dp12) / inv_det;  // change a mask value - which is why the device does not form them at all.
    } else {
      dpdu = (duv12y * dp02 - duv02y * dp12) * inv_det;
      dpdv = (-duv12x * dp02 + duv02x * dp12) * inv_det;
    }
  }
  if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f) coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
  SurfaceInteraction si;
  V3 p_hit = p0 * h.b0 + p1 * h.b1 + p2 * h.b2;
  si.hit = interaction_new(p_hit, v3(0, 0, 0), normalize(-ray.d), v3(0, 0, 0));
  si.uv = P2{uv[0].x * h.b0 + uv[1].x * h.b1 + uv[2].x * h.b2, uv[0].y * h.b0 + uv[1].y * h.b1 + uv[2].y * h.b2};
  si.dpdu = dpdu; si.dpdv = dpdv;
  if (alpha >= 0 && tex_eval_f(alpha, si) == 0.0f) return true;
  if (shadow_ray && shadow_alpha >= 0 && tex_eval_f(shadow_alpha, si) == 0.0f) return true;
  return false;
}

// mesh.rs:321-425 (the alpha test of :353-370 is tri_alpha_rejects above, applied by the callers right after the hit test)
void Scene::tri_fill_interaction(int tri, const Ray& ray, const TriHit& h, SurfaceInteraction* out) const {
  V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
  const float b0 = h.b0, b1 = h.b1, b2 = h.b2;
  V3 dpdu = v3(0, 0, 0), dpdv = v3(0, 0, 0);
  P2 uv[3]; tri_uvs(tri, uv);
  float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y;
  float duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
  V3 dp02 = p0 - p2, dp12 = p1 - p2;
  float determinant = duv02x * duv12y - duv02y * duv12x;
  bool degenerate_uv = fabsf(determinant) < 1e-8f;
  if (!degenerate_uv) {
    float inv_det = 1.0f / determinant;
    dpdu = (duv12y * dp02 - duv02y * dp12) * inv_det;
    dpdv = (-duv12x * dp02 + duv02x * dp12) * inv_det;
  }
  if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f)
    coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
  float x_abs_sum = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
  float y_abs_sum = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
  float z_abs_sum = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
  V3 p_error = gamma_n(7) * v3(x_abs_sum, y_abs_sum, z_abs_sum);
  V3 p_hit = p0 * b0 + p1 * b1 + p2 * b2;
  P2 uv_hit{uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2};
  const uint8_t flags = tri_flags[tri];
  const bool flip = (flags & 1) != 0, has_n = (flags & 2) != 0, has_s = (flags & 8) != 0;
  // SurfaceInteraction::new (interaction.rs:107-147): wo normalised twice
  SurfaceInteraction si;
  si.hit = interaction_new(p_hit, p_error, normalize(-ray.d), v3(0, 0, 0));
  si.uv = uv_hit; si.dpdu = dpdu; si.dpdv = dpdv;
  V3 n = normalize(cross(dp02, dp12));  // :385
  si.hit.n = n; si.shading.n = n;
  const int i0 = idx[3 * tri], i1 = idx[3 * tri + 1], i2 = idx[3 * tri + 2];
  V3 ns = has_n ? normalize(N[i0] * b0 + N[i1] * b1 + N[i2] * b2) : si.hit.n;         // :390-394
  V3 ss = has_s ? normalize(S[i0] * b0 + S[i1] * b1 + S[i2] * b2) : normalize(si.dpdu);  // :396-400
  V3 ts = cross(ss, ns);
  if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
  else coordinate_system(ns, &ss, &ts);
  si.shading.n = ns; si.shading.dpdu = ss; si.shading.dpdv = ts;
  if (has_n) si.hit.n = face_forward(si.hit.n, si.shading.n);  // :417-422
  else if (flip) { si.hit.n = -si.hit.n; si.shading.n = si.hit.n; }
  *out = si;
}

// ============================================================================ traversal
static inline bool slab_test(const B3& b, const Ray& ray, V3 inv_dir, const int neg[3]) {  // bounds.rs:127-157 (quirk 3)
  const V3* bb = &b.mn;  // bb[0]=min, bb[1]=max
  float tmin = (bb[neg[0]].x - ray.o.x) * inv_dir.x;
  float tmax = (bb[1 - neg[0]].x - ray.o.x) * inv_dir.x;
  float tymin = (bb[neg[1]].y - ray.o.y) * inv_dir.y;
  float tymax = (bb[1 - neg[1]].y - ray.o.y) * inv_dir.y;
  if ((tmin > tymax) || (tymin > tmax)) return false;
  if (tymin > tmin) tmin = tymin;
  if (tymax < tmax) tmax = tymax;
  float tzmin = (bb[neg[2]].z - ray.o.z) * inv_dir.z;
  float tzmax = (bb[1 - neg[2]].z - ray.o.z) * inv_dir.z;
  if ((tmin > tzmax) || (tzmin > tmax)) return false;
  if (tzmin > tmin) tmin = tzmin;
  if (tzmax < tmax) tmax = tzmax;
  return tmin < ray.t_max && tmax > 0.0f;
}

// What object_instance wraps in a TransformedPrimitive (api.rs:1073-1086): the aggregate built over the object's primitives, or - for exactly one - that primitive itself
bool Scene::object_intersect_raw(Ray& ray, int* prim_out, TriHit* hit_out, TraceCounters* tc) const {
  if (n_prims() == 1) {
    if (tc) tc->tris += 1;
    TriHit h;
    if (!prim_test(0, ray, &h)) return false;
    if (tc) tc->tri_hits += 1;
    ray.t_max = h.t; *prim_out = 0; *hit_out = h;
    return true;
  }
  const uint64_t rays = tc ? tc->rays_closest : 0;
  const bool r = intersect_raw(ray, prim_out, hit_out, tc);
  if (tc) tc->rays_closest = rays;  // the nested walk is part of the outer ray
  return r;
}
bool Scene::object_intersect_p(const Ray& ray, TraceCounters* tc) const {
  if (n_prims() == 1) {
    if (tc) tc->tris += 1;
    TriHit h;
    const bool r = prim_test(0, ray, &h);
    if (r && tc) tc->tri_hits += 1;
    return r;
  }
  const uint64_t rays = tc ? tc->rays_any : 0;
  const bool r = intersect_p(ray, tc);
  if (tc) tc->rays_any = rays;
  return r;
}
// Transform * Ray (ray.rs:83-93): origin as a point, direction as a vector, t_max kept - no error offset (unlike Ray::transform)
static inline Ray ray_to_object(const M44& w2o, const Ray& ray) { Ray r = ray; r.o = xf_point(w2o, ray.o); r.d = xf_vector(w2o, ray.d); return r; }

bool Scene::intersect_raw(Ray& ray, int* prim_out, TriHit* hit_out, TraceCounters* tc, int* sub_out) const {  // bvh/mod.rs:366-433
  if (tc) tc->rays_closest += 1;
  if (nodes.empty()) return false;
  bool found = false; int best_prim = -1, best_sub = -1; TriHit best{};
  int to_visit = 0, cur = 0; int stack[64];
  V3 inv_dir = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  int neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
  for (;;) {
    const LinearNode& node = nodes[cur];
    if (tc) tc->nodes += 1;
    if (slab_test(node.bounds, ray, inv_dir, neg)) {
      if (node.n_prims > 0) {
        for (int i = 0; i < node.n_prims; ++i) {
          int prim = (int)node.offset + i;
          if (tc) tc->tris += 1;
          TriHit h;
          if (is_instance(ordered[prim])) {  // TransformedPrimitive::intersect (primitive.rs:90-96)
            const Instance& in = instance_of(ordered[prim]);
            Ray r = ray_to_object(in.w2o, ray);
            int oprim;
            if (objects[in.object]->object_intersect_raw(r, &oprim, &h, tc)) { ray.t_max = r.t_max; found = true; best_prim = prim; best_sub = oprim; best = h; }
            continue;
          }
          // `result = prim.intersect(ray).or(result)`: every accepted test replaces the result and
          // shrinks ray.t_max (GeometricPrimitive::intersect, primitive.rs:45-51)
          if (prim_test(ordered[prim], ray, &h) && !tri_alpha_rejects(ordered[prim], ray, h, false)) {
            if (tc) tc->tri_hits += 1;
            ray.t_max = h.t; found = true; best_prim = prim; best_sub = -1; best = h;
          }
        }
        if (to_visit == 0) break;
        cur = stack[--to_visit];
      } else {
        if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = (int)node.offset; }
        else { stack[to_visit++] = (int)node.offset; cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = stack[--to_visit];
    }
  }
  if (found) { *prim_out = best_prim; *hit_out = best; if (sub_out) *sub_out = best_sub; }
  return found;
}
bool Scene::intersect(Ray& ray, SurfaceInteraction* si, TraceCounters* tc) const {
  int prim, osub = -1; TriHit h;
  if (!intersect_raw(ray, &prim, &h, tc, &osub)) return false;
  if (is_instance(ordered[prim])) {  // the object's interaction, then SurfaceInteraction::transform(primitive_to_world) (interaction.rs:156-190)
    const Instance& in = instance_of(ordered[prim]);
    const Scene& o = *objects[in.object];
    const Ray r = ray_to_object(in.w2o, ray);
    SurfaceInteraction s;
    o.prim_fill_interaction(o.ordered[osub], r, h, &s);
    SurfaceInteraction t;
    V3 perr;
    t.hit.p = xf_point_with_error(in.o2w, s.hit.p, s.hit.p_error, &perr); t.hit.p_error = perr;
    t.hit.wo = normalize(xf_vector(in.o2w, s.hit.wo));
    t.hit.n = normalize(xf_normal(in.w2o, s.hit.n));
    t.uv = s.uv;
    t.dpdu = xf_vector(in.o2w, s.dpdu); t.dpdv = xf_vector(in.o2w, s.dpdv);
    t.shading.n = normalize(xf_normal(in.w2o, s.shading.n));
    t.shading.dpdu = xf_vector(in.o2w, s.shading.dpdu); t.shading.dpdv = xf_vector(in.o2w, s.shading.dpdv);
    t.shading.n = face_forward(t.shading.n, t.hit.n);
    t.prim = prim; t.sub = osub;
    *si = t;
    return true;
  }
  // The reference builds the full SurfaceInteraction for every accepted candidate (mesh.rs:321-425);
  // only the last one survives, so building it once for the final hit gives the same value.
  prim_fill_interaction(ordered[prim], ray, h, si);
  si->prim = prim;
  return true;
}
void Scene::prim_fill_interaction(int prim, const Ray& ray, const TriHit& h, SurfaceInteraction* si) const {
  if (!is_sphere(prim)) { tri_fill_interaction(prim, ray, h, si); return; }
  // Sphere::intersect builds its interaction inside the hit test; it is rebuilt here for the accepted hit. The roots, the clipping retry and
  // therefore p_hit do not depend on ray.t_max (it only rejects), so t_max = infinity reproduces the accepted test's decisions.
  Ray r = ray; r.t_max = kInf;
  SphereHit sh; bool ok = quadric_intersect(sphere_of(prim), r, true, &sh); (void)ok;
  SurfaceInteraction s;
  s.hit.p = sh.p; s.hit.p_error = sh.p_error; s.hit.wo = sh.wo; s.hit.n = sh.n;
  s.uv = sh.uv; s.dpdu = sh.dpdu; s.dpdv = sh.dpdv;
  s.shading.n = sh.sh_n; s.shading.dpdu = sh.dpdu; s.shading.dpdv = sh.dpdv;
  *si = s;
}

bool Scene::intersect_p(const Ray& ray, TraceCounters* tc) const {  // bvh/mod.rs:435-501
  if (tc) tc->rays_any += 1;
  if (nodes.empty()) return false;
  int to_visit = 0, cur = 0; int stack[64];
  V3 inv_dir = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  int neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
  for (;;) {
    const LinearNode& node = nodes[cur];
    if (tc) tc->nodes += 1;
    if (slab_test(node.bounds, ray, inv_dir, neg)) {
      if (node.n_prims > 0) {
        for (int i = 0; i < node.n_prims; ++i) {
          if (tc) tc->tris += 1;
          TriHit h;
          if (is_instance(ordered[node.offset + i])) {  // TransformedPrimitive::intersect_p (primitive.rs:98-101)
            const Instance& in = instance_of(ordered[node.offset + i]);
            if (objects[in.object]->object_intersect_p(ray_to_object(in.w2o, ray), tc)) return true;
            continue;
          }
          if (prim_test(ordered[node.offset + i], ray, &h) && !tri_alpha_rejects(ordered[node.offset + i], ray, h, true)) { if (tc) tc->tri_hits += 1; return true; }
        }
        if (to_visit == 0) break;
        cur = stack[--to_visit];
      } else {
        if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = (int)node.offset; }
        else { stack[to_visit++] = (int)node.offset; cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = stack[--to_visit];
    }
  }
  return false;
}

// ============================================================================ shape sampling
void Scene::tri_sample(int tri, P2 u, Interaction* it, float* pdf) const {  // mesh.rs:610-634
  P2 b = uniform_sample_triangle(u);
  V3 p0, p1, p2; tri_verts(tri, &p0, &p1, &p2);
  float b2 = 1.0f - b.x - b.y;
  V3 p = (b.x * p0) + (b.y * p1) + (b2 * p2);
  V3 normal = normalize(cross(p1 - p0, p2 - p0));
  const uint8_t flags = tri_flags[tri];
  if (flags & 2) {
    const int i0 = idx[3 * tri], i1 = idx[3 * tri + 1], i2 = idx[3 * tri + 2];
    V3 ns = b.x * N[i0] + b.y * N[i1] + b2 * N[i2];
    normal = face_forward(normal, ns);
  } else if (flags & 1) {
    normal = normal * -1.0f;
  }
  V3 p_abs_sum = vabs(b.x * p0) + vabs(b.y * p1) + vabs(b2 * p2);
  V3 p_error = gamma_n(6) * p_abs_sum;
  it->p = p; it->p_error = p_error; it->wo = v3(0, 0, 0); it->n = normal;  // wo unused downstream
  *pdf = 1.0f / tri_area(tri);
}
void Scene::shape_sample_si(int tri, const Interaction& ref, P2 u, Interaction* it, float* pdf_out) const {  // shapes/mod.rs:39-53
  if (is_sphere(tri)) {  // Sphere overrides sample_si (sphere.rs:246-308)
    SpherePoint sp = quadric_sample_si(sphere_of(tri), ref, u, pdf_out);
    it->p = sp.p; it->p_error = sp.p_error; it->wo = v3(0, 0, 0); it->n = sp.n;
    return;
  }
  float pdf; tri_sample(tri, u, it, &pdf);
  V3 wi = it->p - ref.p;
  if (length_squared(wi) == 0.0f) pdf = 0.0f;
  else {
    wi = normalize(wi);
    pdf *= distance_squared(ref.p, it->p) / fabsf(dot(it->n, -wi));
    if (std::isinf(pdf)) pdf = 0.0f;
  }
  *pdf_out = pdf;
}
float Scene::shape_pdf_wi(int tri, const Interaction& ref, V3 wi, TraceCounters* tc) const {  // shapes/mod.rs:59-68
  if (is_sphere(tri)) return quadric_pdf_wi(sphere_of(tri), ref, wi, tc ? &tc->tris : nullptr);  // sphere.rs:310-334 / the trait default
  Ray ray = spawn_ray(ref, wi);
  TriHit h;
  if (tc) tc->tris += 1;
  if (!tri_test(tri, ray, &h) || tri_alpha_rejects(tri, ray, h, false)) return 0.0f;  // Shape::pdf_wi calls self.intersect (shapes/mod.rs:61)
  SurfaceInteraction li; tri_fill_interaction(tri, ray, h, &li);
  return distance_squared(ref.p, li.hit.p) / (fabsf(dot(li.hit.n, -wi)) * tri_area(tri));
}

// ============================================================================ lights
void Scene::preprocess_lights() {  // scene.rs:29-49: light.preprocess runs while scene.lights is still empty
  infinite_lights.clear();
  B3 wb = world_bounds();
  for (size_t i = 0; i < lights.size(); ++i) {
    Light& l = lights[i];
    if (l.kind == LIGHT_DIFFUSE_AREA) l.area = shape_area(l.tri);
    if (l.kind == LIGHT_DISTANT || l.kind == LIGHT_INFINITE) b3_bounding_sphere(wb, &l.w_center, &l.w_radius);
    if (l.kind == LIGHT_INFINITE) infinite_lights.push_back((int)i);
  }
}

Scene::LiSample Scene::light_sample_li(const Light& l, const Interaction& it, P2 u) const {
  LiSample s;
  s.p0 = it;
  switch (l.kind) {
    case LIGHT_DIFFUSE_AREA: {  // diffuse.rs:59-70
      Interaction p_shape; float pdf;
      shape_sample_si(l.tri, it, u, &p_shape, &pdf);
      V3 wi = normalize(p_shape.p - it.p);
      s.li = area_light_l(l, p_shape, -wi); s.wi = wi; s.pdf = pdf; s.p1 = p_shape;
      return s;
    }
    case LIGHT_POINT: {  // point.rs:43-54 (quirk 11: I/(4 pi r^2))
      V3 wi = l.pos - it.p;
      float r2 = length_squared(wi);
      s.li = l.intensity / (4.0f * kPi * r2); s.wi = normalize(wi); s.pdf = 1.0f;
      s.p1 = Interaction{l.pos, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
      return s;
    }
    case LIGHT_DISTANT: {  // distant.rs:57-70
      V3 p_outside = it.p + l.pos * (2.0f * l.w_radius);
      s.li = l.intensity; s.wi = l.pos; s.pdf = 1.0f;
      s.p1 = Interaction{p_outside, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
      return s;
    }
    default: return infinite_sample_li(*this, l, it, u);
  }
}
float Scene::light_pdf_li(const Light& l, const Interaction& it, V3 wi, TraceCounters* tc) const {
  switch (l.kind) {
    case LIGHT_DIFFUSE_AREA: return shape_pdf_wi(l.tri, it, wi, tc);  // diffuse.rs:72-74
    case LIGHT_INFINITE: return infinite_pdf_li(*this, l, it, wi);
    default: return 0.0f;
  }
}
RGB Scene::light_le(const Light& l, const Ray& ray) const {
  if (l.kind == LIGHT_INFINITE) return infinite_le(*this, l, ray);
  return rgb(0, 0, 0);  // light/mod.rs:90-92
}

// ============================================================================ Perlin noise (rc/noise.rs)
static const uint8_t kNoisePerm[512] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180,
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
static inline float noise_grad(int x, int y, int z, float dx, float dy, float dz) {  // noise.rs:68-76
  int h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z];
  h &= 15;
  float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
  float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
  return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
static inline float noise_weight(float t) { float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }  // :78-83

float noise_perlin(float x, float y, float z) {  // noise.rs:8-43
  int ix = f2i_sat(floorf(x)), iy = f2i_sat(floorf(y)), iz = f2i_sat(floorf(z));
  float dx = x - (float)ix, dy = y - (float)iy, dz = z - (float)iz;
  ix &= 255; iy &= 255; iz &= 255;
  float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
  float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
  float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
  float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
  float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
  float x00 = lerp_f(wx, w000, w100), x10 = lerp_f(wx, w010, w110), x01 = lerp_f(wx, w001, w101), x11 = lerp_f(wx, w011, w111);
  float y0 = lerp_f(wy, x00, x10), y1 = lerp_f(wy, x01, x11);
  return lerp_f(wz, y0, y1);
}
float noise_fbm(V3 p, V3 dpdx, V3 dpdy, float omega, uint32_t max_octaves) {  // noise.rs:46-66
  float len2 = fmaxf(dot(dpdx, dpdx), dot(dpdy, dpdy));
  float n = clamp_t(-1.0f - 0.5f * log2f(len2), 0.0f, (float)max_octaves);
  uint32_t n_int = f2u_sat(floorf(n));
  float sum = 0.0f, lambda = 1.0f, o = 1.0f;
  for (uint32_t i = 0; i < n_int; ++i) {
    sum += o * noise_perlin(lambda * p.x, lambda * p.y, lambda * p.z);
    lambda *= 1.99f;
    o *= omega;
  }
  float n_partial = n - (float)n_int;
  float v = clamp_t((n_partial - 0.3f) / (0.7f - 0.3f), 0.0f, 1.0f);  // smooth_step, :85-89
  sum += o * (v * v * (-2.0f * v + 3.0f)) * noise_perlin(lambda * p.x, lambda * p.y, lambda * p.z);
  return sum;
}

// ============================================================================ textures
RGB Scene::tex_eval(int id, const SurfaceInteraction& si) const {
  const Texture& t = textures[id];
  switch (t.kind) {
    case TEX_CONST: return t.value;
    case TEX_SCALE: return tex_eval(t.tex1, si) * tex_eval(t.tex2, si);  // scale.rs:23-25
    case TEX_MIX: {                                                       // mix.rs:24
      RGB t1 = tex_eval(t.tex1, si), t2 = tex_eval(t.tex2, si);
      float amt = tex_eval_f(t.amount, si);
      return t1 * (1.0f - amt) + t2 * amt;
    }
    case TEX_CHECKER: {  // checkerboard.rs:102-143 (2D, UVMapping2D)
      P2 st{t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv};
      P2 dstdx{t.su * si.dudx, t.sv * si.dvdx}, dstdy{t.su * si.dudy, t.sv * si.dvdy};
      if (t.amount == 0) {  // AAMethod::None: `floor() as u32` saturates negatives to 0, the u32 sum wraps
        auto as_u32 = [](float f) { uint64_t v = f2u_sat(f); return (uint32_t)(v > 0xffffffffull ? 0xffffffffull : v); };
        uint32_t a = as_u32(floorf(st.x)), b = as_u32(floorf(st.y));
        return ((uint32_t)(a + b) % 2u == 0u) ? tex_eval(t.tex1, si) : tex_eval(t.tex2, si);
      }
      float ds = fmaxf(fabsf(dstdx.x), fabsf(dstdy.x)), dt = fmaxf(fabsf(dstdx.y), fabsf(dstdy.y));
      float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
      if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1)) {  // filter inside one check: point sample
        int sum = (int)((int64_t)f2i_sat(floorf(st.x)) + (int64_t)f2i_sat(floorf(st.y)));  // i32 add (wraps only at 2^31)
        return (sum % 2 == 0) ? tex_eval(t.tex1, si) : tex_eval(t.tex2, si);
      }
      auto bump_int = [](float x) { return floorf(x / 2.0f) + 2.0f * fmaxf(x / 2.0f - floorf(x / 2.0f) - 0.5f, 0.0f); };
      float sint = (bump_int(s1) - bump_int(s0)) / (2.0f * ds);
      float tint = (bump_int(t1) - bump_int(t0)) / (2.0f * dt);
      float area2 = sint + tint - 2.0f * sint * tint;
      if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
      return tex_eval(t.tex1, si) * (1.0f - area2) + tex_eval(t.tex2, si) * area2;
    }
    case TEX_UV: {  // uv.rs:50-54
      P2 st{t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv};
      return rgb(st.x - floorf(st.x), st.y - floorf(st.y), 0.0f);
    }
    case TEX_FBM: return grey(noise_fbm(si.hit.p, si.dpdx, si.dpdy, t.value.r, (uint32_t)(t.amount < 0 ? 0 : t.amount)));  // fbm.rs:18-21, IdentityMapping3D with identity
    default: return image_tex_eval(*this, t, si);
  }
}

// ============================================================================ materials
static Bxdf lambert_r(RGB r) { Bxdf b; b.kind = BX_LAMBERT_R; b.r = r; return b; }
static Bxdf lambert_t(RGB t) { Bxdf b; b.kind = BX_LAMBERT_T; b.r = t; return b; }
static Fresnel fr_diel(float ei, float et) { Fresnel f; f.kind = FR_DIELECTRIC; f.eta_i = ei; f.eta_t = et; return f; }
static Bxdf micro_r(RGB r, float ax, float ay, Fresnel f) { Bxdf b; b.kind = BX_MICRO_R; b.r = r; b.dist.ax = ax; b.dist.ay = ay; b.fresnel = f; return b; }
static Bxdf micro_t(RGB t, float ax, float ay, float ea, float eb) {
  Bxdf b; b.kind = BX_MICRO_T; b.r = t; b.dist.ax = ax; b.dist.ay = ay; b.eta_a = ea; b.eta_b = eb; b.fresnel = fr_diel(ea, eb); return b;
}
static Bxdf spec_r(RGB r, Fresnel f) { Bxdf b; b.kind = BX_SPEC_R; b.r = r; b.fresnel = f; return b; }
static Bxdf spec_t(RGB t, float ea, float eb) { Bxdf b; b.kind = BX_SPEC_T; b.r = t; b.eta_a = ea; b.eta_b = eb; b.fresnel = fr_diel(ea, eb); return b; }

// material::bump (rc/material/mod.rs:50-92). Triangles carry dndu = dndv = 0 (mesh.rs:372-382 passes zero()), and so do
// their shading.dndu / dndv (interaction.rs:118-160), which keeps the terms below in place with zero vectors.
void Scene::bump(int tex, SurfaceInteraction& si) const {
  const V3 dndu = v3(0, 0, 0), dndv = v3(0, 0, 0);
  SurfaceInteraction si_eval = si;
  float du = 0.5f * (fabsf(si.dudx) + fabsf(si.dudy));
  if (du == 0.0f) du = 0.0005f;
  si_eval.hit.p = si.hit.p + du * si.shading.dpdu;
  si_eval.uv = P2{si.uv.x + du, si.uv.y + 0.0f};
  si_eval.hit.n = normalize(cross(si.shading.dpdu, si.shading.dpdv) + du * dndu);
  float u_displace = tex_eval_f(tex, si_eval);
  float dv = 0.5f * (fabsf(si.dvdx) + fabsf(si.dvdy));
  if (dv == 0.0f) dv = 0.0005f;
  si_eval.hit.p = si.hit.p + dv * si.shading.dpdv;
  si_eval.uv = P2{si.uv.x + 0.0f, si.uv.y + dv};
  si_eval.hit.n = normalize(cross(si.shading.dpdu, si.shading.dpdv) + dv * dndv);
  float v_displace = tex_eval_f(tex, si_eval);
  float displace = tex_eval_f(tex, si);
  V3 dpdu = si.shading.dpdu + (u_displace - displace) / du * si.shading.n + displace * dndu;
  V3 dpdv = si.shading.dpdv + (v_displace - displace) / dv * si.shading.n + displace * dndv;
  // set_shading_geometry(dpdu, dpdv, dndu, dndv, false), interaction.rs:218-242
  si.shading.n = normalize(cross(dpdu, dpdv));
  if (si.prim >= 0) {  // reverse_orientation ^ transform_swaps_handedness of si.shape: the object's shape for a hit inside an instance
    int tri; const Scene& o = owner_of(si, &tri);
    if (o.tri_flags[tri] & 1) si.shading.n = si.shading.n * -1.0f;
  }
  si.shading.n = face_forward(si.shading.n, si.hit.n);
  si.shading.dpdu = dpdu; si.shading.dpdv = dpdv;
}
void Scene::build_bsdf(int mat, SurfaceInteraction& si, Bsdf* bsdf, int depth) const {
  const Material& m = materials[mat];
  bsdf->n = 0;
  float eta = 1.0f;
  if (m.kind != MAT_MIX && m.bump >= 0) bump(m.bump, si);  // first statement of every compute_scattering_functions but mix
  switch (m.kind) {
    case MAT_MATTE: {  // matte.rs:37-62
      RGB r = clamp_pos(tex_eval(m.kd, si));
      float sigma = clamp_t(tex_eval_f(m.sigma, si), 0.0f, 1.0f);
      if (!is_black(r)) { if (sigma == 0.0f) bsdf->add(lambert_r(r)); else bsdf->add(make_oren_nayar(r, sigma)); }
      break;
    }
    case MAT_PLASTIC: {  // plastic.rs:45-75
      RGB kd = tex_eval(m.kd, si), ks = tex_eval(m.ks, si);
      if (!is_black(kd)) bsdf->add(lambert_r(kd));
      if (!is_black(ks)) {
        float rough = tex_eval_f(m.roughness, si);
        if (m.remap_roughness) rough = TRDist::roughness_to_alpha(rough);
        bsdf->add(micro_r(ks, rough, rough, fr_diel(1.5f, 1.0f)));
      }
      break;
    }
    case MAT_METAL: {  // metal.rs:50-82
      float ur = tex_eval_f(m.urough >= 0 ? m.urough : m.roughness, si);
      float vr = tex_eval_f(m.vrough >= 0 ? m.vrough : m.roughness, si);
      if (m.remap_roughness) { ur = TRDist::roughness_to_alpha(ur); vr = TRDist::roughness_to_alpha(vr); }
      Fresnel f; f.kind = FR_CONDUCTOR; f.c_eta_i = rgb(1, 1, 1); f.c_eta_t = tex_eval(m.eta, si); f.c_k = tex_eval(m.k, si);
      bsdf->add(micro_r(rgb(1, 1, 1), ur, vr, f));
      break;
    }
    case MAT_MIRROR: {  // mirror.rs:30-48
      RGB R = clamp_pos(tex_eval(m.kr, si));
      if (!is_black(R)) { Fresnel f; f.kind = FR_NOOP; bsdf->add(spec_r(R, f)); }
      break;
    }
    case MAT_GLASS: {  // glass.rs:53-106 (allow_multiple_lobes == true from path.rs:145)
      eta = tex_eval_f(m.eta, si);
      float ur = tex_eval_f(m.urough, si), vr = tex_eval_f(m.vrough, si);
      RGB r = tex_eval(m.kr, si), t = tex_eval(m.kt, si);
      if (!is_black(r) || !is_black(t)) {
        bool is_specular = ur == 0.0f && vr == 0.0f;
        if (is_specular) {
          Bxdf b; b.kind = BX_FRESNEL_SPEC; b.r = r; b.t = t; b.eta_a = 1.0f; b.eta_b = eta; bsdf->add(b);
        } else {
          if (m.remap_roughness) { ur = TRDist::roughness_to_alpha(ur); vr = TRDist::roughness_to_alpha(vr); }
          if (!is_black(r)) bsdf->add(micro_r(r, ur, vr, fr_diel(1.0f, eta)));
          if (!is_black(t)) bsdf->add(micro_t(r, ur, vr, 1.0f, eta));  // quirk 9: passes `r`
        }
      }
      break;
    }
    case MAT_UBER: {  // uber.rs:63-126
      float e = tex_eval_f(m.eta, si);
      RGB op = clamp_pos(tex_eval(m.opacity, si));
      RGB t = clamp_pos(rgb(1, 1, 1) - op);
      eta = e;
      if (!is_black(t)) { eta = 1.0f; bsdf->add(spec_t(t, 1.0f, 1.0f)); }
      RGB kd = op * clamp_pos(tex_eval(m.kd, si));
      if (!is_black(kd)) bsdf->add(lambert_r(kd));
      RGB ks = op * clamp_pos(tex_eval(m.ks, si));
      if (!is_black(ks)) {
        float ru = tex_eval_f(m.urough >= 0 ? m.urough : m.roughness, si);
        float rv = tex_eval_f(m.vrough >= 0 ? m.vrough : m.roughness, si);
        if (m.remap_roughness) { ru = TRDist::roughness_to_alpha(ru); rv = TRDist::roughness_to_alpha(rv); }
        bsdf->add(micro_r(ks, ru, rv, fr_diel(1.0f, e)));
      }
      RGB kr = op * clamp_pos(tex_eval(m.kr, si));
      if (!is_black(kr)) bsdf->add(spec_r(kr, fr_diel(1.0f, e)));
      RGB kt = op * clamp_pos(tex_eval(m.kt, si));
      if (!is_black(kt)) bsdf->add(spec_t(kt, 1.0f, e));
      break;
    }
    case MAT_SUBSTRATE: {  // substrate.rs:43-71
      RGB d = clamp_pos(tex_eval(m.kd, si)), s = clamp_pos(tex_eval(m.ks, si));
      float ru = tex_eval_f(m.urough, si), rv = tex_eval_f(m.vrough, si);
      if (!is_black(d) || !is_black(s)) {
        if (m.remap_roughness) { ru = TRDist::roughness_to_alpha(ru); rv = TRDist::roughness_to_alpha(rv); }
        Bxdf b; b.kind = BX_FRESNEL_BLEND; b.r = d; b.t = s; b.dist.ax = ru; b.dist.ay = rv; bsdf->add(b);
      }
      break;
    }
    case MAT_TRANSLUCENT: {  // translucent.rs:49-101
      eta = 1.5f;
      RGB r = clamp_pos(tex_eval(m.reflect, si)), t = clamp_pos(tex_eval(m.transmit, si));
      if (!is_black(r) || !is_black(t)) {
        RGB kd = clamp_pos(tex_eval(m.kd, si));
        if (!is_black(kd)) {
          if (!is_black(r)) bsdf->add(lambert_r(r * kd));
          if (!is_black(t)) bsdf->add(lambert_t(t * kd));
        }
        RGB ks = clamp_pos(tex_eval(m.ks, si));
        if (!is_black(ks) && (!is_black(r) || !is_black(t))) {
          float rough = tex_eval_f(m.roughness, si);
          if (m.remap_roughness) rough = TRDist::roughness_to_alpha(rough);
          if (!is_black(r)) bsdf->add(micro_r(r * ks, rough, rough, fr_diel(1.0f, eta)));
          if (!is_black(t)) bsdf->add(micro_t(t * ks, rough, rough, 1.0f, eta));
        }
      }
      break;
    }
    case MAT_DISNEY: {  // disney.rs:82-213 (slot reuse: see MatKind)
      auto disney = [](int kind, RGB r, float a = 0.0f, float b = 0.0f) { Bxdf x; x.kind = kind; x.r = r; x.a = a; x.b = b; return x; };
      auto lerp_rgb = [](float t, RGB a, RGB b) { return a * (1.0f - t) + b * t; };
      const bool thin = m.m1 != 0;
      RGB c = clamp_pos(tex_eval(m.kd, si));
      float metallic_weight = tex_eval_f(m.ks, si);
      float e = tex_eval_f(m.eta, si);
      float strans = tex_eval_f(m.opacity, si);
      float diffuse_weight = (1.0f - metallic_weight) * (1.0f - strans);
      float dt = tex_eval_f(m.amount, si) / 2.0f;
      float rough = tex_eval_f(m.roughness, si);
      float lum = lum_y(c);
      RGB c_tint = lum > 0.0f ? c / lum : rgb(1, 1, 1);
      float sheen_weight = tex_eval_f(m.kt, si);
      RGB c_sheen = rgb(0, 0, 0);
      if (sheen_weight > 0.0f) { float stint = tex_eval_f(m.sigma, si); c_sheen = lerp_rgb(stint, rgb(1, 1, 1), c_tint); }
      if (diffuse_weight > 0.0f) {
        if (thin) {
          float flat = tex_eval_f(m.transmit, si);
          bsdf->add(disney(BX_DISNEY_DIFFUSE, diffuse_weight * (1.0f - flat) * (1.0f - dt) * c));
          bsdf->add(disney(BX_DISNEY_FAKESS, diffuse_weight * flat * (1.0f - dt) * c, rough));
        } else {
          RGB sd = tex_eval(m.reflect, si);
          if (is_black(sd)) bsdf->add(disney(BX_DISNEY_DIFFUSE, diffuse_weight * c));
          else bsdf->add(spec_t(grey(1.0f), 1.0f, e));  // stand-in of the missing BSSRDF (:137-144)
        }
        bsdf->add(disney(BX_DISNEY_RETRO, diffuse_weight * c, rough));
        if (sheen_weight > 0.0f) bsdf->add(disney(BX_DISNEY_SHEEN, diffuse_weight * sheen_weight * c_sheen));
      }
      float aspect = sqrtf(1.0f - tex_eval_f(m.urough, si) * 0.9f);
      float ax = fmaxf(0.001f, (rough * rough) / aspect), ay = fmaxf(0.001f, (rough * rough) * aspect);
      float spec_tint = tex_eval_f(m.kr, si);
      float r0eta = ((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f));  // schlick_r0_from_eta, :497-503
      RGB cspec0 = lerp_rgb(metallic_weight, r0eta * lerp_rgb(spec_tint, rgb(1, 1, 1), c_tint), c);
      Fresnel f; f.kind = FR_DISNEY; f.d_r0 = cspec0; f.d_metallic = metallic_weight; f.d_eta = e;
      { Bxdf x = micro_r(c, ax, ay, f); x.dist.separable_g = true; bsdf->add(x); }
      float cc = tex_eval_f(m.vrough, si);
      if (cc > 0.0f) bsdf->add(disney(BX_DISNEY_CLEARCOAT, rgb(0, 0, 0), cc, lerp_f(tex_eval_f(m.k, si), 0.1f, 0.001f)));
      if (strans > 0.0f) {
        RGB t = strans * rgb_sqrt(c);
        if (thin) {
          float rscaled = (0.65f * e - 0.35f) * rough;
          float ax2 = fmaxf(0.001f, (rscaled * rscaled) / aspect), ay2 = fmaxf(0.001f, (rscaled * rscaled) * aspect);
          bsdf->add(micro_t(t, ax2, ay2, 1.0f, e));
        } else { Bxdf x = micro_t(t, ax, ay, 1.0f, e); x.dist.separable_g = true; bsdf->add(x); }
      }
      if (thin) bsdf->add(lambert_t(dt * c));
      eta = 1.0f;  // Bsdf::new(si, 1.0, ..), :212
      break;
    }
    case MAT_MIX: {  // mixmat.rs:34-64
      RGB s1 = clamp_pos(tex_eval(m.amount, si));
      RGB s2 = clamp_pos(rgb(1, 1, 1) - s1);
      Bsdf b1, b2;
      SurfaceInteraction si2 = si;  // mat2 works on a clone taken before mat1 touches si (mixmat.rs:43)
      if (depth < 4) { build_bsdf(m.m1, si, &b1, depth + 1); build_bsdf(m.m2, si2, &b2, depth + 1); }
      *bsdf = b1;  // frame + eta of mat1's Bsdf are kept, lobes replaced
      bsdf->n = 0;
      for (int i = 0; i < b1.n; ++i) { Bxdf b = b1.bxdfs[i]; b.wrap_scaled(s1); bsdf->add(b); }
      for (int i = 0; i < b2.n; ++i) { Bxdf b = b2.bxdfs[i]; b.wrap_scaled(s2); bsdf->add(b); }
      return;
    }
    default: break;
  }
  bsdf->eta = eta;
  bsdf->init_frame(si.shading.n, si.hit.n, si.shading.dpdu);
}

// ============================================================================ light distribution
void LightDistribution::init(const Scene* s, const char* strategy, uint32_t max_voxels) {  // path.rs:86-94
  scene = s;
  size_t nl = s->lights.size();
  uniform = (std::string(strategy) == "uniform") || nl == 1;
  if (uniform) {  // lightdistrib.rs:41-47
    std::vector<float> prob(nl, 1.0f);
    uniform_distrib.init(prob.data(), nl);
    return;
  }
  B3 b = s->world_bounds();  // :67-99
  V3 diag = b3_diagonal(b);
  float b_max = diag[b3_maximum_extent(b)];
  size_t total = 1;
  for (int i = 0; i < 3; ++i) {
    uint32_t v = f2u32_sat(roundf(diag[i] / b_max * (float)max_voxels));
    n_voxels[i] = v > 1u ? v : 1u;
    total *= n_voxels[i];
  }
  table = std::vector<std::atomic<Distribution1D*>>(total);
  for (auto& p : table) p.store(nullptr);
}
void LightDistribution::voxel_of(V3 p, int pi[3]) const {  // :187-198
  V3 offset = b3_offset(scene->world_bounds(), p);
  for (int i = 0; i < 3; ++i) pi[i] = clamp_t(f2i_sat(offset[i] * (float)n_voxels[i]), 0, (int)n_voxels[i] - 1);
}
Distribution1D* LightDistribution::compute_distribution(const int pi[3]) const {  // :101-179
  const B3 wb = scene->world_bounds();
  V3 p0 = v3((float)pi[0] / (float)n_voxels[0], (float)pi[1] / (float)n_voxels[1], (float)pi[2] / (float)n_voxels[2]);
  V3 p1 = v3(((float)pi[0] + 1.0f) / (float)n_voxels[0], ((float)pi[1] + 1.0f) / (float)n_voxels[1], ((float)pi[2] + 1.0f) / (float)n_voxels[2]);
  B3 vb = b3_from_points(b3_lerp(wb, p0), b3_lerp(wb, p1));
  const uint64_t n_samples = 128;
  size_t nl = scene->lights.size();
  std::vector<float> contrib(nl, 0.0f);
  for (uint64_t i = 0; i < n_samples; ++i) {
    V3 po = b3_lerp(vb, v3(radical_inverse(0, i), radical_inverse(1, i), radical_inverse(2, i)));
    Interaction intr = interaction_new(po, v3(0, 0, 0), v3(1, 0, 0), v3(0, 0, 0));
    P2 u{radical_inverse(3, i), radical_inverse(4, i)};
    for (size_t j = 0; j < nl; ++j) {
      Scene::LiSample s = scene->light_sample_li(scene->lights[j], intr, u);
      if (s.pdf > 0.0f) contrib[j] += lum_y(s.li) / s.pdf;
    }
  }
  float sum = 0.0f;
  for (float c : contrib) sum += c;  // iter().sum(): sequential from 0.0
  float avg = sum / (float)(n_samples * (uint64_t)nl);
  float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
  for (float& c : contrib) c = fmaxf(c, min_contrib);
  return new Distribution1D(contrib.data(), nl);
}
const Distribution1D* LightDistribution::lookup(V3 p) {
  if (uniform) return &uniform_distrib;
  int pi[3]; voxel_of(p, pi);
  size_t idx = ((size_t)pi[2] * n_voxels[1] + (size_t)pi[1]) * n_voxels[0] + (size_t)pi[0];
  Distribution1D* d = table[idx].load(std::memory_order_acquire);
  if (d) return d;
  Distribution1D* fresh = compute_distribution(pi);
  Distribution1D* expected = nullptr;
  if (table[idx].compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) return fresh;
  delete fresh;
  return expected;
}

}  // namespace orc
