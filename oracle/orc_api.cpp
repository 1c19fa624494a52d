// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// PathIntegrator::li + estimate_direct + the render driver, and the flat C API that tests,
// smoke() and bench.py's cpu_baseline leg call through ctypes.
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include "orc_mipmap.h"
#include "orc_render.h"

namespace orc {

// ============================================================================ integrator
// rc/integrator/mod.rs:222-318
RGB PathIntegrator::estimate_direct(const SurfaceInteraction& it, const Bsdf& bsdf, P2 u_scattering, const Light& light, int light_index,
                                    P2 u_light, PathStats& st) const {
  const uint32_t bsdf_flags = BSDF_ALL & ~BSDF_SPECULAR;
  RGB ld = rgb(0, 0, 0);
  Scene::LiSample ls = scene->light_sample_li(light, it.hit, u_light);
  RGB li = ls.li;
  if (mis_mode == 2 && !Scene::is_delta(light)) li = rgb(0, 0, 0);  // test hook: BSDF sampling alone
  if (ls.pdf > 0.0f && !is_black(li)) {
    RGB f = bsdf.f(it.hit.wo, ls.wi, bsdf_flags) * fabsf(dot(ls.wi, it.shading.n));
    float scattering_pdf = bsdf.pdf(it.hit.wo, ls.wi, bsdf_flags);
    if (!is_black(f)) {
      Ray r = spawn_ray_to_interaction(ls.p0, ls.p1);  // VisibilityTester::unoccluded, light/mod.rs:52-55
      if (scene->intersect_p(r, &st.shadow)) li = rgb(0, 0, 0);
      if (!is_black(li)) {
        if (Scene::is_delta(light)) ld = ld + f * li / ls.pdf;
        else {
          float weight = power_heuristic(1, ls.pdf, 1, scattering_pdf);
          if (mis_mode == 1) weight = 1.0f;  // test hook: light sampling alone
          ld = ld + f * li * weight / ls.pdf;
        }
      }
    }
  }
  if (!Scene::is_delta(light) && mis_mode != 1) {
    SampleF bs = bsdf.sample_f(it.hit.wo, u_scattering, bsdf_flags);
    RGB f = bs.f * fabsf(dot(bs.wi, it.shading.n));
    bool sampled_specular = (bs.type & BSDF_SPECULAR) != 0;
    if (!is_black(f) && bs.pdf > 0.0f) {
      float weight = 1.0f;
      if (!sampled_specular) {
        st.pdf_wi_tests += (light.kind == LIGHT_DIFFUSE_AREA);
        float light_pdf = scene->light_pdf_li(light, it.hit, bs.wi, nullptr);
        if (light_pdf == 0.0f) return ld;
        weight = power_heuristic(1, bs.pdf, 1, light_pdf);
        if (mis_mode == 2) weight = 1.0f;  // test hook: BSDF sampling alone
      }
      Ray ray = spawn_ray(it.hit, bs.wi);
      SurfaceInteraction light_isect;
      RGB li2;
      if (scene->intersect(ray, &light_isect, &st.mis)) {
        int tri = scene->ordered[light_isect.prim];
        // area_light.id() == light.id()  (mod.rs:295-304): ids are indices into scene.lights here
        if (!scene->is_instance(tri) && scene->tri_light[tri] >= 0 && scene->tri_light[tri] == light_index) li2 = scene->isect_le(light_isect, -bs.wi);
        else li2 = rgb(0, 0, 0);
      } else {
        li2 = scene->light_le(light, ray);
      }
      if (!is_black(li2)) ld = ld + f * li2 * weight / bs.pdf;
    }
  }
  return ld;
}

// rc/integrator/mod.rs:186-220
RGB PathIntegrator::uniform_sample_one_light(const SurfaceInteraction& it, const Bsdf& bsdf, ZeroTwoSequence& sampler, const Distribution1D* d,
                                             PathStats& st) const {
  size_t n_lights = scene->lights.size();
  if (n_lights == 0) return rgb(0, 0, 0);
  float s = sampler.get_1d();
  size_t light_num; float light_pdf;
  d->sample_discrete(s, &light_num, &light_pdf);  // a distribution is always supplied by path.rs:154-163
  if (light_pdf == 0.0f) return rgb(0, 0, 0);
  P2 u_light = sampler.get_2d();
  P2 u_scattering = sampler.get_2d();
  return estimate_direct(it, bsdf, u_scattering, scene->lights[light_num], (int)light_num, u_light, st) / light_pdf;
}

// rc/integrator/path.rs:96-215
RGB PathIntegrator::li(Ray ray, ZeroTwoSequence& sampler, PathStats& st) const {
  RGB l = rgb(0, 0, 0), beta = rgb(1, 1, 1);
  bool specular_bounce = false;
  int bounces = 0;
  float eta_scale = 1.0f;
  for (;;) {
    SurfaceInteraction isect;
    bool found = scene->intersect(ray, &isect, &st.closest);
    if (bounces == 0 || specular_bounce) {
      if (found) l = l + beta * scene->isect_le(isect, -ray.d);
      else for (int li_ : scene->infinite_lights) l = l + beta * scene->light_le(scene->lights[li_], ray);
    }
    if (!found || bounces >= max_depth) break;
    compute_differential(isect, ray);  // interaction.rs:192-203
    Bsdf bsdf;
    scene->build_bsdf(scene->material_of(isect), isect, &bsdf);
    const Distribution1D* d = distrib->lookup(isect.hit.p);
    if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
      st.nee_total += 1;
      RGB ld = beta * uniform_sample_one_light(isect, bsdf, sampler, d, st);
      if (is_black(ld)) st.zero_radiance += 1;
      l = l + ld;
    }
    V3 wo = -ray.d;  // quirk 12: not normalised
    SampleF bs = bsdf.sample_f(wo, sampler.get_2d(), BSDF_ALL);
    if (is_black(bs.f) || bs.pdf <= 0.0f) break;
    beta = beta * bs.f * fabsf(dot(bs.wi, isect.shading.n)) / bs.pdf;
    specular_bounce = (bs.type & BSDF_SPECULAR) != 0;
    if ((bs.type & BSDF_SPECULAR) && (bs.type & BSDF_TRANSMISSION)) {
      float eta = bsdf.eta;
      eta_scale *= dot(wo, isect.hit.n) > 0.0f ? eta * eta : 1.0f / (eta * eta);
    }
    ray = spawn_ray(isect.hit, bs.wi);
    RGB rr_beta = beta * eta_scale;
    if (max_component_value(rr_beta) < rr_threshold && bounces > 3) {
      float q = fmaxf(1.0f - max_component_value(rr_beta), 0.05f);
      if (sampler.get_1d() < q) break;
      beta = beta / (1.0f - q);
    }
    bounces += 1;
  }
  st.path_len_sum += (uint64_t)bounces;
  return l;
}

}  // namespace orc

// ============================================================================ C API
using namespace orc;

extern "C" {

struct orc_render_params {
  int32_t xres, yres;
  float crop[4];           // xmin xmax ymin ymax (fractions)
  int32_t filter_kind;     // FilterKind
  float filter_params[4];  // xwidth, ywidth, alpha|B, C
  float film_scale, max_sample_luminance;
  float cam_to_world[16], cam_to_world_inv[16];
  float fov, lens_radius, focal_distance;
  int32_t spp, sampler_dims, sampler_mode;  // SamplerMode
  int32_t max_depth;
  float rr_threshold;
  int32_t light_strategy;  // 0 = "spatial", 1 = "uniform"
  int32_t pixel_bounds[4]; // "pixelbounds" x0 x1 y0 y1, intersected with the sample bounds when has_pixel_bounds (path.rs:53-69)
  int32_t n_threads, tile_size;
  float screen_window[4];  // xmin xmax ymin ymax; xmax <= xmin => PerspectiveCamera::create's default from the aspect ratio (camera.rs:86-107)
  int32_t has_pixel_bounds; // the parameter was given: a degenerate intersection is kept (only logged, path.rs:66-68) and renders nothing
  int32_t mis_mode;         // test hook, see PathIntegrator::mis_mode (0 = the reference)
};
struct orc_stats {
  uint64_t camera_rays, rays_closest, rays_shadow, rays_mis, nodes_closest, nodes_shadow, nodes_mis, tris_closest, tris_shadow, tris_mis;
  uint64_t pdf_wi_tests, zero_radiance, nee_total, path_len_sum, scrubbed;
  double seconds;
};

void* orc_scene_new() { return new Scene(); }
void orc_scene_free(void* h) { delete (Scene*)h; }

int orc_scene_set_mesh(void* h, const float* P, int nv, const int32_t* idx, int nt, const float* N, const float* UV, const float* S,
                       const int32_t* tri_material, const int32_t* tri_light, const uint8_t* tri_flags) {
  Scene* s = (Scene*)h;
  s->P.resize(nv); for (int i = 0; i < nv; ++i) s->P[i] = v3(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
  s->N.clear(); s->UV.clear(); s->S.clear();
  if (N) { s->N.resize(nv); for (int i = 0; i < nv; ++i) s->N[i] = v3(N[3 * i], N[3 * i + 1], N[3 * i + 2]); }
  if (S) { s->S.resize(nv); for (int i = 0; i < nv; ++i) s->S[i] = v3(S[3 * i], S[3 * i + 1], S[3 * i + 2]); }
  if (UV) { s->UV.resize(nv); for (int i = 0; i < nv; ++i) s->UV[i] = P2{UV[2 * i], UV[2 * i + 1]}; }
  s->idx.assign(idx, idx + 3 * (size_t)nt);
  s->tri_material.assign(tri_material, tri_material + nt);
  s->tri_light.assign(tri_light, tri_light + nt);
  s->tri_flags.assign(tri_flags, tri_flags + nt);
  for (int i = 0; i < nt; ++i) {
    uint8_t f = s->tri_flags[i];
    if ((f & 2) && !N) return -1;
    if ((f & 4) && !UV) return -1;
    if ((f & 8) && !S) return -1;
  }
  return 0;
}
int orc_scene_set_alpha(void* h, const int32_t* tri_alpha2) {  // after orc_scene_set_mesh: per triangle {alpha, shadowalpha} float-texture ids or -1
  Scene* s = (Scene*)h;
  if (!tri_alpha2) s->tri_alpha.clear(); else s->tri_alpha.assign(tri_alpha2, tri_alpha2 + 2 * s->n_tris());
  return 0;
}
// Shape "sphere" (Sphere::create, sphere.rs:53-68), after orc_scene_set_mesh. Returns its primitive id (n_tris + index): what an area light's `tri` names.
// kind: 0 sphere (a = zmin, b = zmax), 1 disk (a = height, b = innerradius; disk.rs:48-62), 2 cylinder (a = z_min, b = z_max; cylinder.rs:26-46)
int orc_scene_add_sphere(void* h, const float* o2w16, const float* w2o16, float radius, float a, float b, float phi_max, int reverse_orientation,
                         int material, int light, int kind) {
  Scene* s = (Scene*)h;
  if (!s->instances.empty()) return -1;  // primitive ids: triangles, spheres, instances - in this order
  Transform t; memcpy(t.m.m, o2w16, 64); memcpy(t.m_inv.m, w2o16, 64);
  Sphere sp = kind == QUADRIC_DISK ? disk_new(t, a, radius, b, phi_max, reverse_orientation != 0)
            : kind == QUADRIC_CYLINDER ? cylinder_new(t, radius, a, b, phi_max, reverse_orientation != 0) : sphere_new(t, radius, a, b, phi_max, reverse_orientation != 0);
  s->spheres.push_back(sp);
  s->tri_material.push_back(material); s->tri_light.push_back(light);
  s->tri_flags.push_back((uint8_t)((sp.reverse_orientation != sp.swaps_handedness) ? 1 : 0));
  return (int)(s->n_prims() - 1);
}
// ObjectBegin ... ObjectEnd (api.rs:1019-1051): a triangle mesh in object space (vertices as the shapes' own CTMs leave them); tri_material indexes the
// scene's materials. Returns the object's index.
int orc_scene_add_object(void* h, const float* P, int nv, const int32_t* idx, int nt, const float* N, const float* UV, const float* S,
                         const int32_t* tri_material, const uint8_t* tri_flags) {
  Scene* s = (Scene*)h;
  auto o = std::make_shared<Scene>();
  std::vector<int32_t> no_light((size_t)nt, -1);
  if (nt < 0 || (nt > 0 && orc_scene_set_mesh(o.get(), P, nv, idx, nt, N, UV, S, tri_material, no_light.data(), tri_flags) != 0)) return -1;  // (nt == 0: an object of quadrics only)
  s->objects.push_back(o);
  return (int)s->objects.size() - 1;
}
// A triangle of an object that sits under an AreaLightSource: GeometricPrimitive::area_light is set (api.rs:934-945), the light itself is dropped with
// `area_lights` when there is a current instance (api.rs:954-964) - it is in the OBJECT's light table here, which nothing samples.
int orc_scene_object_emitter(void* h, int object, int tri, const float* rgbv, int two_sided) {
  Scene* s = (Scene*)h;
  if (object < 0 || (size_t)object >= s->objects.size()) return -1;
  Scene& o = *s->objects[(size_t)object];
  if (tri < 0 || (size_t)tri >= o.tri_light.size()) return -1;
  Light l{}; l.kind = LIGHT_DIFFUSE_AREA; l.l_emit = rgb(rgbv[0], rgbv[1], rgbv[2]); l.two_sided = two_sided != 0; l.tri = tri;
  o.lights.push_back(l);
  o.tri_light[(size_t)tri] = (int32_t)o.lights.size() - 1;
  return 0;
}
// The object definition as a scene of its own: what else ObjectBegin ... ObjectEnd collected (api.rs:1019-1051 pushes EVERY primitive) is added to it through the calls
// a top-level scene takes - orc_scene_add_sphere (a quadric under the CTM inside the definition; light: -1, or the index orc_scene_object_quadric_emitter returns),
// orc_scene_set_alpha (masks on its triangles; the texture ids name the TOP-LEVEL scene's textures: orc_scene_commit hands the tables on).
void* orc_scene_object_handle(void* h, int object) {
  Scene* s = (Scene*)h;
  return (object < 0 || (size_t)object >= s->objects.size()) ? nullptr : (void*)s->objects[(size_t)object].get();
}
// an unlisted emitter for a quadric of an object (before orc_scene_add_sphere names it as the quadric's light): returns its index in the OBJECT's light table
int orc_scene_object_quadric_emitter(void* h, int object, const float* rgbv, int two_sided) {
  Scene* s = (Scene*)h;
  if (object < 0 || (size_t)object >= s->objects.size()) return -1;
  Scene& o = *s->objects[(size_t)object];
  Light l{}; l.kind = LIGHT_DIFFUSE_AREA; l.l_emit = rgb(rgbv[0], rgbv[1], rgbv[2]); l.two_sided = two_sided != 0; l.tri = (int)o.n_prims();  // (the quadric added next)
  o.lights.push_back(l);
  return (int)o.lights.size() - 1;
}
// ObjectInstance (api.rs:1053-1090): TransformedPrimitive{object, primitive_to_world = the CTM}. Returns the primitive id. After every sphere.
int orc_scene_add_instance(void* h, int object, const float* o2w16, const float* w2o16) {
  Scene* s = (Scene*)h;
  if (object < 0 || (size_t)object >= s->objects.size()) return -1;
  Scene::Instance in; in.object = object; memcpy(in.o2w.m, o2w16, 64); memcpy(in.w2o.m, w2o16, 64);
  s->instances.push_back(in);
  s->tri_material.push_back(-1); s->tri_light.push_back(-1); s->tri_flags.push_back(0);
  return (int)(s->n_prims() - 1);
}
int orc_scene_add_mipmap(void* h, int w, int hgt, const float* rgbdata, int trilinear, float max_aniso, int wrap) {
  Scene* s = (Scene*)h;
  if (w <= 0 || hgt <= 0) return -1;
  auto m = std::make_shared<MipMap>();
  std::vector<RGB> img((size_t)w * hgt);
  for (size_t i = 0; i < img.size(); ++i) img[i] = rgb(rgbdata[3 * i], rgbdata[3 * i + 1], rgbdata[3 * i + 2]);
  m->init(w, hgt, img.data(), trilinear != 0, max_aniso, wrap);
  s->mips.push_back(m);
  return (int)s->mips.size() - 1;
}
// Texture::evaluate at a hand-made SurfaceInteraction (tests): uv, p, (dudx, dvdx, dudy, dvdy), dpdx, dpdy.
void orc_tex_probe(void* h, int tex, const float* uv2, const float* p3, const float* duv4, const float* dpdx3, const float* dpdy3, float* out3) {
  Scene* s = (Scene*)h;
  SurfaceInteraction si;
  si.uv = P2{uv2[0], uv2[1]}; si.hit.p = v3(p3[0], p3[1], p3[2]);
  si.dudx = duv4[0]; si.dvdx = duv4[1]; si.dudy = duv4[2]; si.dvdy = duv4[3];
  si.dpdx = v3(dpdx3[0], dpdx3[1], dpdx3[2]); si.dpdy = v3(dpdy3[0], dpdy3[1], dpdy3[2]);
  RGB c = s->tex_eval(tex, si);
  out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}
float orc_noise(float x, float y, float z) { return noise_perlin(x, y, z); }
// Inspection (tests): level `level` of MIP pyramid `mip`; returns the number of levels. rgb_out may be NULL.
int orc_mip_level(void* h, int mip, int level, int* w, int* hgt, float* rgb_out) {
  Scene* s = (Scene*)h;
  if (mip < 0 || mip >= (int)s->mips.size()) return -1;
  const MipMap& m = *s->mips[mip];
  if (level < 0 || level >= m.levels()) return -1;
  const MipLevel& l = m.pyramid[level];
  *w = l.u; *hgt = l.v;
  if (rgb_out) for (size_t i = 0; i < l.data.size(); ++i) { rgb_out[3 * i] = l.data[i].r; rgb_out[3 * i + 1] = l.data[i].g; rgb_out[3 * i + 2] = l.data[i].b; }
  return m.levels();
}
// `&Transform * &Point3f` / `* &Vector3f` (transform.rs:264-303) for a translation (reference KAT: ray.rs:120-128)
void orc_translate_apply(const float* delta3, const float* in3, int is_vector, float* out3) {
  M44 m = xf_translate(v3(delta3[0], delta3[1], delta3[2])).m;
  V3 r = is_vector ? xf_vector(m, v3(in3[0], in3[1], in3[2])) : xf_point(m, v3(in3[0], in3[1], in3[2]));
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
int orc_round_up_pow2(int v) { return MipMap::round_up_pow2(v); }  // lib.rs:215-224 (reference KAT: lib.rs:341-345)
int orc_scene_add_texture(void* h, int kind, const float* value, int tex1, int tex2, int amount, int mip, const float* mapping) {
  Scene* s = (Scene*)h;
  Texture t; t.kind = kind; t.value = rgb(value[0], value[1], value[2]); t.tex1 = tex1; t.tex2 = tex2; t.amount = amount; t.mip = mip;
  if (mapping) { t.su = mapping[0]; t.sv = mapping[1]; t.du = mapping[2]; t.dv = mapping[3]; }
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}
// params: kd ks kr kt sigma roughness urough vrough eta k opacity reflect transmit amount m1 m2
int orc_scene_add_material(void* h, int kind, const int32_t* p, int remap_roughness, int bump) {
  Scene* s = (Scene*)h;
  Material m; m.kind = kind;
  m.kd = p[0]; m.ks = p[1]; m.kr = p[2]; m.kt = p[3]; m.sigma = p[4]; m.roughness = p[5]; m.urough = p[6]; m.vrough = p[7];
  m.eta = p[8]; m.k = p[9]; m.opacity = p[10]; m.reflect = p[11]; m.transmit = p[12]; m.amount = p[13]; m.m1 = p[14]; m.m2 = p[15];
  m.remap_roughness = remap_roughness != 0; m.bump = bump;
  s->materials.push_back(m);
  return (int)s->materials.size() - 1;
}
// kind: LightKind. rgbv: L / I. vec: point position or distant direction (w_light = from - to, already in world space).
int orc_scene_add_light(void* h, int kind, int tri, const float* rgbv, int two_sided, const float* vec, int mip, const float* l2w, const float* w2l) {
  Scene* s = (Scene*)h;
  if (tri <= -2) tri = (int)s->n_tris() + (-2 - tri);  // -2 - k: sphere k (primitive ids of spheres follow the triangles')
  Light l; l.kind = kind; l.tri = tri; l.two_sided = two_sided != 0;
  l.l_emit = rgb(rgbv[0], rgbv[1], rgbv[2]); l.intensity = l.l_emit;
  if (vec) l.pos = v3(vec[0], vec[1], vec[2]);
  if (kind == LIGHT_DISTANT) l.pos = normalize(l.pos);  // distant.rs:27
  l.mip = mip;
  if (l2w) memcpy(l.l2w.m, l2w, 64);
  if (w2l) memcpy(l.w2l.m, w2l, 64);
  if (kind == LIGHT_INFINITE) {
    if (mip < 0 || mip >= (int)s->mips.size()) return -1;
    l.distribution = infinite_build_distribution(*s->mips[mip]);
  }
  s->lights.push_back(l);
  return (int)s->lights.size() - 1;
}
int orc_scene_commit(void* h, int max_prims_per_node) {
  Scene* s = (Scene*)h;
  s->max_prims_per_node = max_prims_per_node > 255 ? 255 : max_prims_per_node;
  for (auto& o : s->objects) if (!o->tri_alpha.empty()) { o->textures = s->textures; o->mips = s->mips; }  // an object's alpha masks are float textures of the scene
  s->build_bvh();
  s->preprocess_lights();
  return 0;
}
int orc_scene_bvh_sizes(void* h, int* n_nodes, int* n_prims) { Scene* s = (Scene*)h; *n_nodes = (int)s->nodes.size(); *n_prims = (int)s->ordered.size(); return 0; }
// nodes_out: n_nodes * 8 floats/ints: min xyz, max xyz, offset(u32 bits), (n_prims | axis<<16) bits
int orc_scene_bvh_get(void* h, float* bounds6, uint32_t* offset, uint16_t* n_prims, uint8_t* axis, int32_t* ordered) {
  Scene* s = (Scene*)h;
  for (size_t i = 0; i < s->nodes.size(); ++i) {
    const LinearNode& n = s->nodes[i];
    bounds6[6 * i + 0] = n.bounds.mn.x; bounds6[6 * i + 1] = n.bounds.mn.y; bounds6[6 * i + 2] = n.bounds.mn.z;
    bounds6[6 * i + 3] = n.bounds.mx.x; bounds6[6 * i + 4] = n.bounds.mx.y; bounds6[6 * i + 5] = n.bounds.mx.z;
    offset[i] = n.offset; n_prims[i] = n.n_prims; axis[i] = n.axis;
  }
  for (size_t i = 0; i < s->ordered.size(); ++i) ordered[i] = s->ordered[i];
  return 0;
}

// Batch tracing: rays = n * 8 floats (o.xyz, tmax, d.xyz, unused). closest: out_hit n*4 floats (t, prim(bits, -1 = miss), b0, b1).
// any-hit: out_hit[i*4] = 1.0 / 0.0.  counters: nodes, tris summed over the batch (u64[2]).
int orc_trace(void* h, const float* rays, int64_t n, int any_hit, float* out_hit, uint64_t* counters) {
  Scene* s = (Scene*)h;
  TraceCounters tc;
  for (int64_t i = 0; i < n; ++i) {
    const float* r = rays + 8 * i;
    Ray ray = ray_segment(v3(r[0], r[1], r[2]), v3(r[4], r[5], r[6]), r[3]);
    float* o = out_hit + 4 * i;
    if (any_hit) {
      o[0] = s->intersect_p(ray, &tc) ? 1.0f : 0.0f; o[1] = o[2] = o[3] = 0.0f;
    } else {
      int prim, osub = -1; TriHit th;
      if (s->intersect_raw(ray, &prim, &th, &tc, &osub)) { o[0] = th.t; o[1] = bits2f((uint32_t)s->hit_id(prim, osub)); o[2] = th.b0; o[3] = th.b1; }
      else { o[0] = kInf; o[1] = bits2f(0xffffffffu); o[2] = o[3] = 0.0f; }
    }
  }
  if (counters) { counters[0] = tc.nodes; counters[1] = tc.tris; }
  return 0;
}

// Sampler tables for one pixel (tests): out1d = dims*spp floats, out2d = dims*spp*2 floats. `seed` is the tile seed (REF) or pixel index (KEYED).
int orc_sampler_tables(int spp, int dims, int mode, uint64_t seed, float* out1d, float* out2d, uint64_t* rng_state_after) {
  ZeroTwoSequence z; z.init((uint32_t)spp, (uint32_t)dims, (SamplerMode)mode);
  if (mode == SAMPLER_REF) z.reseed(seed);
  z.start_pixel(seed);
  for (int d = 0; d < dims; ++d)
    for (uint32_t i = 0; i < z.spp; ++i) {
      out1d[(size_t)d * z.spp + i] = z.samples_1d[d][i];
      out2d[2 * ((size_t)d * z.spp + i)] = z.samples_2d[d][i].x; out2d[2 * ((size_t)d * z.spp + i) + 1] = z.samples_2d[d][i].y;
    }
  if (rng_state_after) { rng_state_after[0] = z.rng.state; rng_state_after[1] = z.rng.inc; }
  return (int)z.spp;
}
// Keyed pixels in [pixel0, pixel0 + n) whose start_pixel stream hits at least one uniform_u32_bounded retry
// (rc/rng.rs:32-40: r < (!b+1)&b). Found by replaying the draw sequence of start_pixel
// (rc/sampler/zerotwosequence.rs:67-108 -> lowdiscrepancy.rs:4-50) without building the tables. Tests use these
// rare pixels to exercise the GPU sampler's retry path.
int orc_sampler_retry_scan(int spp_, int dims, uint64_t pixel0, uint64_t n, uint64_t* out, int cap) {
  uint32_t spp = 1; while (spp < (uint32_t)spp_) spp <<= 1;
  int found = 0;
  for (uint64_t k = 0; k < n && found < cap; ++k) {
    Rng r; r.set_sequence(pixel0 + k);
    bool retried = false;
    for (int t = 0; t < 2 * dims && !retried; ++t) {
      r.uniform_u32(); if (t >= dims) r.uniform_u32();
      for (int half = 0; half < 2; ++half)
        for (uint32_t i = 0; i < spp; ++i) {
          const uint32_t b = half == 0 ? 1u : spp - i, threshold = (~b + 1u) & b;
          if (r.uniform_u32() < threshold) retried = true;  // the reference would draw again here: the stream shifts
        }
    }
    if (retried) out[found++] = pixel0 + k;
  }
  return found;
}
// Raw PCG32 stream (tests): seq < 0 => default-constructed generator
int orc_rng_stream(int64_t seq, int n, uint32_t* out_u32, float* out_f32) {
  Rng r; if (seq >= 0) r.set_sequence((uint64_t)seq);
  Rng r2 = r;
  for (int i = 0; i < n; ++i) { out_u32[i] = r.uniform_u32(); if (out_f32) out_f32[i] = r2.uniform_f32(); }
  return 0;
}
uint32_t orc_rng_bounded(int64_t seq, uint32_t b, int n_skip) { Rng r; r.set_sequence((uint64_t)seq); for (int i = 0; i < n_skip; ++i) r.uniform_u32(); return r.uniform_u32_bounded(b); }
float orc_radical_inverse(int base_index, uint64_t a) { return radical_inverse((uint32_t)base_index, a); }
// Known-answer helpers for the reference's own unit tests
int orc_distribution1d_sample_discrete(const float* f, int n, float u, float* pdf) {
  Distribution1D d(f, (size_t)n); size_t off; d.sample_discrete(u, &off, pdf); return (int)off;
}
int orc_distribution1d_get(const float* f, int n, float* cdf_out, float* func_int) {
  Distribution1D d(f, (size_t)n); for (size_t i = 0; i < d.cdf.size(); ++i) cdf_out[i] = d.cdf[i]; *func_int = d.func_int; return 0;
}
int orc_find_interval_array(const float* a, int n, float x) { return (int)find_interval((size_t)n, [&](size_t i) { return a[i] <= x; }); }
// Bounds2i iteration order (rc/bounds.rs:382-420): the iterator's own state machine
int orc_bounds2i_iter(int x0, int y0, int x1, int y1, int32_t* out_xy, int cap) {
  int n = 0;
  int px = x0 - 1, py = y0;  // into_iter(): "start 1 before p_min.x"
  for (;;) {
    if (x1 <= x0 || y1 <= y0) break;  // degenerate bounds yield nothing
    px += 1;
    if (px == x1) { px = x0; py += 1; }
    if (py >= y1) break;
    if (n < cap) { out_xy[2 * n] = px; out_xy[2 * n + 1] = py; }
    ++n;
  }
  return n;
}
// pcg32_srandom_r(initstate, initseq) + n draws with the generator step of rc/rng.rs:23-30. RNG::set_sequence(seq)
// is pcg32_srandom_r(PCG32_DEFAULT_STATE, seq); the published PCG32 demo vector uses (42, 54).
int orc_pcg32_srandom_stream(uint64_t initstate, uint64_t initseq, int n, uint32_t* out) {
  Rng r; r.state = 0; r.inc = (initseq << 1u) | 1u;
  (void)r.uniform_u32(); r.state += initstate; (void)r.uniform_u32();
  for (int i = 0; i < n; ++i) out[i] = r.uniform_u32();
  return 0;
}
// Single triangle test (tests): tri9 = p0 p1 p2, ray8 = o, tmax, d, pad -> out4 = t b0 b1 b2; returns 1 on hit
int orc_tri_intersect(const float* tri9, const float* ray8, float* out4) {
  Scene s; s.P = {v3(tri9[0], tri9[1], tri9[2]), v3(tri9[3], tri9[4], tri9[5]), v3(tri9[6], tri9[7], tri9[8])};
  s.idx = {0, 1, 2}; s.tri_material = {0}; s.tri_light = {-1}; s.tri_flags = {0};
  Ray ray = ray_segment(v3(ray8[0], ray8[1], ray8[2]), v3(ray8[4], ray8[5], ray8[6]), ray8[3]);
  TriHit h;
  if (!s.tri_test(0, ray, &h)) return 0;
  out4[0] = h.t; out4[1] = h.b0; out4[2] = h.b1; out4[3] = h.b2;
  return 1;
}
// The reference's sphere property test (rustracer-core/tests/shapes.rs:16-54) transplanted to triangles:
// hit the triangle, spawn a ray from the hit in direction w and re-test. Returns -1 original miss, 0 no re-hit, 1 re-hit.
int orc_tri_reintersect(const float* tri9, const float* ray8, const float* w3) {
  Scene s; s.P = {v3(tri9[0], tri9[1], tri9[2]), v3(tri9[3], tri9[4], tri9[5]), v3(tri9[6], tri9[7], tri9[8])};
  s.idx = {0, 1, 2}; s.tri_material = {0}; s.tri_light = {-1}; s.tri_flags = {0};
  Ray ray = ray_segment(v3(ray8[0], ray8[1], ray8[2]), v3(ray8[4], ray8[5], ray8[6]), ray8[3]);
  TriHit h;
  if (!s.tri_test(0, ray, &h)) return -1;
  SurfaceInteraction si; s.tri_fill_interaction(0, ray, h, &si);
  V3 w = v3(w3[0], w3[1], w3[2]);
  if (dot(w, si.hit.n) * dot(-ray.d, si.hit.n) < 0.0f) w = -w;  // leave on the side the ray came from
  Ray r2 = spawn_ray(si.hit, w);
  TriHit h2;
  return s.tri_test(0, r2, &h2) ? 1 : 0;
}
// The reference's own quadric property test, rustracer-core/tests/shapes.rs:16-54 (`full_sphere_reintersect` -> `test_reintersection_convex`), on the oracle's
// Sphere: a full sphere of `radius` at the origin (Transform::default), the ray ray8 = (o, t_max, d, -); if it hits, `n` rays leave the hit point in the
// directions uniform_sample_sphere(u[i]) flipped into the hemisphere of the hit's normal (isect.spawn_ray(&w)) and neither intersect_p nor intersect may find
// the sphere again. Returns -1 when the first ray misses, else the number of re-hits (intersect_p + intersect); the spawned rays go to rays_out (n x 8) when given.
int orc_sphere_reintersect(float radius, const float* ray8, const float* u2, int n, float* rays_out) {
  const Sphere s = sphere_new(Transform{m44_identity(), m44_identity()}, radius, -radius, radius, 360.0f, false);
  const Ray ray = ray_segment(v3(ray8[0], ray8[1], ray8[2]), v3(ray8[4], ray8[5], ray8[6]), ray8[3]);
  SphereHit h;
  if (!sphere_intersect(s, ray, true, &h)) return -1;
  Interaction it; it.p = h.p; it.p_error = h.p_error; it.n = h.n; it.wo = h.wo;
  int rehits = 0;
  for (int i = 0; i < n; ++i) {
    V3 w = uniform_sample_sphere(P2{u2[2 * i], u2[2 * i + 1]});
    if (dot(w, h.n) < 0.0f) w = -w;
    const Ray out = spawn_ray(it, w);
    if (rays_out) { float* r = rays_out + 8 * (size_t)i; r[0] = out.o.x; r[1] = out.o.y; r[2] = out.o.z; r[3] = out.t_max; r[4] = out.d.x; r[5] = out.d.y; r[6] = out.d.z; r[7] = 0.0f; }
    SphereHit h2;
    if (sphere_intersect(s, out, false, &h2)) rehits += 1;  // intersect_p
    if (sphere_intersect(s, out, true, &h2)) rehits += 1;   // intersect
  }
  return rehits;
}
// EFloat arithmetic for rustracer-core/tests/efloat.rs:51-154: op 0 abs, 1 sqrt, 2 add, 3 sub, 4 mul, 5 div on EFloat::new(av, aerr) (and EFloat::new(bv, berr));
// out3 = (v, lower_bound, upper_bound); in_bounds4 = the operands' own (low, high) pairs, which the test draws its exact values from
void orc_efloat_op(int op, float av, float aerr, float bv, float berr, float* out3, float* in_bounds4) {
  const EFloat a = ef_new(av, aerr), b = ef_new(bv, berr);
  EFloat r = a;
  switch (op) { case 0: r = ef_abs(a); break; case 1: r = ef_sqrt(a); break; case 2: r = a + b; break; case 3: r = a - b; break; case 4: r = a * b; break; default: r = a / b; }
  out3[0] = r.v; out3[1] = r.low; out3[2] = r.high;
  in_bounds4[0] = a.low; in_bounds4[1] = a.high; in_bounds4[2] = b.low; in_bounds4[3] = b.high;
}
// BSDF probes (tests): evaluates f / pdf / sample_f of the Bsdf a material builds at a canonical hit (n = +z)
int orc_bsdf_probe(void* h, int material, const float* wo3, const float* wi3, const float* u2, float* f3_out, float* pdf_out, float* sample_out8) {
  Scene* s = (Scene*)h;
  SurfaceInteraction si; si.hit = Interaction{v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 1), v3(0, 0, 1)}; si.uv = P2{0.5f, 0.5f};
  si.dpdu = v3(1, 0, 0); si.dpdv = v3(0, 1, 0); si.shading.n = v3(0, 0, 1); si.shading.dpdu = v3(1, 0, 0); si.shading.dpdv = v3(0, 1, 0);
  Bsdf b; s->build_bsdf(material, si, &b);
  V3 wo = v3(wo3[0], wo3[1], wo3[2]), wi = v3(wi3[0], wi3[1], wi3[2]);
  RGB f = b.f(wo, wi, BSDF_ALL); f3_out[0] = f.r; f3_out[1] = f.g; f3_out[2] = f.b;
  *pdf_out = b.pdf(wo, wi, BSDF_ALL);
  SampleF sf = b.sample_f(wo, P2{u2[0], u2[1]}, BSDF_ALL);
  sample_out8[0] = sf.f.r; sample_out8[1] = sf.f.g; sample_out8[2] = sf.f.b; sample_out8[3] = sf.wi.x; sample_out8[4] = sf.wi.y; sample_out8[5] = sf.wi.z;
  sample_out8[6] = sf.pdf; sample_out8[7] = (float)sf.type;
  return b.n;
}
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
float orc_gamma(int n) { return gamma_n((uint32_t)n); }

// LookAt (rc/transform.rs:119-154): m = world->camera, m_inv = camera->world
int orc_look_at(const float* pos, const float* look, const float* up, float* m, float* m_inv) {
  Transform t = xf_look_at(v3(pos[0], pos[1], pos[2]), v3(look[0], look[1], look[2]), v3(up[0], up[1], up[2]));
  memcpy(m, t.m.m, 64); memcpy(m_inv, t.m_inv.m, 64);
  return 0;
}
// Camera set-up products (tests): raster_to_camera (16), dx_camera(3), dy_camera(3), filter table (256), sample bounds (4: x0 y0 x1 y1)
int orc_camera_film_setup(const orc_render_params* p, float* r2c, float* dxdy, float* filter_table, int32_t* sample_bounds, int32_t* cropped) {
  Filter f; f.kind = p->filter_kind; f.xw = p->filter_params[0]; f.yw = p->filter_params[1]; f.a = p->filter_params[2]; f.b = p->filter_params[3];
  Film film; film.init(p->xres, p->yres, p->crop, f, p->film_scale, p->max_sample_luminance);
  Camera cam; Transform c2w; memcpy(c2w.m.m, p->cam_to_world, 64); memcpy(c2w.m_inv.m, p->cam_to_world_inv, 64);
  float sw[4]; Camera::default_screen_window(p->xres, p->yres, sw);
  if (p->screen_window[1] > p->screen_window[0]) memcpy(sw, p->screen_window, 16);
  cam.init(c2w, sw, p->lens_radius, p->focal_distance, p->fov, p->xres, p->yres);
  memcpy(r2c, cam.raster_to_camera.m.m, 64);
  dxdy[0] = cam.dx_camera.x; dxdy[1] = cam.dx_camera.y; dxdy[2] = cam.dx_camera.z; dxdy[3] = cam.dy_camera.x; dxdy[4] = cam.dy_camera.y; dxdy[5] = cam.dy_camera.z;
  memcpy(filter_table, film.filter_table, 1024);
  B2i sb = film.sample_bounds();
  sample_bounds[0] = sb.x0; sample_bounds[1] = sb.y0; sample_bounds[2] = sb.x1; sample_bounds[3] = sb.y1;
  cropped[0] = film.cropped.x0; cropped[1] = film.cropped.y0; cropped[2] = film.cropped.x1; cropped[3] = film.cropped.y1;
  return 0;
}

// Dense dump of the spatial light distribution (tests): returns n_voxels; out_func = nvox*nl, out_cdf = nvox*(nl+1), out_int = nvox
int orc_light_distrib(void* h, int strategy, int32_t* n_voxels, float* out_func, float* out_cdf, float* out_int, int64_t max_voxels_out) {
  Scene* s = (Scene*)h;
  LightDistribution ld; ld.init(s, strategy == 1 ? "uniform" : "spatial");
  n_voxels[0] = (int)ld.n_voxels[0]; n_voxels[1] = (int)ld.n_voxels[1]; n_voxels[2] = (int)ld.n_voxels[2];
  if (ld.uniform) { n_voxels[0] = n_voxels[1] = n_voxels[2] = 0; return 0; }
  size_t nl = s->lights.size();
  int64_t total = (int64_t)ld.n_voxels[0] * ld.n_voxels[1] * ld.n_voxels[2];
  if (!out_func) return 0;
  int64_t lim = total < max_voxels_out ? total : max_voxels_out;
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      for (int64_t v = t; v < lim; v += nt) {
        int pi[3] = {(int)(v % ld.n_voxels[0]), (int)((v / ld.n_voxels[0]) % ld.n_voxels[1]), (int)(v / ((int64_t)ld.n_voxels[0] * ld.n_voxels[1]))};
        Distribution1D* d = ld.compute_distribution(pi);
        for (size_t j = 0; j < nl; ++j) out_func[v * nl + j] = d->func[j];
        for (size_t j = 0; j < nl + 1; ++j) out_cdf[v * (nl + 1) + j] = d->cdf[j];
        out_int[v] = d->func_int;
        delete d;
      }
    });
  for (auto& x : th) x.join();
  return 0;
}

// ---------------------------------------------------------------- render (rc/renderer.rs:22-143)
int orc_render(void* h, const orc_render_params* p, float* film_xyzw, orc_stats* stats_out) {
  Scene* s = (Scene*)h;
  Filter f; f.kind = p->filter_kind; f.xw = p->filter_params[0]; f.yw = p->filter_params[1]; f.a = p->filter_params[2]; f.b = p->filter_params[3];
  Film film; film.init(p->xres, p->yres, p->crop, f, p->film_scale, p->max_sample_luminance);
  Camera cam; Transform c2w; memcpy(c2w.m.m, p->cam_to_world, 64); memcpy(c2w.m_inv.m, p->cam_to_world_inv, 64);
  float sw[4]; Camera::default_screen_window(p->xres, p->yres, sw);
  if (p->screen_window[1] > p->screen_window[0]) memcpy(sw, p->screen_window, 16);
  cam.init(c2w, sw, p->lens_radius, p->focal_distance, p->fov, p->xres, p->yres);

  auto t0 = std::chrono::steady_clock::now();
  LightDistribution distrib;  // integrator.preprocess, renderer.rs:30
  distrib.init(s, p->light_strategy == 1 ? "uniform" : "spatial");
  PathIntegrator integ; integ.scene = s; integ.distrib = &distrib;
  integ.max_depth = (int)(uint8_t)p->max_depth; integ.rr_threshold = p->rr_threshold; integ.mis_mode = p->mis_mode;
  const B2i sample_bounds = film.sample_bounds();
  B2i pixel_bounds = sample_bounds;  // path.rs:53-69
  if (p->has_pixel_bounds) {
    B2i pb{p->pixel_bounds[0], p->pixel_bounds[2], p->pixel_bounds[1], p->pixel_bounds[3]};
    pixel_bounds = B2i{max_po(pixel_bounds.x0, pb.x0), max_po(pixel_bounds.y0, pb.y0), min_po(pixel_bounds.x1, pb.x1), min_po(pixel_bounds.y1, pb.y1)};
  }
  integ.pixel_bounds = pixel_bounds;
  const int bs = p->tile_size > 0 ? p->tile_size : 16;
  const int ext_x = sample_bounds.x1 - sample_bounds.x0, ext_y = sample_bounds.y1 - sample_bounds.y0;
  const int ntx = (ext_x + bs - 1) / bs, nty = (ext_y + bs - 1) / bs;
  const int n_tiles = (ntx > 0 && nty > 0) ? ntx * nty : 0;
  std::atomic<int> next_tile{0};  // Mutex<iterator> over tiles in row-major order (:47,68-71)
  int nthreads = p->n_threads > 0 ? p->n_threads : (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  PathStats total; std::mutex stats_mtx;
  auto worker = [&]() {
    ZeroTwoSequence sampler; sampler.init((uint32_t)p->spp, (uint32_t)p->sampler_dims, (SamplerMode)p->sampler_mode);
    PathStats st;
    for (;;) {
      int tile = next_tile.fetch_add(1);
      if (tile >= n_tiles) break;
      int tx = tile % ntx, ty = tile / ntx;
      if (sampler.mode == SAMPLER_REF) sampler.reseed((uint64_t)(ty * ntx + tx));  // :83-84
      int x0 = sample_bounds.x0 + tx * bs, x1 = min_po(x0 + bs, sample_bounds.x1);
      int y0 = sample_bounds.y0 + ty * bs, y1 = min_po(y0 + bs, sample_bounds.y1);
      B2i tb{x0, y0, x1, y1};
      FilmTile ft; ft.init(film, tb);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
          uint64_t pixel_index = (uint64_t)(y - sample_bounds.y0) * (uint64_t)ext_x + (uint64_t)(x - sample_bounds.x0);
          // renderer.rs:96-104 starts every pixel and then skips those outside pixel_bounds: in the tile-sequential mode the skipped pixel's draws
          // advance the tile's RNG stream; in the pixel-keyed mode the next pixel re-keys it, so the tables of a skipped pixel are never observed
          if (sampler.mode != SAMPLER_REF && !b2i_inside_exclusive(pixel_bounds, x, y)) continue;
          sampler.start_pixel(pixel_index);
          if (!b2i_inside_exclusive(pixel_bounds, x, y)) continue;
          for (;;) {
            P2 o = sampler.get_2d();  // get_camera_sample, zerotwosequence.rs:182-192
            P2 p_film{(float)x + o.x, (float)y + o.y};
            (void)sampler.get_1d();
            P2 p_lens = sampler.get_2d();
            Ray ray = cam.generate_ray_differential(p_film, p_lens);
            scale_differentials(ray, 1.0f / sqrtf((float)sampler.spp));
            st.camera_rays += 1;
            RGB c = integ.li(ray, sampler, st);
            bool bad = false;
            if (has_nan(c)) { c = rgb(0, 0, 0); bad = true; }             // :115-126
            if (lum_y(c) < -1e-5f) { c = rgb(0, 0, 0); bad = true; }
            if (std::isinf(lum_y(c))) { c = rgb(0, 0, 0); bad = true; }
            if (bad) st.scrubbed += 1;
            ft.add_sample(p_film.x, p_film.y, c);
            if (!sampler.start_next_sample()) break;
          }
        }
      film_merge_tile(film, ft);
    }
    std::lock_guard<std::mutex> lk(stats_mtx);
    total.add(st);
  };
  std::vector<std::thread> threads;
  for (int i = 0; i < nthreads; ++i) threads.emplace_back(worker);
  for (auto& t : threads) t.join();
  auto t1 = std::chrono::steady_clock::now();

  if (film_xyzw)
    for (size_t i = 0; i < film.pixels.size(); ++i) {
      film_xyzw[4 * i] = film.pixels[i].xyz[0]; film_xyzw[4 * i + 1] = film.pixels[i].xyz[1]; film_xyzw[4 * i + 2] = film.pixels[i].xyz[2];
      film_xyzw[4 * i + 3] = film.pixels[i].weight;
    }
  if (stats_out) {
    orc_stats& o = *stats_out;
    o.camera_rays = total.camera_rays; o.rays_closest = total.closest.rays_closest; o.rays_shadow = total.shadow.rays_any; o.rays_mis = total.mis.rays_closest;
    o.nodes_closest = total.closest.nodes; o.nodes_shadow = total.shadow.nodes; o.nodes_mis = total.mis.nodes;
    o.tris_closest = total.closest.tris; o.tris_shadow = total.shadow.tris; o.tris_mis = total.mis.tris;
    o.pdf_wi_tests = total.pdf_wi_tests; o.zero_radiance = total.zero_radiance; o.nee_total = total.nee_total; o.path_len_sum = total.path_len_sum;
    o.scrubbed = total.scrubbed;
    o.seconds = std::chrono::duration<double>(t1 - t0).count();
  }
  return 0;
}

// Film XYZW -> linear RGB exactly as Film::write_image (rc/film.rs:196-234)
int orc_film_to_rgb(const float* film_xyzw, int64_t n_pixels, float scale, float* rgb_out) {
  for (int64_t i = 0; i < n_pixels; ++i) {
    FilmPixel px{{film_xyzw[4 * i], film_xyzw[4 * i + 1], film_xyzw[4 * i + 2]}, film_xyzw[4 * i + 3]};
    RGB c = film_pixel_rgb(px, scale);
    rgb_out[3 * i] = c.r; rgb_out[3 * i + 1] = c.g; rgb_out[3 * i + 2] = c.b;
  }
  return 0;
}

// Single-path probe (tests/debug): radiance of one (pixel, sample) in KEYED mode.
int orc_li_keyed(void* h, const orc_render_params* p, int px, int py, int sample, float* rgb_out) {
  Scene* s = (Scene*)h;
  Filter f; f.kind = p->filter_kind; f.xw = p->filter_params[0]; f.yw = p->filter_params[1]; f.a = p->filter_params[2]; f.b = p->filter_params[3];
  Film film; film.init(p->xres, p->yres, p->crop, f, p->film_scale, p->max_sample_luminance);
  Camera cam; Transform c2w; memcpy(c2w.m.m, p->cam_to_world, 64); memcpy(c2w.m_inv.m, p->cam_to_world_inv, 64);
  float sw[4]; Camera::default_screen_window(p->xres, p->yres, sw);
  if (p->screen_window[1] > p->screen_window[0]) memcpy(sw, p->screen_window, 16);
  cam.init(c2w, sw, p->lens_radius, p->focal_distance, p->fov, p->xres, p->yres);
  LightDistribution distrib; distrib.init(s, p->light_strategy == 1 ? "uniform" : "spatial");
  PathIntegrator integ; integ.scene = s; integ.distrib = &distrib; integ.max_depth = (int)(uint8_t)p->max_depth; integ.rr_threshold = p->rr_threshold;
  B2i sb = film.sample_bounds();
  ZeroTwoSequence sampler; sampler.init((uint32_t)p->spp, (uint32_t)p->sampler_dims, SAMPLER_KEYED);
  uint64_t pixel_index = (uint64_t)(py - sb.y0) * (uint64_t)(sb.x1 - sb.x0) + (uint64_t)(px - sb.x0);
  sampler.start_pixel(pixel_index);
  for (int i = 0; i < sample; ++i) sampler.start_next_sample();
  P2 o = sampler.get_2d(); P2 p_film{(float)px + o.x, (float)py + o.y};
  (void)sampler.get_1d(); P2 p_lens = sampler.get_2d();
  Ray ray = cam.generate_ray_differential(p_film, p_lens);
  scale_differentials(ray, 1.0f / sqrtf((float)sampler.spp));
  PathStats st;
  RGB c = integ.li(ray, sampler, st);
  rgb_out[0] = c.r; rgb_out[1] = c.g; rgb_out[2] = c.b;
  return 0;
}

}  // extern "C"
