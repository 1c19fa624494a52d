// rtx_shade_kernels.h - K3: the shade kernel k_shade<MODE, ...> and its Bsdf front-ends (split from rtx_kernels.h in round 6; compiled by rtx_shade.hip only).
#pragma once
#include "rtx_kernels.h"

namespace rtx {

// ================================================================================ K3 shade
struct PathSampler {  // ZeroTwoSequence::get_1d / get_2d (zerotwosequence.rs:158-180) for one (pixel, sample)
  Tables tb; unsigned pix, s; int c1, c2; Pcg32 rng;
  RT_DEV float get_1d() {
    if (c1 < (int)tb.dims) return table_1d(tb, pix, (unsigned)c1++, s);
    return rng.next_f32();
  }
  RT_DEV f2 get_2d() {
    if (c2 < (int)tb.dims) return table_2d(tb, pix, (unsigned)c2++, s);
    float x = rng.next_f32();
    float y = rng.next_f32();
    return mk2(y, x);  // (second draw, first draw), :174-179
  }
};

// ---- Bsdf front-ends. GenericBsdf is the tagged-lobe aggregate of rtx_dev_bsdf.h. SingleLambert is the
// same arithmetic specialised for the Bsdf a constant-texture matte material builds (one
// LambertianReflection lobe, or none when Kd is black): with one matching lobe the component choice,
// the u remap (u*1-0), the pdf average (/1) and the lobe sums (0+x) of Bsdf::{f,pdf,sample_f} are
// identities, so both front-ends return bit-identical values; the specialised one needs no lobe
// array in scratch and a fraction of the registers.
struct GenericBsdf {
  Bsdf b;
  RT_DEV void build(const DScene& sc, int mat, SurfaceInteraction& si) { build_bsdf(sc, mat, si, b); }  // a bump map rewrites si's shading geometry
  RT_DEV int num_nonspecular() const { return bsdf_num_components(b, BSDF_ALL & ~BSDF_SPECULAR); }
  RT_DEV rgb3 f(f3 wo, f3 wi, unsigned flags) const { return bsdf_f(b, wo, wi, flags); }
  RT_DEV float pdf(f3 wo, f3 wi, unsigned flags) const { return bsdf_pdf(b, wo, wi, flags); }
  RT_DEV LobeSample sample_f(f3 wo, f2 u, unsigned flags) const { return bsdf_sample_f(b, wo, u, flags); }
  RT_DEV float eta() const { return b.eta; }
};
template <bool TEXTURED, bool BOUNCED = false>  // TEXTURED: Kd may be any texture (evaluated out of line); false: constant Kd only, no call in the kernel; BOUNCED: the vertex is past the camera ray (no differentials: the inline level-0 bilinear lookup)
struct SingleLambertT {
  rgb3 r; bool has; f3 ns, ng, ss, ts;
  RT_DEV void build(const DScene& sc, int mat, SurfaceInteraction& si) {  // matte.rs:37-62 with sigma == 0 and no bump map
    const int kd = sc.materials[mat].slot[0];
    const DTexture& t = sc.textures[kd];
    if (TEXTURED && t.kind != RT_TEX_CONST && !RT_DBG(sc, 1)) r = clamp_pos(BOUNCED ? tex_eval_leaf_bounced(sc, kd, si) : tex_eval_leaf(sc, kd, si));
    else r = (TEXTURED && t.kind != RT_TEX_CONST) ? mkc(0.75f, 0.75f, 0.75f) : clamp_pos(mkc(t.v[0], t.v[1], t.v[2]));
    has = !is_black(r);
    ss = si.ssb; ns = si.sh_n; ng = si.hit.n; ts = cross(si.sh_n, ss);  // Bsdf::new, bsdf/mod.rs:77-91 (ssb = normalize(si.sh_dpdu))
  }
  RT_DEV f3 to_local(f3 v) const { return mk3(dot(v, ss), dot(v, ts), dot(v, ns)); }
  RT_DEV int num_nonspecular() const { return has ? 1 : 0; }
  RT_DEV rgb3 f(f3 wo_w, f3 wi_w, unsigned) const {
    f3 wo = to_local(wo_w);
    if (!has || wo.z == 0.0f) return mkc(0, 0, 0);
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
    return refl ? r * kInvPi : mkc(0, 0, 0);
  }
  RT_DEV float pdf(f3 wo_w, f3 wi_w, unsigned) const {
    if (!has) return 0.0f;
    f3 wo = to_local(wo_w);
    if (wo.z == 0.0f) return 0.0f;
    f3 wi = to_local(wi_w);
    return default_pdf(wo, wi);
  }
  RT_DEV LobeSample sample_f(f3 wo_w, f2 u, unsigned) const {
    if (!has) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
    f2 ur = mk2(fminf(u.x * 1.0f - 0.0f, kOneMinusEpsilon), u.y);
    f3 wo = to_local(wo_w);
    if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, BSDF_DIFFUSE | BSDF_REFLECTION);
    f3 wi = cosine_sample_hemisphere(ur);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    float pdf = default_pdf(wo, wi);
    if (pdf == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
    f3 wi_w = mk3(ss.x * wi.x + ts.x * wi.y + ns.x * wi.z, ss.y * wi.x + ts.y * wi.y + ns.y * wi.z, ss.z * wi.x + ts.z * wi.y + ns.z * wi.z);
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
    return mk_ls(refl ? r * kInvPi : mkc(0, 0, 0), wi_w, pdf, 0u);
  }
  RT_DEV float eta() const { return 1.0f; }
};
typedef SingleLambertT<false> SingleLambert;

// Register-resident front-end for the materials that build at most two lobes out of {Lambertian, Oren-Nayar, microfacet
// reflection, specular reflection}: matte (any sigma), plastic, metal, mirror, without bump map. Bsdf::{f, pdf, sample_f}
// (bsdf/mod.rs:94-251) restated over two named lobes; every lobe function is entered with its kind as a constant, so only that
// kind's code is instantiated. Same operations in the same order as GenericBsdf (sums start from the same zero, the same component
// choice and u remap), hence the same values.
template <bool WIDE, bool CONST_TEX = false>  // WIDE: + glass, substrate and the opaque uber form (FresnelSpecular, FresnelBlend, microfacet transmission; a Bsdf eta); CONST_TEX: every texture parameter is a constant (no out-of-line image lookup is instantiated)
struct SmallBsdfT {
  static RT_DEV rgb3 tc(const DScene& sc, int id, const SurfaceInteraction& si) { if (CONST_TEX) { const DTexture& t = sc.textures[id]; return mkc(t.v[0], t.v[1], t.v[2]); } return tex_eval_c(sc, id, si); }
  static RT_DEV float tcf(const DScene& sc, int id, const SurfaceInteraction& si) { return tc(sc, id, si).r; }
  f3 ns, ng, ss, ts; int n; Lobe l0, l1; float eta_;
  RT_DEV void add(const Lobe& l) { if (n == 0) l0 = l; else l1 = l; ++n; }
  RT_DEV void build(const DScene& sc, int mat, SurfaceInteraction& si) {
    const DMaterial& m = sc.materials[mat]; const int* s = m.slot;
    n = 0; l0 = lobe_zero(LB_LAMBERT_R); l1 = l0; eta_ = 1.0f;
    if (WIDE && m.kind == 4) {  // glass.rs:53-106, allow_multiple_lobes = true (path.rs:145): one FresnelSpecular lobe, or microfacet reflection + transmission
      eta_ = tcf(sc, s[8], si);
      float ur = tcf(sc, s[6], si), vr = tcf(sc, s[7], si);
      rgb3 r = tc(sc, s[2], si), t = tc(sc, s[3], si);
      if (!is_black(r) || !is_black(t)) {
        if (ur == 0.0f && vr == 0.0f) {
          Lobe l = lobe_zero(LB_FRESNEL_SPEC); l.r = r; l.t = t; l.eta_a = 1.0f; l.eta_b = eta_; add(l);
        } else {
          if (m.remap) { ur = tr_roughness_to_alpha(ur); vr = tr_roughness_to_alpha(vr); }
          if (!is_black(r)) add(mk_micro_r(r, ur, vr, FR_DIELECTRIC, 1.0f, eta_));
          if (!is_black(t)) add(mk_micro_t(r, ur, vr, 1.0f, eta_));  // passes `r` (glass.rs:97)
        }
      }
    } else if (WIDE && m.kind == 5) {  // uber.rs:63-126 where the host found opacity, Kr and Kt constant with 1 - opacity, Kr and Kt black: no specular lobe
      float e = tcf(sc, s[8], si);
      rgb3 op = clamp_pos(tc(sc, s[10], si));
      eta_ = e;
      rgb3 kd = op * clamp_pos(tc(sc, s[0], si));
      if (!is_black(kd)) add(mk_lambert(LB_LAMBERT_R, kd));
      rgb3 ks = op * clamp_pos(tc(sc, s[1], si));
      if (!is_black(ks)) {
        float ru = tcf(sc, s[6] >= 0 ? s[6] : s[5], si), rv = tcf(sc, s[7] >= 0 ? s[7] : s[5], si);
        if (m.remap) { ru = tr_roughness_to_alpha(ru); rv = tr_roughness_to_alpha(rv); }
        add(mk_micro_r(ks, ru, rv, FR_DIELECTRIC, 1.0f, e));
      }
    } else if (WIDE && m.kind == 6) {  // substrate.rs:43-71
      rgb3 d = clamp_pos(tc(sc, s[0], si)), sp = clamp_pos(tc(sc, s[1], si));
      float ru = tcf(sc, s[6], si), rv = tcf(sc, s[7], si);
      if (!is_black(d) || !is_black(sp)) {
        if (m.remap) { ru = tr_roughness_to_alpha(ru); rv = tr_roughness_to_alpha(rv); }
        Lobe l = lobe_zero(LB_FRESNEL_BLEND); l.r = d; l.t = sp; l.ax = ru; l.ay = rv; add(l);
      }
    } else if (m.kind == 0) {  // matte.rs:37-62
      rgb3 r = clamp_pos(tc(sc, s[0], si));
      float sigma = clampf(tcf(sc, s[4], si), 0.0f, 1.0f);
      if (!is_black(r)) {
        if (sigma == 0.0f) add(mk_lambert(LB_LAMBERT_R, r));
        else {  // OrenNayar::new, oren_nayar.rs:17-27
          Lobe l = lobe_zero(LB_OREN_NAYAR); l.r = r;
          float sigma_rad = sigma * (kPi / 180.0f);
          float sigma2 = sigma_rad * sigma_rad;
          l.ax = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
          l.ay = 0.45f * sigma2 / (sigma2 + 0.09f);
          add(l);
        }
      }
    } else if (m.kind == 1) {  // plastic.rs:45-75
      rgb3 kd = tc(sc, s[0], si), ks = tc(sc, s[1], si);
      if (!is_black(kd)) add(mk_lambert(LB_LAMBERT_R, kd));
      if (!is_black(ks)) {
        float rough = tcf(sc, s[5], si);
        if (m.remap) rough = tr_roughness_to_alpha(rough);
        add(mk_micro_r(ks, rough, rough, FR_DIELECTRIC, 1.5f, 1.0f));
      }
    } else if (m.kind == 2) {  // metal.rs:50-82
      float ur = tcf(sc, s[6] >= 0 ? s[6] : s[5], si), vr = tcf(sc, s[7] >= 0 ? s[7] : s[5], si);
      if (m.remap) { ur = tr_roughness_to_alpha(ur); vr = tr_roughness_to_alpha(vr); }
      Lobe l = mk_micro_r(mkc(1, 1, 1), ur, vr, FR_CONDUCTOR, 1.0f, 1.0f);
      l.t = tc(sc, s[8], si); l.k = tc(sc, s[9], si);
      add(l);
    } else {  // mirror.rs:30-48
      rgb3 R = clamp_pos(tc(sc, s[2], si));
      if (!is_black(R)) { Lobe l = lobe_zero(LB_SPEC_R); l.r = R; add(l); }
    }
    ss = si.ssb; ns = si.sh_n; ng = si.hit.n; ts = cross(si.sh_n, ss);  // Bsdf::new, bsdf/mod.rs:77-91 (ssb = normalize(si.sh_dpdu))
  }
  RT_DEV f3 to_local(f3 v) const { return mk3(dot(v, ss), dot(v, ts), dot(v, ns)); }
  // one lobe function entered with its kind as a compile-time constant
  template <class F> RT_DEV static auto with_kind(const Lobe& l, F fn) -> decltype(fn(l)) {
    Lobe c = l;
    switch (l.kind) {
      case LB_OREN_NAYAR: c.kind = LB_OREN_NAYAR; return fn(c);
      case LB_MICRO_R: c.kind = LB_MICRO_R; return fn(c);
      case LB_SPEC_R: c.kind = LB_SPEC_R; return fn(c);
      case LB_FRESNEL_SPEC: if (WIDE) { c.kind = LB_FRESNEL_SPEC; return fn(c); } break;
      case LB_FRESNEL_BLEND: if (WIDE) { c.kind = LB_FRESNEL_BLEND; return fn(c); } break;
      case LB_MICRO_T: if (WIDE) { c.kind = LB_MICRO_T; return fn(c); } break;
      default: break;
    }
    c.kind = LB_LAMBERT_R; return fn(c);
  }
  RT_DEV static rgb3 lf(const Lobe& l, f3 wo, f3 wi) { return with_kind(l, [&](const Lobe& c) { return lobe_f_inner(c, wo, wi); }); }
  RT_DEV static float lp(const Lobe& l, f3 wo, f3 wi) { return with_kind(l, [&](const Lobe& c) { return lobe_pdf_inner(c, wo, wi); }); }
  RT_DEV static LobeSample lsamp(const Lobe& l, f3 wo, f2 u) { return with_kind(l, [&](const Lobe& c) { return lobe_sample_inner<false>(c, wo, u); }); }
  RT_DEV int num(unsigned flags) const { return (n > 0 && lobe_matches(l0.kind, flags) ? 1 : 0) + (n > 1 && lobe_matches(l1.kind, flags) ? 1 : 0); }
  RT_DEV int num_nonspecular() const { return num(BSDF_ALL & ~BSDF_SPECULAR); }
  RT_DEV static bool admits(const Lobe& l, unsigned flags, bool refl) {
    const unsigned ty = lobe_type(l.kind);
    return ((ty & flags) == ty) && ((refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION)));
  }
  RT_DEV rgb3 f(f3 wo_w, f3 wi_w, unsigned flags) const {  // :94-111
    f3 wi = to_local(wi_w), wo = to_local(wo_w);
    if (wo.z == 0.0f) return mkc(0, 0, 0);
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
    rgb3 c = mkc(0, 0, 0);
    if (n > 0 && admits(l0, flags, refl)) c = c + lf(l0, wo, wi);
    if (n > 1 && admits(l1, flags, refl)) c = c + lf(l1, wo, wi);
    return c;
  }
  RT_DEV float pdf(f3 wo_w, f3 wi_w, unsigned flags) const {  // :113-136
    if (n == 0) return 0.0f;
    f3 wo = to_local(wo_w);
    if (wo.z == 0.0f) return 0.0f;
    f3 wi = to_local(wi_w);
    int matched = 0; float p = 0.0f;
    if (n > 0 && lobe_matches(l0.kind, flags)) { ++matched; p += lp(l0, wo, wi); }
    if (n > 1 && lobe_matches(l1.kind, flags)) { ++matched; p += lp(l1, wo, wi); }
    return matched == 0 ? 0.0f : p / (float)matched;
  }
  RT_DEV LobeSample sample_f(f3 wo_w, f2 u, unsigned flags) const {  // :138-251
    const bool m0 = n > 0 && lobe_matches(l0.kind, flags), m1 = n > 1 && lobe_matches(l1.kind, flags);
    const int m = (m0 ? 1 : 0) + (m1 ? 1 : 0);
    if (m == 0) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
    int comp_i = (int)f2u_sat(floorf(u.x * (float)m));
    if (comp_i > m - 1) comp_i = m - 1;
    const bool second = m0 ? (comp_i == 1) : true;  // the comp_i-th matching lobe
    const Lobe bx = second ? l1 : l0;
    const unsigned bty = lobe_type(bx.kind);
    f2 ur = mk2(fminf(u.x * (float)m - (float)comp_i, kOneMinusEpsilon), u.y);
    f3 wo = to_local(wo_w);
    if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, bty);
    LobeSample s = lsamp(bx, wo, ur);
    if (s.pdf == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
    f3 wi = s.wi;
    f3 wi_w = mk3(ss.x * wi.x + ts.x * wi.y + ns.x * wi.z, ss.y * wi.x + ts.y * wi.y + ns.y * wi.z, ss.z * wi.x + ts.z * wi.y + ns.z * wi.z);
    float pdf = s.pdf;
    if (!(bty & BSDF_SPECULAR) && m > 1) pdf += second ? lp(l0, wo, wi) : lp(l1, wo, wi);  // the other matching lobe
    if (m > 1) pdf /= (float)m;
    rgb3 fv = s.f;
    if (!(bty & BSDF_SPECULAR)) {
      bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
      fv = mkc(0, 0, 0);
      if (n > 0 && admits(l0, flags, refl)) fv = fv + lf(l0, wo, wi);
      if (n > 1 && admits(l1, flags, refl)) fv = fv + lf(l1, wo, wi);
    }
    return mk_ls(fv, wi_w, pdf, s.type);
  }
  RT_DEV float eta() const { return WIDE ? eta_ : 1.0f; }
};

// The SurfaceInteraction of a hit inside an object instance: the object-space interaction of the object's primitive, then SurfaceInteraction::transform
// (primitive_to_world), rc/interaction.rs:156-190. Returns the primitive's index in the scene's arrays (material, flags).
// OBJ_GENERAL: the object may hold quadrics (DScene::obj_general) - an instantiation of its own, so that scenes of plain objects keep the function they had (instances-10k:
// one function with both branches cost the generic shade launches 80 B of call frame and 12 %)
template <bool OBJ_GENERAL>
RT_DEVN int instance_fill_interaction(const DScene& sc, unsigned hit_id, float ox, float oy, float oz, float dx, float dy, float dz, float b0, float b1, float b2,
                                      SurfaceInteraction& si) {
  unsigned lo = 0, hi = sc.n_instances;  // the last instance whose id_base <= hit_id
  while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (sc.instances[mid].id_base <= hit_id) lo = mid; else hi = mid; }
  const DInstance& in = sc.instances[lo];
  const int gprim = (int)(in.prim_base + (hit_id - in.id_base));
  const f3 d_obj = xf34_vector(in.w2o, mk3(dx, dy, dz));  // Transform * Ray: the direction as a vector (the origin does not enter a triangle's interaction)
  TriHit th; th.t = 0.0f; th.b0 = b0; th.b1 = b1; th.b2 = b2;
  SurfaceInteraction s;
  if (OBJ_GENERAL && (tri_flags(sc.tri_p, gprim) & RT_FLAG_SPHERE)) {  // a quadric of the object (round 6): Sphere::intersect builds its interaction from the OBJECT-space ray, then SurfaceInteraction::transform
    (void)sphere_fill_interaction(sc.spheres[prim_sphere_index(sc.tri_p, gprim)], xf34_point(in.w2o, mk3(ox, oy, oz)), d_obj, s);
  } else tri_fill_interaction_inl(sc, gprim, d_obj, th, s);
  f3 perr;
  si.hit.p = xf34_point_with_error(in.o2w, s.hit.p, s.hit.p_error, perr); si.hit.p_error = perr;
  si.hit.wo = normalize(xf34_vector(in.o2w, s.hit.wo));
  si.hit.n = normalize(xf34_normal(in.w2o, s.hit.n));
  si.uv = s.uv;
  si.dpdu = xf34_vector(in.o2w, s.dpdu); si.dpdv = xf34_vector(in.o2w, s.dpdv);
  si.dudx = si.dvdx = si.dudy = si.dvdy = 0.0f; si.dpdx = si.dpdy = mk3(0, 0, 0);
  si.sh_n = normalize(xf34_normal(in.w2o, s.sh_n));
  si.sh_dpdu = xf34_vector(in.o2w, s.sh_dpdu); si.sh_dpdv = xf34_vector(in.o2w, s.sh_dpdv);
  si.sh_n = face_forward(si.sh_n, si.hit.n);
  si.ssb = normalize(si.sh_dpdu);
  si.prim = gprim;
  return gprim;
}

// MODE 0: any material / texture / light. MODE 1: every material is matte with constant Kd and
// sigma == 0 and every light is a DiffuseAreaLight (decided by the host from the material and light
// tables); no texture then reads the camera-ray differentials and the kernel makes no out-of-line call.
// MODE 3: matte materials with sigma == 0 and no bump map - Kd any texture - under any kind of light: the register-resident
// front-end with the generic light and texture functions. MODE 5: matte (any sigma), plastic, metal and mirror without bump map
// through SmallBsdfT<false>; MODE 6: glass, substrate and opaque uber as well, through SmallBsdfT<true> (the narrow kernel is 5 % faster on its classes).
// k_shade<1> is bound to FOUR waves per SIMD (round 4): with the vertex body a function of (entry, slot) it fits 128 VGPRs with nothing spilled (round 3: 84 B of
// scratch at four, 381 -> 404 ms). S1 shade 299 -> 264 ms, 1248 -> 1300 Msamples/s; S2 1403 -> 1432 (two interleaved rounds).
#ifndef RT_SHADE_MIN_WAVES
#define RT_SHADE_MIN_WAVES 4
#endif
#ifndef RT_SHADE0_MIN_WAVES
#define RT_SHADE0_MIN_WAVES 2
#endif
// Round 4: the plain forms of the textured front-ends are bound to THREE waves per SIMD. Round 3 measured that as a loss (k_shade<3> 207 -> 168 VGPRs with 40
// spilled: S4 shade 4074 -> 4166 ms); since the vertex body became a function of (entry, slot) the kernels need 187 / 207 / 214 VGPRs, and at 168 with 24 / 46 /
// 59 spilled dwords (48 / 112 / 128 B of scratch) the third wave now pays: S4 shade 3027 -> 2972 ms (k_shade<3>) and -> 2854 ms (k_shade<5 | 6>), 353.0 -> 356.3 /
// 363.1 Msamples/s, two interleaved rounds on one box (scripts/ab_bench.sh).
#ifndef RT_SHADE_GEN_MIN_WAVES  // the GENERAL forms of the register-resident front-ends (quadric / instance hits, masked emitters)
#define RT_SHADE_GEN_MIN_WAVES 2
#endif
#ifndef RT_SHADE_BOUNCED_MIN_WAVES
#define RT_SHADE_BOUNCED_MIN_WAVES 3
#endif
#ifndef RT_SHADE_LEAN_MIN_WAVES
#define RT_SHADE_LEAN_MIN_WAVES 3
#endif
#ifndef RT_SHADE56_MIN_WAVES  // the two-lobe front-ends under any light / texture (k_shade<5 | 6>, plain form)
#define RT_SHADE56_MIN_WAVES 3
#endif
#ifndef RT_SHADE3_MIN_WAVES  // the Lambert front-end under any light (k_shade<3>)
#define RT_SHADE3_MIN_WAVES 3
#endif
// GENERAL (generic front-end only): some emitter triangle carries an alpha mask, so Shape::pdf_wi's re-intersection evaluates it; such scenes shade
// every vertex through k_shade<0, true>, every other scene never instantiates the mask evaluator in a shade kernel.
// LEAN (front-ends 3 / 5 / 6): every light is a diffuse area light on a triangle and every texture a constant - what MODE 1 assumes, for the other material
// classes. No out-of-line light or texture evaluator is instantiated, so the kernel's allocation is its own: 155 / 168 / 168 VGPRs under a three-wave bound
// (4 / 12 spilled dwords in the two-lobe forms) instead of 208 / 230 / 236 at two waves.
// BOUNCED (front-end 3, launches of bounces >= 1): no vertex of the launch is a camera vertex, so no differentials exist, image maps are level-0 bilinear lookups
// (inline) and the light evaluators are taken inline too: 183 VGPRs of its own, 168 under the three-wave bound with 5 spilled dwords.
// QLIGHTS (with LEAN, round 4): the LEAN form for scenes whose area lights may sit on analytic spheres - what veach-mis.pbrt is. The launch holds vertices on
// TRIANGLES only (k_bin_count sends every quadric hit to the generic bin, DScene::route_quadric_hits) and the host has checked that no triangle reaches into
// an emitter sphere (rt_scene_create: sphere_lights_clear), so Sphere::sample_si and Sphere::pdf_wi only ever take their cone branches (sphere.rs:264-308,
// 325-333) - inlined here, no out-of-line evaluator, three waves per SIMD like the other LEAN forms. Round 3 shaded such scenes through the GENERAL forms:
// 256 VGPRs and 352 - 448 B of scratch.
// LDSREC: 0 = every table in HBM; 1 = the scene's shade / traversal records, lights, materials and textures in LDS (small scenes); 2 = materials and textures only;
// 3 = lights, materials, textures and image headers (the plain forms of scenes with few of each)
template <int MODE, bool GENERAL = false, bool LEAN = false, bool BOUNCED = false, bool QLIGHTS = false, int LDSREC = 0>
__global__ void __launch_bounds__(256, (MODE == 1 || LEAN || BOUNCED) ? ((LEAN || BOUNCED) ? (BOUNCED ? RT_SHADE_BOUNCED_MIN_WAVES : RT_SHADE_LEAN_MIN_WAVES) : RT_SHADE_MIN_WAVES) : (MODE == 3 && !GENERAL ? RT_SHADE3_MIN_WAVES : ((MODE == 5 || MODE == 6) && !GENERAL ? RT_SHADE56_MIN_WAVES : (MODE != 0 && GENERAL ? RT_SHADE_GEN_MIN_WAVES : RT_SHADE0_MIN_WAVES)))) k_shade(DScene sc, FrameParams fp, PassState ps) {
  // LDSREC (MODE 1, round 5): a scene of <= RT_SMALL_TRIS triangles and <= RT_LDS_LIGHTS emitters keeps its shade records, traversal records and light table in LDS
  // for the launch. k_shade<1> is busy issuing VALU instructions half of the time and waits for memory two thirds of a wave's life (SQ counters), yet neither ~10 % fewer
  // instructions nor an earlier scan load moved it - what it waits for is the texture path's address processing: ~40 vector memory instructions per vertex, the
  // gathers among them (a vertex's triangle, the picked light, the light's triangle: every lane its own address) served a few lanes per clock. Read from LDS they
  // do not go there at all. Same values, same arithmetic.
  __shared__ float4 s_rec[LDSREC == 1 ? 8 * RT_SMALL_TRIS : 1];
  __shared__ float4 s_trip[LDSREC == 1 ? 3 * RT_SMALL_TRIS : 1];
  __shared__ unsigned s_lights[(LDSREC == 1 || LDSREC == 3) ? RT_LDS_LIGHTS * (sizeof(DLight) / 4) : 1];
  __shared__ unsigned s_imgs[LDSREC == 3 ? RT_LDS_IMAGES * (sizeof(DImage) / 4) : 1];
  __shared__ unsigned s_mats[LDSREC ? RT_LDS_MATERIALS * (sizeof(DMaterial) / 4) : 1];
  __shared__ unsigned s_texs[LDSREC ? RT_LDS_TEXTURES * (sizeof(DTexture) / 4) : 1];
  // (round 6 also kept ONE environment light's marginal distribution - cdf, func, guide: 10 KB - in LDS for the plain forms: S4 shade 2368 -> 2378 ms, not kept; commit ec235f5)
  if (LDSREC) {
    if (LDSREC == 1) {
      for (unsigned k = threadIdx.x; k < 8u * sc.n_tris; k += blockDim.x) s_rec[k] = sc.tri_rec[k];
      for (unsigned k = threadIdx.x; k < 3u * sc.n_tris; k += blockDim.x) s_trip[k] = sc.tri_p[k];
    }
    if (LDSREC == 1 || LDSREC == 3) {
      const unsigned nl = (unsigned)sc.n_lights_all * (unsigned)(sizeof(DLight) / 4);
      for (unsigned k = threadIdx.x; k < nl; k += blockDim.x) s_lights[k] = ((const unsigned*)sc.lights)[k];
    }
    if (LDSREC == 3) for (unsigned k = threadIdx.x; k < (unsigned)sc.n_images * (unsigned)(sizeof(DImage) / 4); k += blockDim.x) s_imgs[k] = ((const unsigned*)sc.images)[k];
    for (unsigned k = threadIdx.x; k < (unsigned)sc.n_materials * (unsigned)(sizeof(DMaterial) / 4); k += blockDim.x) s_mats[k] = ((const unsigned*)sc.materials)[k];
    for (unsigned k = threadIdx.x; k < (unsigned)sc.n_textures * (unsigned)(sizeof(DTexture) / 4); k += blockDim.x) s_texs[k] = ((const unsigned*)sc.textures)[k];
    __syncthreads();
    if (LDSREC == 1) { sc.tri_rec = (const float4*)s_rec; sc.tri_p = (const float4*)s_trip; }
    if (LDSREC == 1 || LDSREC == 3) sc.lights = (const DLight*)s_lights;
    sc.materials = (const DMaterial*)s_mats; sc.textures = (const DTexture*)s_texs;
  }
  // The kernel's once-through streams (path records in and out, shadow / MIS ray records) with or without the non-temporal hint (SPtr, RT_NT_STREAMS): with it where the launch
  // also GATHERS from tables larger than a cache (triangle records, texels, environment rows - the hint keeps the streams from evicting them: S4 shade 2489 -> 2325 ms, S2
  // 22.0 -> 20.6, S3 117.6 -> 114.3); without it where every table sits in LDS and the streams are all the launch reads (LDSREC == 1: S1 shade 219 -> 234 ms WITH the hint).
  constexpr bool NTK = LDSREC != 1;
  const auto in_o = sp<NTK>(ps.in.o), in_d = sp<NTK>(ps.in.d), in_beta = sp<NTK>(ps.in.beta); const auto in_st = sp<NTK>(ps.in.st); const auto pfilm_ = sp<NTK>(ps.pfilm);
  const auto out_o = sp<NTK>(ps.out.o), out_d = sp<NTK>(ps.out.d), out_beta = sp<NTK>(ps.out.beta); const auto out_st = sp<NTK>(ps.out.st);
  const auto sh_o = sp<NTK>(ps.sh.o), sh_d = sp<NTK>(ps.sh.d), sh_add = sp<NTK>(ps.sh.add);
  const auto mi_o = sp<NTK>(ps.mi.o), mi_d = sp<NTK>(ps.mi.d), mi_a = sp<NTK>(ps.mi.a), mi_b = sp<NTK>(ps.mi.b), mi_c = sp<NTK>(ps.mi.c); const auto mi_flags = sp<NTK>(ps.mi.flags);
  QView qv; if (ps.cnt_in) qv.init(ps.q_in, ps.cnt_in, ps.shard_cap);
  unsigned first = 0, count = ps.cnt_in ? qv.total() : ps.cap;  // no counts: bounce 0 of a pass whose samples are all traced (entry i = slot i = path i)
  if (MODE != 1 && ps.range) { first = ps.range[0]; count = ps.range[1]; }
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned n_shaded = 0, n_unreached = 0, n_tail = 0;
  const DScene& gsc = *sc.self;  // what out-of-line functions get: the scene record in device memory, not a private copy of the kernel argument
#ifdef RT_ABLATE
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = clock64();
#endif
  // one vertex per lane: entry i of the launch's queue, its records at slot rslot. Must be reached by every thread of the workgroup (the queue appends at its
  // end are workgroup-wide).
  auto shade_vertex = [&](const bool lane_live, const unsigned i, const unsigned rslot) {
    RT_STAMP(7);  // loop overhead / previous iteration's tail
    bool cont = false, want_shadow = false, want_mis = false, mis_occlusion_only = false, tail = false;
    // what a continuing path takes to its slot in the next bounce's queue (stored after the append below has named the slot)
    f3 nr_o = mk3(0, 0, 0), nr_d = mk3(0, 0, 0); rgb3 beta = mkc(0, 0, 0); float eta_scale = 1.0f; unsigned st_out = 0u, pid = 0u; unsigned long long rng_out = 0ull;
    if (lane_live) {
      // the vertex's records: four 16-byte loads at consecutive slots of consecutive lanes, requested together
      const float4 d4 = in_d[rslot], h4 = sraw(ps.hit)[rslot];  // (hit: read twice in a launch - never with the non-temporal hint)
      float4 b4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f); uint4 s4 = make_uint4(pack_state(0, false, 1, 2), rslot, 0u, 0u);
      // (fresh: bounce 0 of a pass whose samples are all traced - k_raygen left the record out, PassState::fresh; compiled out of the forms whose frames keep both)
      constexpr int FRESH = (MODE == 1 && LDSREC == 1) ? RT_FRESH_RECORDS_LDS : RT_FRESH_RECORDS;  // (rt_render: fresh_planes - the frames of exactly this kernel)
      if (!((FRESH & 1) && RT_FRESH_BETA(ps))) b4 = in_beta[rslot];
      if (!((FRESH & 2) && RT_FRESH_ST(ps))) s4 = in_st[rslot];
      pid = s4.y;
      unsigned sl, pix; split_path_id(ps, pid, sl, pix); const unsigned s = ps.s0 + sl;
      f3 ray_d = mk3(d4.x, d4.y, d4.z);
      beta = mkc(b4.x, b4.y, b4.z); eta_scale = b4.w;
      const unsigned st = s4.x;
      int bounces = (int)(st & 0xffu); bool specular_bounce = (st >> 8) & 1u;
      PathSampler smp; smp.tb = tables_of(ps); smp.pix = pix; smp.s = s; smp.c1 = (int)((st >> 9) & 15u); smp.c2 = (int)((st >> 13) & 15u);
      int x, y; unsigned long long pixel_index; owned_pixel(fp, fp.chunk_first + pix, x, y, pixel_index);
      smp.rng.inc = ((pixel_index * (unsigned long long)ps.spp + s + (1ull << 32)) << 1u) | 1ull;
      smp.rng.state = (unsigned long long)s4.z | ((unsigned long long)s4.w << 32);
      if ((FRESH & 2) && RT_FRESH_ST(ps)) smp.rng.state = (smp.rng.inc + 0x853c49e6748fea9bULL) * 0x5851f42d4c957f2dULL + smp.rng.inc;  // Pcg32::set_sequence (rng.rs:46-52) of the sample's keyed stream, as k_raygen leaves it
      int prim = __float_as_int(h4.y);
      const bool found = prim >= 0;
      // the frame loop's hit record is (b2, prim, b0, b1): the three barycentrics of the accepted test
      SurfaceInteraction si; TriHit th; th.t = 0.0f; th.b0 = h4.z; th.b1 = h4.w; th.b2 = h4.x;
      if (found) {
        if (GENERAL && sc.n_instances != 0u && (unsigned)prim >= sc.n_top_prims) {  // a hit inside an object instance: from here on `prim` is the object's primitive
          const float4 o4 = in_o[rslot];
          prim = sc.obj_general ? instance_fill_interaction<true>(gsc, (unsigned)prim, o4.x, o4.y, o4.z, ray_d.x, ray_d.y, ray_d.z, th.b0, th.b1, th.b2, si)
                                : instance_fill_interaction<false>(gsc, (unsigned)prim, o4.x, o4.y, o4.z, ray_d.x, ray_d.y, ray_d.z, th.b0, th.b1, th.b2, si);
        }
        else if (GENERAL && (tri_flags(sc.tri_p, prim) & RT_FLAG_SPHERE)) {  // Sphere::intersect builds its interaction from the ray: origin and direction of the path's ray
          const float4 o4 = in_o[rslot];
          (void)sphere_fill_interaction(sc.spheres[prim_sphere_index(sc.tri_p, prim)], mk3(o4.x, o4.y, o4.z), ray_d, si);
          si.ssb = normalize(si.sh_dpdu);
          si.prim = prim;
        }
        else if (MODE != 0) tri_fill_interaction_inl<MODE == 1 || MODE == 3>(sc, prim, ray_d, th, si);
        else tri_fill_interaction(gsc, prim, ray_d, th, si);
      }
      RT_STAMP(0);  // path state loads + SurfaceInteraction
      // path.rs:127-136 emitted light at the vertex / from the environment
      if (bounces == 0 || specular_bounce) {
        if (found) {
          int li = rec_light(sc.tri_rec, prim);
          // The path's radiance so far is not this kernel's business (round 4): it only adds what the vertex emits towards the path, where there is something to
          // add (rare), as a read-modify-write of lacc[pid] right here. The reference's order of a path's terms is kept: this one reaches lacc before the bounce's
          // direct-light terms (any-hit epilogue, k_resolve), after the previous bounce's.
          if (li >= 0) { const float4 l4 = ps.lacc[pid]; const rgb3 L = mkc(l4.x, l4.y, l4.z) + beta * area_light_l(sc.lights[li], si.hit.n, -ray_d); ps.lacc[pid] = make_float4(L.r, L.g, L.b, l4.w); }
        } else if ((MODE != 1 && !LEAN) && sc.n_infinite > 0) {
          const float4 l4 = ps.lacc[pid]; rgb3 L = mkc(l4.x, l4.y, l4.z);
          for (int k = 0; k < sc.n_infinite; ++k) L = L + beta * infinite_le<BOUNCED>(sc, sc.lights[k == 0 ? sc.infinite_ids[0] : (k == 1 ? sc.infinite_ids[1] : (k == 2 ? sc.infinite_ids[2] : sc.infinite_ids[3]))], ray_d);  // constant indices: the kernel argument stays in SGPRs
          ps.lacc[pid] = make_float4(L.r, L.g, L.b, l4.w);
        }
      }
      if (found && bounces < fp.max_depth) {  // path.rs:139
        if ((MODE != 1 && !LEAN && !BOUNCED) && bounces == 0 && sc.needs_differentials && !RT_DBG(sc, 16)) {  // only the camera ray carries differentials (interaction.rs:245-314)
          f2 pf; { float2 t = pfilm_[pid]; pf = mk2(t.x, t.y); }
          const f2 pl = fp.lens_radius > 0.0f ? table_2d(smp.tb, pix, 1, s) : mk2(0.0f, 0.0f);
          CameraRay cr = generate_camera_ray(fp, pf, pl, 1.0f / sqrtf((float)ps.spp));
          if (MODE == 0) compute_differential_call(si, cr.rx_o, cr.ry_o, cr.rx_d, cr.ry_d); else compute_differential(si, cr.rx_o, cr.ry_o, cr.rx_d, cr.ry_d);
        }
        typename std::conditional<MODE == 1, SingleLambert, typename std::conditional<MODE == 3, SingleLambertT<!LEAN, BOUNCED>,
                                  typename std::conditional<MODE == 5, SmallBsdfT<false, LEAN>, typename std::conditional<MODE == 6, SmallBsdfT<true, LEAN>, GenericBsdf>::type>::type>::type>::type bsdf;
        RT_STAMP(1);  // emission + differentials
        if (MODE == 0) bsdf.build(gsc, rec_material(sc.tri_rec, prim), si); else bsdf.build(sc, rec_material(sc.tri_rec, prim), si);
        RT_STAMP(2);  // material: textures + lobes
        // light_distribution.lookup(p) (path.rs:154-158)
        const float* ld_func; const float* ld_cdf; float ld_int; long ld_row = 0;
        float4 ld_r0 = make_float4(0, 0, 0, 0), ld_r1 = ld_r0;  // (DScene::ld_rows8: the voxel's whole distribution, scenes of <= 3 lights)
        const bool rows8 = sc.ld_rows8 != nullptr;
        if (sc.ld_uniform) { ld_func = sc.ld_func; ld_cdf = sc.ld_cdf; if (rows8) { ld_r0 = sc.ld_rows8[0]; ld_r1 = sc.ld_rows8[1]; ld_int = ld_r0.x; } else ld_int = sc.ld_int[0]; }
        else if (sc.ld_dense8 != nullptr) {  // (<= 3 lights, a grid of moderate size: the voxel's record directly, two loads in flight together and no slot before them)
          const long v = voxel_of(sc, si.hit.p);
          ld_r0 = sc.ld_dense8[2 * v]; ld_r1 = sc.ld_dense8[2 * v + 1]; ld_int = ld_r0.x; ld_func = sc.ld_func; ld_cdf = sc.ld_cdf;
        }
        else {
          const long slot = sc.ld_slot[voxel_of(sc, si.hit.p)];
          if (slot >= 0) {
            ld_func = sc.ld_func + slot * sc.n_lights; ld_cdf = sc.ld_cdf + slot * (sc.n_lights + 1); ld_row = slot;
            if (rows8) { ld_r0 = sc.ld_rows8[2 * slot]; ld_r1 = sc.ld_rows8[2 * slot + 1]; ld_int = ld_r0.x; } else ld_int = sc.ld_int[slot];
          }
          else { ld_func = sc.ld_func; ld_cdf = sc.ld_cdf; ld_int = -1.0f; }
        }
        const unsigned nonspec = BSDF_ALL & ~BSDF_SPECULAR;
        // voxels are built eagerly for every cell a surface point can fall into (k_lightdist_mark); the rest carry -1.
        // Looking one up would mean the marking missed a cell: count it (rt_render then fails the frame) and skip.
        const bool voxel_ok = !(ld_int < 0.0f);
        if (!voxel_ok) atomicAdd(&ps.stats[ST_UNBUILT_VOXEL], 1ull);
        if (voxel_ok && bsdf.num_nonspecular() > 0 && sc.n_lights > 0) {  // uniform_sample_one_light, integrator/mod.rs:186-220
          float su = smp.get_1d();
          int light_num; float light_pdf;
          if (RT_DBG(sc, 128)) { light_num = clampi((int)(su * (float)sc.n_lights), 0, sc.n_lights - 1); light_pdf = 1.0f / (float)sc.n_lights; }  // (measurement builds: no row search)
          else if (rows8) d1_sample_discrete_row8(ld_r0, ld_r1, sc.n_lights, su, light_num, light_pdf);
          else if (MODE != 1 && sc.ld_glog >= 0) d1_sample_discrete_guided(ld_func, ld_cdf, ld_int, sc.n_lights, su, sc.ld_guide + ld_row * ((1 << sc.ld_glog) + 1), sc.ld_glog, light_num, light_pdf);
          else d1_sample_discrete(ld_func, ld_cdf, ld_int, sc.n_lights, su, light_num, light_pdf);
          RT_STAMP(3);  // light pick: voxel row + discrete search
          if (light_pdf != 0.0f) {
            f2 u_light = smp.get_2d();
            f2 u_scattering = smp.get_2d();
            const DLight& light = sc.lights[RT_DBG(sc, 256) ? 0 : light_num];  // (measurement builds, 256: one light's record for every lane - no gather)
            // ---- estimate_direct (integrator/mod.rs:222-318), light-sampling half
            rgb3 ld1 = mkc(0, 0, 0); f3 sh_dir = mk3(0, 0, 0);
            const bool q_light = QLIGHTS && (tri_flags(sc.tri_p, light.prim) & RT_FLAG_SPHERE) != 0u;  // the picked light sits on a sphere
            f3 q_center = mk3(0, 0, 0);
            LiSample ls;
            if (q_light) {  // DiffuseAreaLight::sample_li (diffuse.rs:59-70) over the cone branch of Sphere::sample_si, as light_sample_li_inl<true> assembles it
              const DSphere& sp = sc.spheres[prim_sphere_index(sc.tri_p, light.prim)];
              q_center = xf34_point(sp.o2w, mk3(0, 0, 0));
              float pdf; const SpherePoint pt = sphere_cone_sample_si(sp, q_center, si.hit, u_light, pdf);
              ls.p1.p = pt.p; ls.p1.p_error = pt.p_error; ls.p1.n = pt.n;
              ls.wi = normalize(pt.p - si.hit.p); ls.pdf = pdf; ls.li = area_light_l(light, pt.n, -ls.wi);
            } else ls = (MODE == 1 || LEAN) ? area_light_sample_li(sc, light, si.hit, u_light) : light_sample_li_full<GENERAL, false, BOUNCED>(gsc, light, si.hit, u_light);
            if (ls.pdf > 0.0f && !is_black(ls.li)) {
              rgb3 f = bsdf.f(si.hit.wo, ls.wi, nonspec) * fabsf(dot(ls.wi, si.sh_n));
              float scattering_pdf = ((MODE != 1 && !LEAN) && light_is_delta(light)) ? 0.0f : bsdf.pdf(si.hit.wo, ls.wi, nonspec);  // read by the power heuristic only: a delta light has none
              if (!is_black(f)) {
                Ray sr = spawn_ray_to_interaction(si.hit, ls.p1);  // VisibilityTester, light/mod.rs:52-55
                sh_o[i] = make_float4(sr.o.x, sr.o.y, sr.o.z, sr.t_max);  // (shadow and MIS records sit at the vertex's position in THIS launch's queue, see PassState::sh)
                sh_dir = sr.d;
                want_shadow = true;
                if (light_is_delta(light)) ld1 = vdiv(f * ls.li, ls.pdf);
                else ld1 = vdiv(f * ls.li * power_heuristic1(ls.pdf, scattering_pdf), ls.pdf);
              }
            }
            RT_STAMP(4);  // light-sampling half: sample_li, f, pdf, shadow ray
            // ---- BSDF-sampling half
            rgb3 f2v = mkc(0, 0, 0); float w2 = 0.0f, spdf2 = 1.0f;
            if (!light_is_delta(light) && !RT_DBG(sc, 8)) {
              LobeSample bs = bsdf.sample_f(si.hit.wo, u_scattering, nonspec);
              rgb3 f = bs.f * fabsf(dot(bs.wi, si.sh_n));
              if (!is_black(f) && bs.pdf > 0.0f) {
                float weight = 1.0f; bool go = true;
                if (!(bs.type & BSDF_SPECULAR)) {
                  float lp;
                  if (q_light) lp = sphere_cone_pdf_wi(sc.spheres[prim_sphere_index(sc.tri_p, light.prim)], q_center, si.hit);
                  else if (RT_DBG(sc, 512)) lp = 1.0f;  // (measurement builds: no re-intersection of the emitter)
                  else lp = (MODE == 1 || LEAN) ? area_light_pdf_li<false>(sc, light, si.hit, bs.wi) : light_pdf_li<GENERAL, BOUNCED>(gsc, light, si.hit, bs.wi);
                  if (lp == 0.0f) go = false;  // `return ld`
                  else weight = power_heuristic1(bs.pdf, lp);
                  if ((GENERAL || QLIGHTS) && go && ps.skip_unreachable_mis && light.kind == 0 && (tri_flags(sc.tri_p, light.prim) & RT_FLAG_SPHERE)) {
                    const float4 b0 = sc.tri_p[3 * (size_t)light.prim], b1 = sc.tri_p[3 * (size_t)light.prim + 1];  // a quadric's leaf record: its world box
                    if (!ray_may_reach_box(mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z), si.hit.p, bs.wi)) { go = false; n_unreached += 1u; }
                  }
                }
                if (go) {
                  Ray mr = spawn_ray(si.hit, bs.wi);
                  mi_o[i] = make_float4(mr.o.x, mr.o.y, mr.o.z, kInf);
                  mi_d[i] = make_float4(mr.d.x, mr.d.y, mr.d.z, __uint_as_float(pid));  // the path the record belongs to
                  want_mis = true; f2v = f; w2 = weight; spdf2 = bs.pdf;
                  // An infinite light is never the emitter a ray hits (integrator/mod.rs:291-309): the term is `Le(ray)` if the ray leaves the
                  // scene and nothing otherwise, so occlusion is all this ray has to report.
                  mis_occlusion_only = (MODE != 1 && !LEAN) && ps.mis_any && light.kind == 3;
                }
              }
            }
            if (want_mis) {  // both halves are combined by k_resolve once both rays are back
              mi_a[i] = make_float4(ld1.r, ld1.g, ld1.b, light_pdf);
              mi_b[i] = make_float4(f2v.r, f2v.g, f2v.b, w2);
              mi_c[i] = make_float4(beta.r, beta.g, beta.b, spdf2);
              mi_flags[i] = (want_shadow ? RT_PEND_SHADOW : 0u) | 2u | ((unsigned)light_num << 2) | (mis_occlusion_only ? RT_PEND_MIS_ANY : 0u);
              if (!want_shadow) ps.occ_sh[i] = (unsigned char)1;  // no light-sampling term: as good as blocked (the any-hit kernel writes the byte of every other vertex)
            } else if (want_shadow) {  // L += beta * ((0 + Ld1) / pick_pdf) if unoccluded, applied by the any-hit kernel
              rgb3 add = beta * vdiv(mkc(0, 0, 0) + ld1, light_pdf);
              sh_add[i] = make_float4(add.r, add.g, add.b, 0.0f);
            }
            if (want_shadow) sh_d[i] = make_float4(sh_dir.x, sh_dir.y, sh_dir.z, __uint_as_float((want_mis ? 0u : 0x80000000u) | pid));  // bit 31: complete here (no MIS ray), bits 0-30: the path
          }
        }
        RT_STAMP(5);  // BSDF-sampling half + records
        // ---- sample the BSDF for the next direction (path.rs:172-196)
        f3 wo = -ray_d;  // not normalised (reference quirk)
        LobeSample bs = bsdf.sample_f(wo, smp.get_2d(), BSDF_ALL);
        if (!(is_black(bs.f) || bs.pdf <= 0.0f)) {
          beta = vdiv(beta * bs.f * fabsf(dot(bs.wi, si.sh_n)), bs.pdf);
          specular_bounce = (bs.type & BSDF_SPECULAR) != 0u;
          if ((bs.type & BSDF_SPECULAR) && (bs.type & BSDF_TRANSMISSION)) {
            float eta = bsdf.eta();
            eta_scale *= dot(wo, si.hit.n) > 0.0f ? eta * eta : vdiv(1.0f, eta * eta);
          }
          cont = true;
          rgb3 rr_beta = beta * eta_scale;  // path.rs:201-209
          if (max_component_value(rr_beta) < fp.rr_threshold && bounces > 3) {
            float q = fmaxf(1.0f - max_component_value(rr_beta), 0.05f);
            if (smp.get_1d() < q) cont = false;
            else beta = vdiv(beta, 1.0f - q);
          }
          if (cont) bounces += 1;
          // the next iteration would trace this ray, add what it reaches only after a specular bounce, and leave at bounces >= max_depth (path.rs:127-139)
          if (cont && ps.skip_dead_tail && bounces >= fp.max_depth && !specular_bounce) { cont = false; tail = true; }
          if (cont) { const Ray nr = spawn_ray(si.hit, bs.wi); nr_o = nr.o; nr_d = nr.d; }
        }
      }
      st_out = pack_state(bounces, specular_bounce, smp.c1, smp.c2); rng_out = smp.rng.state;
    }
    RT_STAMP(6);  // continuation sample, spawn, state stores
    // (a wave-uniform count, kept in a scalar register: a per-lane counter is one more live vector register in every form, and an atomic where the paths end
    // is ~7 M atomics on one word in the launch at the depth limit - k_shade<1> 271 -> 329 ms per S1 frame, measured)
    n_tail += (unsigned)__popcll(__ballot(tail));
    constexpr int NQ = (MODE == 1 || LEAN) ? 3 : 4;  // area lights only: every MIS ray needs its closest hit
    const int ci[4] = {0, 1, 2, 3}; const bool pr[4] = {cont, want_shadow, want_mis && !mis_occlusion_only, want_mis && mis_occlusion_only}; unsigned slot[4];
    block_push<NQ>(ps.cnt_out, ps.shard_cap, ci, pr, slot);
    if (cont) {  // the path's records for the next bounce, at its slot of that bounce's queue: a wave's stores are runs of consecutive slots
      out_o[slot[0]] = make_float4(nr_o.x, nr_o.y, nr_o.z, kInf);
      out_d[slot[0]] = make_float4(nr_d.x, nr_d.y, nr_d.z, 0.0f);
      out_beta[slot[0]] = make_float4(beta.r, beta.g, beta.b, eta_scale);
      out_st[slot[0]] = make_uint4(st_out, pid, (unsigned)rng_out, (unsigned)(rng_out >> 32));
    }
    if (want_shadow) ps.q_shadow[slot[1]] = i;  // the ray queues name RECORDS (= this launch's queue positions)
    if (pr[2]) ps.q_mis[slot[2]] = i;
    if (NQ == 4 && pr[3]) ps.q_misany[slot[3]] = i;
  };
  // slot = entry i of the sharded queue by the shard counts alone; on a material-sorted queue the sorted list names the slot
  if (MODE != 1) {
    for (unsigned base = first + blockIdx.x * blockDim.x; base < count; base += stride) {
      const unsigned i = base + threadIdx.x;
      const bool lane_live = i < count;
      n_shaded += lane_live ? 1u : 0u;
      shade_vertex(lane_live, i, lane_live ? (ps.cnt_in ? qv.get(i) : i) : 0u);
    }
  } else {
    // MODE 1 (area lights only: a ray that left the scene adds nothing and ends its path): the workgroup COMPACTS its entries before it shades them. A fifth of
    // S1's vertices are such misses (the box is open towards the camera) and their lanes sat through the ~4000 instructions of the others: 44 of 64 lanes per
    // VALU instruction. Each iteration the 256 threads look at the hit records of 256 entries, append the (entry, slot) pairs of the hits to a ring in LDS, and
    // whenever the ring holds 256 of them a full workgroup of vertices is shaded; the remainder at the end. Which lane shades a vertex is irrelevant (paths are
    // independent; a vertex's shadow / MIS records sit at its own entry number whoever writes them): same film, same counters.
    __shared__ unsigned s_ring_slot[512], s_ring_i[512], s_wave_hits[4];
    unsigned head = 0, n_pend = 0;  // (workgroup-uniform)
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (unsigned base = first + blockIdx.x * blockDim.x; base < count; base += stride) {
      const unsigned i = base + threadIdx.x;
      const bool lane_live = i < count;
      n_shaded += lane_live ? 1u : 0u;
      const unsigned rslot = lane_live ? (ps.cnt_in ? qv.get(i) : i) : 0u;
      const int hprim = lane_live ? __float_as_int(sraw(ps.hit)[rslot].y) : -1;
      const bool hit = hprim >= 0;
      const unsigned long long m = __ballot(hit);
      if (lane == 0u) s_wave_hits[wv] = (unsigned)__popcll(m);
      __syncthreads();
      unsigned before = 0, total = 0;
#pragma unroll
      for (unsigned w = 0; w < 4u; ++w) { const unsigned c = w < (blockDim.x >> 6) ? s_wave_hits[w] : 0u; before += w < wv ? c : 0u; total += c; }
      if (hit) { const unsigned pos = (head + n_pend + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) & 511u; s_ring_slot[pos] = rslot; s_ring_i[pos] = i; }
      n_pend += total;
      __syncthreads();
      if (n_pend >= blockDim.x) {
        const unsigned pos = (head + threadIdx.x) & 511u;
        shade_vertex(true, s_ring_i[pos], s_ring_slot[pos]);
        head = (head + blockDim.x) & 511u; n_pend -= blockDim.x;
      }
    }
    if (n_pend > 0u) {
      const unsigned pos = (head + threadIdx.x) & 511u;
      const bool lv = threadIdx.x < n_pend;
      shade_vertex(lv, lv ? s_ring_i[pos] : 0u, lv ? s_ring_slot[pos] : 0u);
    }
  }
  if (GENERAL || QLIGHTS) {
    for (int off = 32; off > 0; off >>= 1) n_unreached += __shfl_down(n_unreached, off);
    if ((threadIdx.x & 63u) == 0u && n_unreached) atomicAdd(&ps.stats[ST_MIS_UNREACHED], (unsigned long long)n_unreached);
  }
  if ((threadIdx.x & 63u) == 0u && n_tail) atomicAdd(&ps.stats[ST_TAIL_UNCAST], (unsigned long long)n_tail);  // (the wave's count, the same in every lane)
  for (int off = 32; off > 0; off >>= 1) n_shaded += __shfl_down(n_shaded, off);
  if ((threadIdx.x & 63u) == 0u && n_shaded) atomicAdd(&ps.stats[ST_SHADED + (MODE == 1 ? 0 : (MODE == 3 ? 1 : (MODE == 5 || MODE == 6 ? 2 : 3)))], (unsigned long long)n_shaded);
#ifdef RT_ABLATE
  if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 8; ++k) atomicAdd(&ps.stats[ST_STAMP + 8 * (MODE == 1 ? 0 : (MODE == 3 ? 1 : (MODE == 5 || MODE == 6 ? 2 : 3))) + k], stamp_acc[k]);
#endif
}


}  // namespace rtx
