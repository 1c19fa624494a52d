// rtx_dev_bsdf.h — device BxDF lobes and the Bsdf aggregate.
// Restates rc/bsdf/{mod,bxdf,lambertian,oren_nayar,fresnel,microfacet}.rs and rc/sampling/mod.rs,
// keeping the reference's deviations from pbrt-v3 so that the
// estimator is the same one the CPU integrator evaluates.
#pragma once
#include "rtx_dev_math.h"

namespace rtx {

// ---------------------------------------------------------------- rc/sampling/mod.rs
// sin and cos of concentric_sample_disk's angle (round 5). theta is (pi/4) q or pi/2 - (pi/4) q with |q| <= 1: never outside [-pi/4, 3 pi/4]. The library's sincosf is
// written for every float - 243 instructions a call site with its large-argument reduction, ~70 of them executed - and is called twice per vertex. Here: in the
// second case the angle goes back to [-pi/4, pi/4] as t = (theta - fl(pi/2)) - (pi/2 - fl(pi/2)) (the first difference is exact, Sterbenz), sin theta = cos t,
// cos theta = -sin t; on [-pi/4, pi/4] the Cephes single-precision minimax polynomials, Horner with fused steps. Measured against the correctly rounded values over
// 4 M arguments per case: at most 1.47 ulp, the last bit differs on a quarter of them - the class of difference the device library's sinf / cosf already had against
// glibc's (the film gate absorbs it; ocml documents 2 ulp). `second`: the angle is the pi/2 - (pi/4) q form.
#ifndef RT_DISK_SINCOS
#define RT_DISK_SINCOS 1
#endif
RT_DEV void disk_sincos(float theta, bool second, float& sn, float& cs) {
  const float x = second ? ((theta - 1.57079637050628662109375f) - (-4.37113882867379e-8f)) : theta;
  const float s = x * x;
  float p = __builtin_fmaf(-1.9515295891e-4f, s, 8.3321608736e-3f); p = __builtin_fmaf(p, s, -1.6666654611e-1f);
  const float sx = __builtin_fmaf(p * s, x, x);
  float q = __builtin_fmaf(2.443315711809948e-5f, s, -1.388731625493765e-3f); q = __builtin_fmaf(q, s, 4.166664568298827e-2f);
  const float cx = __builtin_fmaf(q * s, s, __builtin_fmaf(-0.5f, s, 1.0f));
  sn = second ? cx : sx; cs = second ? -sx : cx;
}
RT_DEV f2 concentric_sample_disk(f2 u) {  // :28-47
  float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
  if (ox == 0.0f && oy == 0.0f) return mk2(0.0f, 0.0f);
  float r, theta; const bool second = !(fabsf(ox) > fabsf(oy));
  if (!second) { r = ox; theta = kPiOver4 * (oy / ox); }
  else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
  float sn, cs;
  if (RT_DISK_SINCOS) disk_sincos(theta, second, sn, cs);
  else sincosf(theta, &sn, &cs);  // one argument reduction and one pair of polynomials instead of two (k_shade<1>: 4537 -> 4294 instructions)
  return mk2(r * cs, r * sn);
}
RT_DEV f3 cosine_sample_hemisphere(f2 u) {  // :22-26
  f2 d = concentric_sample_disk(u);
  float z = sqrtf(fmaxf(1.0f - d.x * d.x - d.y * d.y, 0.0f));
  return mk3(d.x, d.y, z);
}
RT_DEV f2 uniform_sample_triangle(f2 u) { float su0 = sqrtf(u.x); return mk2(1.0f - su0, u.y * su0); }  // :49-52
RT_DEV float power_heuristic1(float f_pdf, float g_pdf) {  // :59-63 with nf = ng = 1
  float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
  return vdiv(f * f, f * f + g * g);
}

enum : unsigned { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };

// ---------------------------------------------------------------- Fresnel (rc/bsdf/fresnel.rs)
RT_DEV f3 reflect(f3 wo, f3 n) { return -wo + n * 2.0f * dot(wo, n); }  // :14-16
RT_DEV bool refract(f3 i, f3 n, float eta, f3& wt) {                   // :19-30
  float cos_theta_i = dot(n, i);
  float sin2theta_i = fmaxf(1.0f - cos_theta_i * cos_theta_i, 0.0f);
  float sin2theta_t = eta * eta * sin2theta_i;
  if (sin2theta_t >= 1.0f) return false;
  float cos_theta_t = sqrtf(1.0f - sin2theta_t);
  wt = eta * -i + (eta * cos_theta_i - cos_theta_t) * n;
  return true;
}
RT_DEV float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {  // :33-58
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  if (cos_theta_i <= 0.0f) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_theta_i = fabsf(cos_theta_i); }
  float sin_theta_i = sqrtf(fmaxf(1.0f - cos_theta_i * cos_theta_i, 0.0f));
  float sin_theta_t = eta_i / eta_t * sin_theta_i;
  if (sin_theta_t >= 1.0f) return 1.0f;
  float cos_theta_t = sqrtf(fmaxf(1.0f - sin_theta_t * sin_theta_t, 0.0f));
  float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}
// the same for callers that only scale radiance by it (fresnel_eval); FresnelSpecular::sample_f, which picks reflection or transmission by it, keeps the one above
RT_DEV float fr_dielectric_v(float cos_theta_i, float eta_i, float eta_t) {
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  if (cos_theta_i <= 0.0f) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_theta_i = fabsf(cos_theta_i); }
  float sin_theta_i = sqrtf(fmaxf(1.0f - cos_theta_i * cos_theta_i, 0.0f));
  float sin_theta_t = vdiv(eta_i, eta_t) * sin_theta_i;
  if (sin_theta_t >= 1.0f) return 1.0f;
  float cos_theta_t = sqrtf(fmaxf(1.0f - sin_theta_t * sin_theta_t, 0.0f));
  float r_parl = vdiv((eta_t * cos_theta_i) - (eta_i * cos_theta_t), (eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  float r_perp = vdiv((eta_i * cos_theta_i) - (eta_t * cos_theta_t), (eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}
RT_DEVN rgb3 fr_conductor(float cos_theta_i, rgb3 eta_i, rgb3 eta_t, rgb3 k) {  // :60-82
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  rgb3 eta = vdiv(eta_t, eta_i), eta_k = vdiv(k, eta_i);
  float cos2 = cos_theta_i * cos_theta_i, sin2 = 1.0f - cos2;
  rgb3 eta2 = eta * eta, eta_k2 = eta_k * eta_k;
  rgb3 t0 = eta2 - eta_k2 - sin2;
  rgb3 a2plusb2 = sqrt3(t0 * t0 + 4.0f * eta2 * eta_k2);
  rgb3 t1 = a2plusb2 + cos2;
  rgb3 a = sqrt3(0.5f * (a2plusb2 + t0));
  rgb3 t2 = 2.0f * cos_theta_i * a;
  rgb3 r_s = vdiv(t1 - t2, t1 + t2);
  rgb3 t3 = cos2 * a2plusb2 + sin2 * sin2;
  rgb3 t4 = t2 * sin2;
  rgb3 r_p = vdiv(r_s * (t3 - t4), t3 + t4);
  return 0.5f * (r_p + r_s);
}

// ---------------------------------------------------------------- lobes
enum { LB_LAMBERT_R = 0, LB_LAMBERT_T, LB_OREN_NAYAR, LB_SPEC_R, LB_SPEC_T, LB_FRESNEL_SPEC, LB_FRESNEL_BLEND, LB_MICRO_R, LB_MICRO_T,
       LB_DISNEY_DIFFUSE, LB_DISNEY_FAKESS, LB_DISNEY_RETRO, LB_DISNEY_SHEEN, LB_DISNEY_CLEARCOAT };  // rc/material/disney.rs:215-418
// fr_kind: bits 0-1 = Fresnel kind; bit 2 = the microfacet distribution is Disney's (separable masking-shadowing, disney.rs:444-476)
enum { FR_NOOP = 0, FR_DIELECTRIC = 1, FR_CONDUCTOR = 2, FR_DISNEY = 3, FR_SEPARABLE_G = 4 };
struct Lobe {
  int kind, fr_kind;
  rgb3 r;        // R / T / FresnelSpecular R / FresnelBlend Rd
  rgb3 t;        // FresnelSpecular T / FresnelBlend Rs / conductor eta_t / DisneyFresnel R0
  rgb3 k;        // conductor k
  float ax, ay;  // Trowbridge-Reitz alphas, or Oren-Nayar A, B; Disney FakeSS / Retro: ax = roughness; ClearCoat: ax = weight, ay = gloss
  float eta_a, eta_b;  // transmission lobes / FresnelSpecular; dielectric Fresnel uses (fr_ei, fr_et)
  float fr_ei, fr_et;  // DisneyFresnel: metallic, eta
  int n_scales; rgb3 scale0, scale1;  // ScaledBxDF nesting (bxdf.rs:48-71), innermost first
};
RT_DEV unsigned lobe_type(int kind) {
  switch (kind) {
    case LB_LAMBERT_R: return BSDF_DIFFUSE | BSDF_REFLECTION;
    case LB_LAMBERT_T: return BSDF_DIFFUSE | BSDF_TRANSMISSION;
    case LB_OREN_NAYAR: return BSDF_REFLECTION | BSDF_DIFFUSE;
    case LB_SPEC_R: return BSDF_SPECULAR | BSDF_REFLECTION;
    case LB_SPEC_T: return BSDF_SPECULAR | BSDF_TRANSMISSION;
    case LB_FRESNEL_SPEC: return BSDF_SPECULAR | BSDF_REFLECTION | BSDF_TRANSMISSION;
    case LB_FRESNEL_BLEND: return BSDF_REFLECTION | BSDF_GLOSSY;
    case LB_MICRO_R: return BSDF_REFLECTION | BSDF_GLOSSY;
    case LB_DISNEY_DIFFUSE: case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN: return BSDF_REFLECTION | BSDF_DIFFUSE;
    case LB_DISNEY_CLEARCOAT: return BSDF_REFLECTION | BSDF_GLOSSY;
    default: return BSDF_TRANSMISSION | BSDF_GLOSSY;
  }
}
RT_DEV bool lobe_matches(int kind, unsigned flags) { unsigned t = lobe_type(kind); return (t & flags) == t; }  // bxdf.rs:29-31
// Schlick helpers of rc/material/disney.rs:478-503
RT_DEV float schlick_weight(float cos_theta) { float m = clampf(1.0f - cos_theta, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
RT_DEV float fr_schlick(float r0, float cos_theta) { return lerpf(schlick_weight(cos_theta), r0, 1.0f); }
RT_DEV rgb3 fresnel_eval(const Lobe& l, float cos_theta_i) {  // fresnel.rs:110-136 (abs() first: quirk 7)
  const int k = l.fr_kind & 3;
  if (k == FR_DIELECTRIC) return grey(fr_dielectric_v(fabsf(cos_theta_i), l.fr_ei, l.fr_et));
  if (k == FR_CONDUCTOR) return fr_conductor(fabsf(cos_theta_i), mkc(1.0f, 1.0f, 1.0f), l.t, l.k);
  if (k == FR_DISNEY) {  // DisneyFresnel::evaluate (disney.rs:434-442): lerp(metallic, dielectric, schlick); no abs() here
    const float w = schlick_weight(cos_theta_i), m = l.fr_ei;
    rgb3 a = grey(fr_dielectric_v(cos_theta_i, 1.0f, l.fr_et)), b = l.t * (1.0f - w) + mkc(1.0f, 1.0f, 1.0f) * w;
    return a * (1.0f - m) + b * m;
  }
  return mkc(1.0f, 1.0f, 1.0f);
}

// TrowbridgeReitzDistribution, microfacet.rs:470-650
RT_DEV float tr_roughness_to_alpha(float roughness) {  // :485-493
  roughness = fmaxf(roughness, 1e-3f);
  float x = logf(roughness);
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
RT_DEV float tr_d(float ax, float ay, f3 wh) {  // :576-588
  float tan2theta = tan2_theta(wh);
  if (isinf(tan2theta)) return 0.0f;
  float cos4theta = cos2_theta(wh) * cos2_theta(wh);
  float e = (vdiv(cos2_phi(wh), ax * ax) + vdiv(sin2_phi(wh), ay * ay)) * tan2theta;
  return vdiv(1.0f, kPi * ax * ay * cos4theta * (1.0f + e) * (1.0f + e));
}
RT_DEV float tr_lambda(float ax, float ay, f3 w) {  // :590-602
  float abs_tan_theta = fabsf(tan_theta(w));
  if (isinf(abs_tan_theta)) return 0.0f;
  float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
  float a2t2 = (alpha * abs_tan_theta) * (alpha * abs_tan_theta);
  return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
RT_DEV float tr_g1(float ax, float ay, f3 w) { return vdiv(1.0f, 1.0f + tr_lambda(ax, ay, w)); }
RT_DEV float tr_g(float ax, float ay, f3 wi, f3 wo) { return vdiv(1.0f, 1.0f + tr_lambda(ax, ay, wi) + tr_lambda(ax, ay, wo)); }
RT_DEV float lobe_g(const Lobe& l, f3 wi, f3 wo) { return (l.fr_kind & FR_SEPARABLE_G) ? tr_g1(l.ax, l.ay, wi) * tr_g1(l.ax, l.ay, wo) : tr_g(l.ax, l.ay, wi, wo); }
RT_DEV float tr_pdf(float ax, float ay, f3 wo, f3 wh) { return vdiv(tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * fabsf(dot(wo, wh)), abs_cos_theta(wo)); }
RT_DEV void tr_sample11(float cos_theta_, float u1, float u2, float& sx, float& sy) {  // :517-572
  if (cos_theta_ > 0.9999f) {
    float r = sqrtf(u1 / (1.0f - u1));
    float phi = kTau * u2;
    float sn, cs; sincosf(phi, &sn, &cs); sx = r * cs; sy = r * sn;
    return;
  }
  float sin_theta_ = sqrtf(fmaxf(1.0f - cos_theta_ * cos_theta_, 0.0f));
  float tan_theta_ = sin_theta_ / cos_theta_;
  float a = 1.0f / tan_theta_;
  float G1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
  float A = 2.0f * u1 / G1 - 1.0f;
  float tmp = 1.0f / (A * A - 1.0f);
  if (tmp > 1e10f) tmp = 1e10f;
  float B = tan_theta_;
  float D = sqrtf(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
  float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
  float slope_x = (A < 0.0f || slope_x_2 > 1.0f / tan_theta_) ? slope_x_1 : slope_x_2;
  float S;
  if (u2 > 0.5f) { S = 1.0f; u2 = 2.0f * (u2 - 0.5f); } else { S = -1.0f; u2 = 2.0f * (0.5f - u2); }
  float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  sx = slope_x;
  sy = S * z * sqrtf(1.0f + slope_x * slope_x);
}
RT_DEVN f3 tr_sample_wh(float ax, float ay, f3 wo, f2 u) {  // :495-514, 604-645 (visible-area sampling always on)
  bool flip = wo.z < 0.0f;
  f3 w = flip ? -wo : wo;
  f3 ws = normalize(mk3(ax * w.x, ay * w.y, w.z));
  float slope_x, slope_y;
  tr_sample11(cos_theta(ws), u.x, u.y, slope_x, slope_y);
  float tmp = cos_phi(ws) * slope_x - sin_phi(ws) * slope_y;
  slope_y = sin_phi(ws) * slope_x + cos_phi(ws) * slope_y;
  slope_x = tmp;
  slope_x *= ax; slope_y *= ay;
  f3 wh = normalize(mk3(-slope_x, -slope_y, 1.0f));
  return flip ? -wh : wh;
}

RT_DEV float pow5(float v) { return (v * v) * (v * v) * v; }
RT_DEV float disney_gtr1(float cos_theta, float alpha) {  // disney.rs:505-511 (log10, as the reference has it)
  float alpha2 = alpha * alpha;
  return (alpha2 - 1.0f) / (kPi * log10f(alpha2) * (1.0f + (alpha2 - 1.0f) * cos_theta * cos_theta));
}
RT_DEV float disney_smith_g_ggx(float cos_theta, float alpha) {  // disney.rs:513-519
  float alpha2 = alpha * alpha, cos_theta2 = cos_theta * cos_theta;
  return 1.0f / (cos_theta + sqrtf(alpha2 + cos_theta2 - alpha2 * cos_theta2));
}
RT_DEV float default_pdf(f3 wo, f3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f; }  // bxdf.rs:38-44

// A microfacet lobe evaluates D at the half vector through sin^2 = 1 - cos^2 of wh.z: with alpha ~ 1e-3 the lobe lives where sin^2 ~ alpha^2 = 1e-6, and ONE ulp in the
// length of wh (2e-7 in cos^2) is a fifth of that - the reference's own value is this ill-conditioned, so parity needs its wh bit for bit: below this alpha the half vector is
// normalised by the correctly rounded sqrt and quotients instead of v_rsq_f32 (found by scripts/fuzz_shading.py: un-remapped roughness 0.001 put films 3e-3 ... 1.6e-2 from
// the oracle's; at alpha 0.02 the same ulp is 5e-4 of one sample's D, random in sign). Remapped roughnesses never come near (roughness_to_alpha(0.001) = 0.054).
#ifndef RT_SHARP_ALPHA
#define RT_SHARP_ALPHA 0.02f
#endif
RT_DEV bool sharp_lobe(const Lobe& l) {
#ifdef RT_STRICT_SHADE
  return true;
#else
  return fminf(l.ax, l.ay) < RT_SHARP_ALPHA;
#endif
}

RT_DEV rgb3 lobe_f_inner(const Lobe& l, f3 wo, f3 wi) {
  switch (l.kind) {
    case LB_LAMBERT_R: case LB_LAMBERT_T: return l.r * kInvPi;
    case LB_OREN_NAYAR: {  // oren_nayar.rs:31-53
      float sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
      float max_cos = 0.0f;
      if (sin_theta_i > 1e-4f && sin_theta_o > 1e-4f) {
        float d_cos = sin_phi(wi) * sin_phi(wo) + cos_phi(wi) * cos_phi(wo);
        max_cos = fmaxf(d_cos, 0.0f);
      }
      float sin_alpha, tan_beta;
      if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / abs_cos_theta(wi); }
      else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / abs_cos_theta(wo); }
      return l.r * kInvPi * (l.ax + l.ay * max_cos * sin_alpha * tan_beta);
    }
    case LB_SPEC_R: case LB_SPEC_T: case LB_FRESNEL_SPEC: return mkc(0, 0, 0);
    case LB_FRESNEL_BLEND: {  // fresnel.rs:357-374 (r = Rd, t = Rs)
      rgb3 diffuse = (28.0f / (23.0f * kPi)) * l.r * (mkc(1, 1, 1) - l.t) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) *
                     (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
      f3 wh = wi + wo;
      if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mkc(0, 0, 0);
      wh = normalize(wh);
      rgb3 schlick = l.t + pow5(1.0f - dot(wi, wh)) * (mkc(1, 1, 1) - l.t);
      rgb3 specular = tr_d(l.ax, l.ay, wh) / (4.0f * fabsf(dot(wi, wh)) * fmaxf(abs_cos_theta(wi), abs_cos_theta(wo))) * schlick;
      return diffuse + specular;
    }
    case LB_MICRO_R: {  // microfacet.rs:36-53
      float cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
      f3 wh = wi + wo;
      if (cos_theta_o == 0.0f || cos_theta_i == 0.0f) return mkc(0, 0, 0);
      if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mkc(0, 0, 0);
      wh = sharp_lobe(l) ? normalize(wh) : vnormalize(wh);
      rgb3 fr = fresnel_eval(l, dot(wi, wh));
      return vdiv(l.r * tr_d(l.ax, l.ay, wh) * lobe_g(l, wo, wi) * fr, 4.0f * cos_theta_i * cos_theta_o);
    }
    case LB_DISNEY_DIFFUSE: {  // disney.rs:228-236
      float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
      return l.r * kInvPi * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
    }
    case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN: case LB_DISNEY_CLEARCOAT: {
      f3 wh = wi + wo;
      if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mkc(0, 0, 0);
      wh = normalize(wh);
      float cos_theta_d = dot(wi, wh);
      if (l.kind == LB_DISNEY_SHEEN) return l.r * schlick_weight(cos_theta_d);  // :332-341
      if (l.kind == LB_DISNEY_CLEARCOAT) {  // :363-378
        float Dr = disney_gtr1(abs_cos_theta(wh), l.ay);
        float Fr = fr_schlick(0.04f, dot(wo, wh));
        float Gr = disney_smith_g_ggx(abs_cos_theta(wo), 0.25f) * disney_smith_g_ggx(abs_cos_theta(wi), 0.25f);
        return grey(l.ax * Gr * Fr * Dr / 4.0f);
      }
      float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
      if (l.kind == LB_DISNEY_FAKESS) {  // :258-274
        float fss90 = cos_theta_d * cos_theta_d * l.ax;
        float fss = lerpf(fo, 1.0f, fss90) * lerpf(fi, 1.0f, fss90);
        float ss = 1.25f * (fss * (1.0f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - 0.5f) + 0.5f);
        return l.r * kInvPi * ss;
      }
      float rr = 2.0f * l.ax * cos_theta_d * cos_theta_d;  // DisneyRetro, :296-308
      return l.r * kInvPi * rr * (fo + fi + fo * fi * (rr - 1.0f));
    }
    default: {  // LB_MICRO_T, microfacet.rs:127-172 (mode == RADIANCE)
      if (same_hemisphere(wo, wi)) return mkc(0, 0, 0);
      float cos_theta_o = cos_theta(wo), cos_theta_i = cos_theta(wi);
      if (cos_theta_o == 0.0f || cos_theta_i == 0.0f) return mkc(0, 0, 0);
      float eta = cos_theta_o > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
      f3 wh = normalize(wo + wi * eta);
      if (wh.z < 0.0f) wh = -wh;
      rgb3 fr = fresnel_eval(l, dot(wo, wh));
      float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
      float factor = 1.0f / eta;
      return (mkc(1, 1, 1) - fr) * l.r *
             fabsf(tr_d(l.ax, l.ay, wh) * lobe_g(l, wo, wi) * eta * eta * fabsf(dot(wi, wh)) * fabsf(dot(wo, wh)) * factor * factor /
                   (cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom));
    }
  }
}
RT_DEV float lobe_pdf_inner(const Lobe& l, f3 wo, f3 wi) {
  switch (l.kind) {
    case LB_SPEC_R: case LB_SPEC_T: case LB_FRESNEL_SPEC: return 0.0f;
    case LB_FRESNEL_BLEND: {  // fresnel.rs:376-384
      if (!same_hemisphere(wo, wi)) return 0.0f;
      f3 wh = normalize(wo + wi);
      float pdf_wh = tr_pdf(l.ax, l.ay, wo, wh);
      return 0.5f * (abs_cos_theta(wi) * kInvPi + pdf_wh / (4.0f * dot(wo, wh)));
    }
    case LB_MICRO_R: {  // microfacet.rs:86-93
      if (!same_hemisphere(wo, wi)) return 0.0f;
      f3 wh = sharp_lobe(l) ? normalize(wo + wi) : vnormalize(wo + wi);
      return vdiv(tr_pdf(l.ax, l.ay, wo, wh), 4.0f * dot(wo, wh));
    }
    case LB_MICRO_T: {  // microfacet.rs:210-226
      if (same_hemisphere(wo, wi)) return 0.0f;
      float eta = cos_theta(wo) > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
      f3 wh = normalize(wo + wi * eta);
      float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
      float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
      return tr_pdf(l.ax, l.ay, wo, wh) * dwh_dwi;
    }
    case LB_DISNEY_CLEARCOAT: {  // disney.rs:397-413
      if (!same_hemisphere(wo, wi)) return 0.0f;
      f3 wh = wo + wi;
      if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
      wh = normalize(wh);
      float Dr = disney_gtr1(abs_cos_theta(wh), l.ay);
      return Dr * abs_cos_theta(wh) / (4.0f * dot(wo, wh));
    }
    default: return default_pdf(wo, wi);  // Lambertian R and T (quirk 6), Oren-Nayar, Disney diffuse lobes
  }
}
struct LobeSample { rgb3 f; f3 wi; float pdf; unsigned type; };
RT_DEV LobeSample mk_ls(rgb3 f, f3 wi, float pdf, unsigned type) { LobeSample s; s.f = f; s.wi = wi; s.pdf = pdf; s.type = type; return s; }
// WANT_F = false: the value of a NON-specular lobe is left out (zero). Bsdf::sample_f discards it - for a non-specular sample f is recomputed as the sum over
// every matching lobe (bsdf/mod.rs:228-247) - and with it goes one full microfacet evaluation (D, G, Fresnel) per sample. Specular lobes always return theirs.
template <bool WANT_F = true>
RT_DEV LobeSample lobe_sample_inner(const Lobe& l, f3 wo, f2 u) {
  const unsigned ty = lobe_type(l.kind);
  switch (l.kind) {
    case LB_SPEC_R: {  // fresnel.rs:158-163
      f3 wi = mk3(-wo.x, -wo.y, wo.z);
      return mk_ls(fresnel_eval(l, cos_theta(wi)) * l.r / abs_cos_theta(wi), wi, 1.0f, ty);
    }
    case LB_SPEC_T: {  // fresnel.rs:202-230
      bool entering = cos_theta(wo) > 0.0f;
      float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
      f3 wi;
      if (refract(wo, face_forward(mk3(0, 0, 1), wo), eta_i / eta_t, wi)) {
        rgb3 ft = l.r * (mkc(1, 1, 1) - fresnel_eval(l, cos_theta(wi)));
        ft = ft * (eta_i * eta_i) / (eta_t * eta_t);
        return mk_ls(ft / abs_cos_theta(wi), wi, 1.0f, ty);
      }
      return mk_ls(mkc(1, 1, 1), mk3(0, 0, 0), 0.0f, 0u);
    }
    case LB_FRESNEL_SPEC: {  // fresnel.rs:275-324
      float fr = fr_dielectric(cos_theta(wo), l.eta_a, l.eta_b);
      if (u.x < fr) {
        f3 wi = mk3(-wo.x, -wo.y, wo.z);
        return mk_ls(fr * l.r / abs_cos_theta(wi), wi, fr, BSDF_SPECULAR | BSDF_REFLECTION);
      }
      bool entering = cos_theta(wo) > 0.0f;
      float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
      f3 wi;
      if (refract(wo, face_forward(mk3(0, 0, 1), wo), eta_i / eta_t, wi)) {
        rgb3 ft = l.t * (1.0f - fr);
        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
        return mk_ls(ft / abs_cos_theta(wi), wi, 1.0f - fr, BSDF_SPECULAR | BSDF_TRANSMISSION);
      }
      return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
    }
    case LB_FRESNEL_BLEND: {  // fresnel.rs:386-407
      f3 wi;
      if (u.x < 0.5f) {
        u.x = fminf(2.0f * u.x, kOneMinusEpsilon);
        wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
      } else {
        u.x = fminf(2.0f * (u.x - 0.5f), kOneMinusEpsilon);
        f3 wh = tr_sample_wh(l.ax, l.ay, wo, u);
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return mk_ls(mkc(0, 0, 0), wi, 0.0f, ty);
      }
      return mk_ls(WANT_F ? lobe_f_inner(l, wo, wi) : mkc(0, 0, 0), wi, lobe_pdf_inner(l, wo, wi), ty);
    }
    case LB_MICRO_R: {  // microfacet.rs:61-84 (no wo.wh < 0 rejection: quirk 8)
      if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
      f3 wh = tr_sample_wh(l.ax, l.ay, wo, u);
      f3 wi = reflect(wo, wh);
      if (!same_hemisphere(wo, wi)) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
      float pdf = tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh));
      return mk_ls(WANT_F ? lobe_f_inner(l, wo, wi) : mkc(0, 0, 0), wi, pdf, ty);
    }
    case LB_MICRO_T: {  // microfacet.rs:180-208
      if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
      f3 wh = tr_sample_wh(l.ax, l.ay, wo, u);
      float eta = cos_theta(wo) > 0.0f ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
      f3 wi;
      if (refract(wo, wh, eta, wi)) {
        float pdf = lobe_pdf_inner(l, wo, wi);
        return mk_ls(WANT_F ? lobe_f_inner(l, wo, wi) : mkc(0, 0, 0), wi, pdf, ty);
      }
      return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
    }
    case LB_DISNEY_CLEARCOAT: {  // disney.rs:380-395
      if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
      float alpha2 = l.ay * l.ay;
      float cos_theta_ = sqrtf(fmaxf(0.0f, (1.0f - powf(alpha2, 1.0f - u.x)) / (1.0f - alpha2)));
      float sin_theta_ = sqrtf(fmaxf(0.0f, 1.0f - cos_theta_ * cos_theta_));
      float phi = 2.0f * kPi * u.y;
      float sn_phi, cs_phi; sincosf(phi, &sn_phi, &cs_phi);
      f3 wh = mk3(sin_theta_ * cs_phi, sin_theta_ * sn_phi, cos_theta_);  // spherical_direction, geometry/mod.rs:112-114
      if (!same_hemisphere(wo, wh)) wh = -wh;
      f3 wi = reflect(wo, wh);
      if (!same_hemisphere(wo, wi)) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, ty);
      float pdf = lobe_pdf_inner(l, wo, wi);
      return mk_ls(WANT_F ? lobe_f_inner(l, wo, wi) : mkc(0, 0, 0), wi, pdf, ty);
    }
    default: {  // default BxDF::sample_f (bxdf.rs:18-25): cosine sampling, EMPTY sampled-type flags (quirk 5)
      f3 wi = cosine_sample_hemisphere(u);
      if (wo.z < 0.0f) wi.z *= -1.0f;
      float pdf = lobe_pdf_inner(l, wo, wi);
      return mk_ls(WANT_F ? lobe_f_inner(l, wo, wi) : mkc(0, 0, 0), wi, pdf, 0u);
    }
  }
}
RT_DEV rgb3 apply_scales(const Lobe& l, rgb3 v) {
  if (l.n_scales > 0) v = v * l.scale0;
  if (l.n_scales > 1) v = v * l.scale1;
  return v;
}
RT_DEV rgb3 lobe_f(const Lobe& l, f3 wo, f3 wi) { return apply_scales(l, lobe_f_inner(l, wo, wi)); }
RT_DEV float lobe_pdf(const Lobe& l, f3 wo, f3 wi) { return l.n_scales > 0 ? default_pdf(wo, wi) : lobe_pdf_inner(l, wo, wi); }  // quirk 10
RT_DEV LobeSample lobe_sample(const Lobe& l, f3 wo, f2 u) { LobeSample s = lobe_sample_inner<false>(l, wo, u); s.f = apply_scales(l, s.f); return s; }  // only Bsdf::sample_f samples lobes

// ---------------------------------------------------------------- Bsdf (bsdf/mod.rs:64-269)
#ifndef RT_MAX_LOBES
#define RT_MAX_LOBES 8
#endif
struct Bsdf {
  float eta;
  f3 ns, ng, ss, ts;
  int n;
  Lobe lobes[RT_MAX_LOBES];
};
RT_DEV void bsdf_init_frame(Bsdf& b, f3 shading_n, f3 geom_n, f3 shading_dpdu) {  // :77-91
  b.ss = normalize(shading_dpdu);
  b.ns = shading_n; b.ng = geom_n;
  b.ts = cross(shading_n, b.ss);
}
RT_DEV f3 world_to_local(const Bsdf& b, f3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
RT_DEV f3 local_to_world(const Bsdf& b, f3 v) {
  return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
RT_DEV int bsdf_num_components(const Bsdf& b, unsigned flags) { int c = 0; for (int i = 0; i < b.n; ++i) c += lobe_matches(b.lobes[i].kind, flags); return c; }
RT_DEVN rgb3 bsdf_f(const Bsdf& b, f3 wo_w, f3 wi_w, unsigned flags) {  // :94-111
  f3 wi = world_to_local(b, wi_w), wo = world_to_local(b, wo_w);
  if (wo.z == 0.0f) return mkc(0, 0, 0);
  bool refl = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
  rgb3 c = mkc(0, 0, 0);
  for (int i = 0; i < b.n; ++i) {
    unsigned ty = lobe_type(b.lobes[i].kind);
    if (((ty & flags) == ty) && ((refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION)))) c = c + lobe_f(b.lobes[i], wo, wi);
  }
  return c;
}
RT_DEVN float bsdf_pdf(const Bsdf& b, f3 wo_w, f3 wi_w, unsigned flags) {  // :113-136
  if (b.n == 0) return 0.0f;
  f3 wo = world_to_local(b, wo_w);
  if (wo.z == 0.0f) return 0.0f;
  f3 wi = world_to_local(b, wi_w);
  int matched = 0; float p = 0.0f;
  for (int i = 0; i < b.n; ++i) if (lobe_matches(b.lobes[i].kind, flags)) { ++matched; p += lobe_pdf(b.lobes[i], wo, wi); }
  return matched == 0 ? 0.0f : p / (float)matched;
}
RT_DEVN LobeSample bsdf_sample_f(const Bsdf& b, f3 wo_w, f2 u, unsigned flags) {  // :138-251
  int m = bsdf_num_components(b, flags);
  if (m == 0) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
  int comp_i = (int)f2u_sat(floorf(u.x * (float)m));
  if (comp_i > m - 1) comp_i = m - 1;
  int chosen = 0, cnt = comp_i;
  for (int i = 0; i < b.n; ++i)
    if (lobe_matches(b.lobes[i].kind, flags)) { if (cnt == 0) { chosen = i; break; } --cnt; }
  const Lobe& bx = b.lobes[chosen];
  const unsigned bty = lobe_type(bx.kind);
  f2 ur = mk2(fminf(u.x * (float)m - (float)comp_i, kOneMinusEpsilon), u.y);
  f3 wo = world_to_local(b, wo_w);
  if (wo.z == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, bty);
  LobeSample s = lobe_sample(bx, wo, ur);
  if (s.pdf == 0.0f) return mk_ls(mkc(0, 0, 0), mk3(0, 0, 0), 0.0f, 0u);
  f3 wi = s.wi;
  f3 wi_w = local_to_world(b, wi);
  float pdf = s.pdf;
  if (!(bty & BSDF_SPECULAR) && m > 1)
    for (int i = 0; i < b.n; ++i) if (i != chosen && lobe_matches(b.lobes[i].kind, flags)) pdf += lobe_pdf(b.lobes[i], wo, wi);
  if (m > 1) pdf /= (float)m;
  rgb3 f = s.f;
  if (!(bty & BSDF_SPECULAR)) {
    bool refl = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
    f = mkc(0, 0, 0);
    for (int i = 0; i < b.n; ++i) {
      unsigned ty = lobe_type(b.lobes[i].kind);
      if (((ty & flags) == ty) && ((refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION)))) f = f + lobe_f(b.lobes[i], wo, wi);
    }
  }
  return mk_ls(f, wi_w, pdf, s.type);
}

}  // namespace rtx
