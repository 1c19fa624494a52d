// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// BxDFs and Bsdf: rc/bsdf/{mod,bxdf,lambertian,oren_nayar,fresnel,microfacet}.rs and
// rc/sampling/mod.rs. All reference quirks listed in SURVEY.md §8(a) are kept on purpose.
#pragma once
#include "orc_math.h"
#include "orc_sampler.h"

namespace orc {

// ---------------------------------------------------------------- rc/sampling/mod.rs
inline P2 concentric_sample_disk(P2 u) {  // :28-47
  float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
  if (ox == 0.0f && oy == 0.0f) return P2{0.0f, 0.0f};
  float r, theta;
  if (fabsf(ox) > fabsf(oy)) { r = ox; theta = kPiOver4 * (oy / ox); }
  else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
  return P2{r * cosf(theta), r * sinf(theta)};
}
inline V3 cosine_sample_hemisphere(P2 u) {  // :22-26
  P2 d = concentric_sample_disk(u);
  float z = sqrtf(fmaxf(1.0f - d.x * d.x - d.y * d.y, 0.0f));
  return v3(d.x, d.y, z);
}
inline P2 uniform_sample_triangle(P2 u) {  // :49-52
  float su0 = sqrtf(u.x);
  return P2{1.0f - su0, u.y * su0};
}
inline float power_heuristic(uint32_t nf, float f_pdf, uint32_t ng, float g_pdf) {  // :59-63
  float f = (float)nf * f_pdf, g = (float)ng * g_pdf;
  return (f * f) / (f * f + g * g);
}

// ---------------------------------------------------------------- BxDF type flags (bsdf/mod.rs:24-32)
enum : uint32_t {
  BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16,
  BSDF_ALL = 31
};

// ---------------------------------------------------------------- Fresnel (bsdf/fresnel.rs)
inline V3 reflect(V3 wo, V3 n) { return -wo + n * 2.0f * dot(wo, n); }  // :14-16  ((n*2)*dot)
inline bool refract(V3 i, V3 n, float eta, V3* wt) {                   // :19-30
  float cos_theta_i = dot(n, i);
  float sin2theta_i = fmaxf(1.0f - cos_theta_i * cos_theta_i, 0.0f);
  float sin2theta_t = eta * eta * sin2theta_i;
  if (sin2theta_t >= 1.0f) return false;
  float cos_theta_t = sqrtf(1.0f - sin2theta_t);
  *wt = eta * -i + (eta * cos_theta_i - cos_theta_t) * n;
  return true;
}
inline float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {  // :33-58
  cos_theta_i = clamp_t(cos_theta_i, -1.0f, 1.0f);
  if (cos_theta_i <= 0.0f) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_theta_i = fabsf(cos_theta_i); }
  float sin_theta_i = sqrtf(fmaxf(1.0f - cos_theta_i * cos_theta_i, 0.0f));
  float sin_theta_t = eta_i / eta_t * sin_theta_i;
  if (sin_theta_t >= 1.0f) return 1.0f;
  float cos_theta_t = sqrtf(fmaxf(1.0f - sin_theta_t * sin_theta_t, 0.0f));
  float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}
inline RGB fr_conductor(float cos_theta_i, RGB eta_i, RGB eta_t, RGB k) {  // :60-82
  cos_theta_i = clamp_t(cos_theta_i, -1.0f, 1.0f);
  RGB eta = eta_t / eta_i, eta_k = k / eta_i;
  float cos2 = cos_theta_i * cos_theta_i, sin2 = 1.0f - cos2;
  RGB eta2 = eta * eta, eta_k2 = eta_k * eta_k;
  RGB t0 = eta2 - eta_k2 - sin2;
  RGB a2plusb2 = rgb_sqrt(t0 * t0 + 4.0f * eta2 * eta_k2);
  RGB t1 = a2plusb2 + cos2;
  RGB a = rgb_sqrt(0.5f * (a2plusb2 + t0));
  RGB t2 = 2.0f * cos_theta_i * a;
  RGB r_s = (t1 - t2) / (t1 + t2);
  RGB t3 = cos2 * a2plusb2 + sin2 * sin2;
  RGB t4 = t2 * sin2;
  RGB r_p = r_s * (t3 - t4) / (t3 + t4);
  return 0.5f * (r_p + r_s);
}
// Schlick helpers of rc/material/disney.rs:478-503
inline float schlick_weight(float cos_theta) { float m = clamp_t(1.0f - cos_theta, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
inline float fr_schlick(float r0, float cos_theta) { return lerp_f(schlick_weight(cos_theta), r0, 1.0f); }
inline RGB fr_schlick_rgb(RGB r0, float cos_theta) { float w = schlick_weight(cos_theta); return r0 * (1.0f - w) + rgb(1, 1, 1) * w; }
enum FresnelKind { FR_NOOP = 0, FR_DIELECTRIC = 1, FR_CONDUCTOR = 2, FR_DISNEY = 3 };
struct Fresnel {
  int kind = FR_NOOP;
  float eta_i = 1.0f, eta_t = 1.0f;  // dielectric
  RGB c_eta_i{1, 1, 1}, c_eta_t{1, 1, 1}, c_k{0, 0, 0};
  RGB d_r0{0, 0, 0}; float d_metallic = 0.0f, d_eta = 1.5f;  // DisneyFresnel (disney.rs:420-442)
  RGB evaluate(float cos_theta_i) const {  // :110-136 (quirk 7: abs() before fr_dielectric)
    if (kind == FR_DIELECTRIC) return grey(fr_dielectric(fabsf(cos_theta_i), eta_i, eta_t));
    if (kind == FR_CONDUCTOR) return fr_conductor(fabsf(cos_theta_i), c_eta_i, c_eta_t, c_k);
    if (kind == FR_DISNEY) {  // lerp(metallic, dielectric, schlick); no abs() here
      RGB a = grey(fr_dielectric(cos_theta_i, 1.0f, d_eta)), b = fr_schlick_rgb(d_r0, cos_theta_i);
      return a * (1.0f - d_metallic) + b * d_metallic;
    }
    return rgb(1, 1, 1);
  }
};

// ---------------------------------------------------------------- TrowbridgeReitz (bsdf/microfacet.rs:470-650)
struct TRDist {
  float ax = 0.1f, ay = 0.1f;
  bool separable_g = false;  // DisneyMicrofacetDistribution (disney.rs:444-476): g = g1(wi) * g1(wo)
  static float roughness_to_alpha(float roughness) {  // :485-493
    roughness = fmaxf(roughness, 1e-3f);
    float x = logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
  }
  float d(V3 wh) const {  // :576-588
    float tan2theta = tan2_theta(wh);
    if (std::isinf(tan2theta)) return 0.0f;
    float cos4theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * tan2theta;
    return 1.0f / (kPi * ax * ay * cos4theta * (1.0f + e) * (1.0f + e));
  }
  float lambda(V3 w) const {  // :590-602
    float abs_tan_theta = fabsf(tan_theta(w));
    if (std::isinf(abs_tan_theta)) return 0.0f;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float a2t2 = (alpha * abs_tan_theta) * (alpha * abs_tan_theta);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
  }
  float g1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }                        // :235-237
  float g(V3 wi, V3 wo) const { return separable_g ? g1(wi) * g1(wo) : 1.0f / (1.0f + lambda(wi) + lambda(wo)); }   // :239-241
  float pdf(V3 wo, V3 wh) const { return d(wh) * g1(wo) * fabsf(dot(wo, wh)) / abs_cos_theta(wo); }  // :243-249 (visible area)
  static void sample11(float cos_theta_, float u1, float u2, float* sx, float* sy) {  // :517-572
    if (cos_theta_ > 0.9999f) {
      float r = sqrtf(u1 / (1.0f - u1));
      float phi = kTau * u2;
      *sx = r * cosf(phi); *sy = r * sinf(phi);
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
    float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *sx = slope_x;
    *sy = S * z * sqrtf(1.0f + slope_x * slope_x);
  }
  V3 sample(V3 wi, float u1, float u2) const {  // :495-514
    V3 ws = normalize(v3(ax * wi.x, ay * wi.y, wi.z));
    float slope_x, slope_y;
    sample11(cos_theta(ws), u1, u2, &slope_x, &slope_y);
    float tmp = cos_phi(ws) * slope_x - sin_phi(ws) * slope_y;
    slope_y = sin_phi(ws) * slope_x + cos_phi(ws) * slope_y;
    slope_x = tmp;
    slope_x *= ax; slope_y *= ay;
    return normalize(v3(-slope_x, -slope_y, 1.0f));
  }
  V3 sample_wh(V3 wo, P2 u) const {  // :604-645, sample_visible_area == true always (:481)
    bool flip = wo.z < 0.0f;
    V3 w = flip ? -wo : wo;
    V3 wh = sample(w, u.x, u.y);
    return flip ? -wh : wh;
  }
};

// ---------------------------------------------------------------- BxDF tagged union
enum BxdfKind {
  BX_LAMBERT_R = 0, BX_LAMBERT_T, BX_OREN_NAYAR, BX_SPEC_R, BX_SPEC_T, BX_FRESNEL_SPEC,
  BX_FRESNEL_BLEND, BX_MICRO_R, BX_MICRO_T,
  BX_DISNEY_DIFFUSE, BX_DISNEY_FAKESS, BX_DISNEY_RETRO, BX_DISNEY_SHEEN, BX_DISNEY_CLEARCOAT  // rc/material/disney.rs:215-418
};
inline float disney_gtr1(float cos_theta, float alpha) {  // disney.rs:505-511 (log10, as the reference has it)
  float alpha2 = alpha * alpha;
  return (alpha2 - 1.0f) / (kPi * log10f(alpha2) * (1.0f + (alpha2 - 1.0f) * cos_theta * cos_theta));
}
inline float disney_smith_g_ggx(float cos_theta, float alpha) {  // disney.rs:513-519
  float alpha2 = alpha * alpha, cos_theta2 = cos_theta * cos_theta;
  return 1.0f / (cos_theta + sqrtf(alpha2 + cos_theta2 - alpha2 * cos_theta2));
}
struct SampleF { RGB f; V3 wi; float pdf; uint32_t type; };

inline float pow5(float v) { return (v * v) * (v * v) * v; }  // fresnel.rs:414-417

struct Bxdf {
  int kind = BX_LAMBERT_R;
  RGB r{0, 0, 0};   // R / T / (FresnelSpecular: R) / (FresnelBlend: Rd)
  RGB t{0, 0, 0};   // FresnelSpecular: T ; FresnelBlend: Rs
  float a = 0, b = 0;          // OrenNayar A,B; Disney FakeSS / Retro: a = roughness; ClearCoat: a = weight, b = gloss
  float eta_a = 1, eta_b = 1;  // transmission / FresnelSpecular
  Fresnel fresnel;             // SpecularReflection / MicrofacetReflection / (dielectric of *Transmission)
  TRDist dist;
  int n_scales = 0;            // nesting depth of ScaledBxDF wrappers (bxdf.rs:48-71), innermost first
  RGB scales[4] = {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}};
  void wrap_scaled(RGB s) { if (n_scales < 4) scales[n_scales++] = s; }

  uint32_t get_type() const {
    switch (kind) {
      case BX_LAMBERT_R: return BSDF_DIFFUSE | BSDF_REFLECTION;                    // lambertian.rs:23
      case BX_LAMBERT_T: return BSDF_DIFFUSE | BSDF_TRANSMISSION;                  // lambertian.rs:44
      case BX_OREN_NAYAR: return BSDF_REFLECTION | BSDF_DIFFUSE;                   // oren_nayar.rs:56
      case BX_SPEC_R: return BSDF_SPECULAR | BSDF_REFLECTION;                      // fresnel.rs:170
      case BX_SPEC_T: return BSDF_SPECULAR | BSDF_TRANSMISSION;                    // fresnel.rs:237
      case BX_FRESNEL_SPEC: return BSDF_SPECULAR | BSDF_REFLECTION | BSDF_TRANSMISSION;  // :331
      case BX_FRESNEL_BLEND: return BSDF_REFLECTION | BSDF_GLOSSY;                 // :410
      case BX_MICRO_R: return BSDF_REFLECTION | BSDF_GLOSSY;                       // microfacet.rs:56
      case BX_DISNEY_DIFFUSE: case BX_DISNEY_FAKESS: case BX_DISNEY_RETRO: case BX_DISNEY_SHEEN: return BSDF_REFLECTION | BSDF_DIFFUSE;
      case BX_DISNEY_CLEARCOAT: return BSDF_REFLECTION | BSDF_GLOSSY;              // disney.rs:415-417
      default: return BSDF_TRANSMISSION | BSDF_GLOSSY;                             // microfacet.rs:175
    }
  }
  bool matches(uint32_t flags) const { return (get_type() & flags) == get_type(); }  // bxdf.rs:29-31

  static float default_pdf(V3 wo, V3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f; }  // bxdf.rs:38-44

  RGB f_inner(V3 wo, V3 wi) const {
    switch (kind) {
      case BX_LAMBERT_R: case BX_LAMBERT_T: return r * kInvPi;  // lambertian.rs:19,40
      case BX_OREN_NAYAR: {                                     // oren_nayar.rs:31-53
        float sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
        float max_cos = 0.0f;
        if (sin_theta_i > 1e-4f && sin_theta_o > 1e-4f) {
          float sin_phi_i = sin_phi(wi), cos_phi_i = cos_phi(wi), sin_phi_o = sin_phi(wo), cos_phi_o = cos_phi(wo);
          float d_cos = sin_phi_i * sin_phi_o + cos_phi_i * cos_phi_o;
          max_cos = fmaxf(d_cos, 0.0f);
        }
        float sin_alpha, tan_beta;
        if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / abs_cos_theta(wi); }
        else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / abs_cos_theta(wo); }
        return r * kInvPi * (a + b * max_cos * sin_alpha * tan_beta);
      }
      case BX_SPEC_R: case BX_SPEC_T: case BX_FRESNEL_SPEC: return rgb(0, 0, 0);
      case BX_FRESNEL_BLEND: {  // fresnel.rs:357-374  (r = rd, t = rs)
        RGB diffuse = (28.0f / (23.0f * kPi)) * r * (rgb(1, 1, 1) - t) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) *
                      (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return rgb(0, 0, 0);
        wh = normalize(wh);
        float ct = dot(wi, wh);
        RGB schlick = t + pow5(1.0f - ct) * (rgb(1, 1, 1) - t);  // :351-353
        RGB specular = dist.d(wh) / (4.0f * fabsf(dot(wi, wh)) * fmaxf(abs_cos_theta(wi), abs_cos_theta(wo))) * schlick;
        return diffuse + specular;
      }
      case BX_MICRO_R: {  // microfacet.rs:36-53
        float cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
        V3 wh = wi + wo;
        if (cos_theta_o == 0.0f || cos_theta_i == 0.0f) return rgb(0, 0, 0);
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return rgb(0, 0, 0);
        wh = normalize(wh);
        RGB fr = fresnel.evaluate(dot(wi, wh));
        return r * dist.d(wh) * dist.g(wo, wi) * fr / (4.0f * cos_theta_i * cos_theta_o);
      }
      case BX_DISNEY_DIFFUSE: {  // disney.rs:228-236
        float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        return r * kInvPi * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
      }
      case BX_DISNEY_FAKESS: case BX_DISNEY_RETRO: case BX_DISNEY_SHEEN: case BX_DISNEY_CLEARCOAT: {
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return rgb(0, 0, 0);
        wh = normalize(wh);
        float cos_theta_d = dot(wi, wh);
        if (kind == BX_DISNEY_SHEEN) return r * schlick_weight(cos_theta_d);  // :332-341
        if (kind == BX_DISNEY_CLEARCOAT) {  // :363-378
          float Dr = disney_gtr1(abs_cos_theta(wh), b);
          float Fr = fr_schlick(0.04f, dot(wo, wh));
          float Gr = disney_smith_g_ggx(abs_cos_theta(wo), 0.25f) * disney_smith_g_ggx(abs_cos_theta(wi), 0.25f);
          return grey(a * Gr * Fr * Dr / 4.0f);
        }
        float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        if (kind == BX_DISNEY_FAKESS) {  // :258-274
          float fss90 = cos_theta_d * cos_theta_d * a;
          float fss = lerp_f(fo, 1.0f, fss90) * lerp_f(fi, 1.0f, fss90);
          float ss = 1.25f * (fss * (1.0f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - 0.5f) + 0.5f);
          return r * kInvPi * ss;
        }
        float rr = 2.0f * a * cos_theta_d * cos_theta_d;  // DisneyRetro, :296-308
        return r * kInvPi * rr * (fo + fi + fo * fi * (rr - 1.0f));
      }
      default: {  // BX_MICRO_T microfacet.rs:127-172  (r = T)
        if (same_hemisphere(wo, wi)) return rgb(0, 0, 0);
        float cos_theta_o = cos_theta(wo), cos_theta_i = cos_theta(wi);
        if (cos_theta_o == 0.0f || cos_theta_i == 0.0f) return rgb(0, 0, 0);
        float eta = cos_theta_o > 0.0f ? eta_b / eta_a : eta_a / eta_b;
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        RGB fr = fresnel.evaluate(dot(wo, wh));
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1.0f / eta;  // TransportMode::RADIANCE
        return (rgb(1, 1, 1) - fr) * r *
               fabsf(dist.d(wh) * dist.g(wo, wi) * eta * eta * fabsf(dot(wi, wh)) * fabsf(dot(wo, wh)) * factor * factor /
                     (cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom));
      }
    }
  }
  float pdf_inner(V3 wo, V3 wi) const {
    switch (kind) {
      case BX_SPEC_R: case BX_SPEC_T: case BX_FRESNEL_SPEC: return 0.0f;
      case BX_FRESNEL_BLEND: {  // fresnel.rs:376-384
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        float pdf_wh = dist.pdf(wo, wh);
        return 0.5f * (abs_cos_theta(wi) * kInvPi + pdf_wh / (4.0f * dot(wo, wh)));
      }
      case BX_MICRO_R: {  // microfacet.rs:86-93
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        return dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
      }
      case BX_MICRO_T: {  // microfacet.rs:210-226
        if (same_hemisphere(wo, wi)) return 0.0f;
        float eta = cos_theta(wo) > 0.0f ? eta_b / eta_a : eta_a / eta_b;
        V3 wh = normalize(wo + wi * eta);
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
        return dist.pdf(wo, wh) * dwh_dwi;
      }
      case BX_DISNEY_CLEARCOAT: {  // disney.rs:397-413
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = wo + wi;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
        wh = normalize(wh);
        float Dr = disney_gtr1(abs_cos_theta(wh), b);
        return Dr * abs_cos_theta(wh) / (4.0f * dot(wo, wh));
      }
      default: return default_pdf(wo, wi);  // Lambertian R/T (quirk 6), OrenNayar, Disney diffuse lobes
    }
  }
  SampleF sample_inner(V3 wo, P2 u) const {
    switch (kind) {
      case BX_SPEC_R: {  // fresnel.rs:158-163
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        RGB s = fresnel.evaluate(cos_theta(wi)) * r / abs_cos_theta(wi);
        return {s, wi, 1.0f, get_type()};
      }
      case BX_SPEC_T: {  // fresnel.rs:202-230
        bool entering = cos_theta(wo) > 0.0f;
        float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
        V3 wi;
        if (refract(wo, face_forward(v3(0, 0, 1), wo), eta_i / eta_t, &wi)) {
          RGB ft = r * (rgb(1, 1, 1) - fresnel.evaluate(cos_theta(wi)));
          ft = ft * (eta_i * eta_i) / (eta_t * eta_t);
          return {ft / abs_cos_theta(wi), wi, 1.0f, get_type()};
        }
        return {rgb(1, 1, 1), v3(0, 0, 0), 0.0f, 0u};
      }
      case BX_FRESNEL_SPEC: {  // fresnel.rs:275-324
        float fr = fr_dielectric(cos_theta(wo), eta_a, eta_b);
        if (u.x < fr) {
          V3 wi = v3(-wo.x, -wo.y, wo.z);
          return {fr * r / abs_cos_theta(wi), wi, fr, BSDF_SPECULAR | BSDF_REFLECTION};
        }
        bool entering = cos_theta(wo) > 0.0f;
        float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
        V3 wi;
        if (refract(wo, face_forward(v3(0, 0, 1), wo), eta_i / eta_t, &wi)) {
          RGB ft = t * (1.0f - fr);
          ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
          return {ft / abs_cos_theta(wi), wi, 1.0f - fr, BSDF_SPECULAR | BSDF_TRANSMISSION};
        }
        return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, 0u};
      }
      case BX_FRESNEL_BLEND: {  // fresnel.rs:386-407
        V3 wi;
        if (u.x < 0.5f) {
          u.x = fminf(2.0f * u.x, kOneMinusEpsilon);
          wi = cosine_sample_hemisphere(u);
          if (wo.z < 0.0f) wi.z *= -1.0f;
        } else {
          u.x = fminf(2.0f * (u.x - 0.5f), kOneMinusEpsilon);
          V3 wh = dist.sample_wh(wo, u);
          wi = reflect(wo, wh);
          if (!same_hemisphere(wo, wi)) return {rgb(0, 0, 0), wi, 0.0f, get_type()};
        }
        return {f_inner(wo, wi), wi, pdf_inner(wo, wi), get_type()};
      }
      case BX_MICRO_R: {  // microfacet.rs:61-84 (quirk 8: no wo·wh<0 rejection)
        if (wo.z == 0.0f) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
        V3 wh = dist.sample_wh(wo, u);
        V3 wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
        float pdf = dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
        return {f_inner(wo, wi), wi, pdf, get_type()};
      }
      case BX_MICRO_T: {  // microfacet.rs:180-208
        if (wo.z == 0.0f) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
        V3 wh = dist.sample_wh(wo, u);
        float eta = cos_theta(wo) > 0.0f ? eta_a / eta_b : eta_b / eta_a;
        V3 wi;
        if (refract(wo, wh, eta, &wi)) {
          float pdf = pdf_inner(wo, wi);
          return {f_inner(wo, wi), wi, pdf, get_type()};
        }
        return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
      }
      case BX_DISNEY_CLEARCOAT: {  // disney.rs:380-395
        if (wo.z == 0.0f) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
        float alpha2 = b * b;
        float cos_theta_ = sqrtf(fmaxf(0.0f, (1.0f - powf(alpha2, 1.0f - u.x)) / (1.0f - alpha2)));
        float sin_theta_ = sqrtf(fmaxf(0.0f, 1.0f - cos_theta_ * cos_theta_));
        float phi = 2.0f * kPi * u.y;
        V3 wh = v3(sin_theta_ * cosf(phi), sin_theta_ * sinf(phi), cos_theta_);  // spherical_direction, geometry/mod.rs:112-114
        if (!same_hemisphere(wo, wh)) wh = -wh;
        V3 wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, get_type()};
        float pdf = pdf_inner(wo, wi);
        return {f_inner(wo, wi), wi, pdf, get_type()};
      }
      default: {  // default trait sample_f, bxdf.rs:18-25 (quirk 5: returns empty type flags)
        V3 wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        float pdf = pdf_inner(wo, wi);
        return {f_inner(wo, wi), wi, pdf, 0u};
      }
    }
  }
  // ScaledBxDF wrapper semantics (bxdf.rs:59-71): f and sample_f scale; pdf is the DEFAULT
  // cosine pdf, not the wrapped lobe's (quirk 10).
  RGB f(V3 wo, V3 wi) const { RGB v = f_inner(wo, wi); for (int i = 0; i < n_scales; ++i) v = v * scales[i]; return v; }
  float pdf(V3 wo, V3 wi) const { return n_scales > 0 ? default_pdf(wo, wi) : pdf_inner(wo, wi); }
  SampleF sample_f(V3 wo, P2 u) const {
    SampleF s = sample_inner(wo, u);
    for (int i = 0; i < n_scales; ++i) s.f = s.f * scales[i];
    return s;
  }
};

inline Bxdf make_oren_nayar(RGB r, float sigma) {  // oren_nayar.rs:17-27
  Bxdf b; b.kind = BX_OREN_NAYAR; b.r = r;
  float sigma_rad = to_radians(sigma);
  float sigma2 = sigma_rad * sigma_rad;
  b.a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
  b.b = 0.45f * sigma2 / (sigma2 + 0.09f);
  return b;
}

// ---------------------------------------------------------------- Bsdf (bsdf/mod.rs:64-269)
struct Bsdf {
  float eta = 1.0f;
  V3 ns, ng, ss, ts;
  Bxdf bxdfs[8];
  int n = 0;
  void add(const Bxdf& b) { if (n < 8) bxdfs[n++] = b; }
  void init_frame(V3 shading_n, V3 geom_n, V3 shading_dpdu) {  // :77-91
    ss = normalize(shading_dpdu);
    ns = shading_n; ng = geom_n;
    ts = cross(shading_n, ss);
  }
  V3 world_to_local(V3 v) const { return v3(dot(v, ss), dot(v, ts), dot(v, ns)); }  // :253-255
  V3 local_to_world(V3 v) const {                                                    // :257-263
    return v3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
  }
  int num_components(uint32_t flags) const { int c = 0; for (int i = 0; i < n; ++i) if (bxdfs[i].matches(flags)) ++c; return c; }
  RGB f(V3 wo_w, V3 wi_w, uint32_t flags) const {  // :94-111
    V3 wi = world_to_local(wi_w), wo = world_to_local(wo_w);
    if (wo.z == 0.0f) return rgb(0, 0, 0);
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
    RGB c = rgb(0, 0, 0);
    for (int i = 0; i < n; ++i) {
      const Bxdf& b = bxdfs[i];
      if (b.matches(flags) && ((refl && (b.get_type() & BSDF_REFLECTION)) || (!refl && (b.get_type() & BSDF_TRANSMISSION))))
        c = c + b.f(wo, wi);
    }
    return c;
  }
  float pdf(V3 wo_w, V3 wi_w, uint32_t flags) const {  // :113-136
    if (n == 0) return 0.0f;
    V3 wo = world_to_local(wo_w);
    if (wo.z == 0.0f) return 0.0f;
    V3 wi = world_to_local(wi_w);
    int matched = 0; float p = 0.0f;
    for (int i = 0; i < n; ++i) if (bxdfs[i].matches(flags)) { ++matched; p += bxdfs[i].pdf(wo, wi); }
    return matched == 0 ? 0.0f : p / (float)matched;
  }
  SampleF sample_f(V3 wo_w, P2 u, uint32_t flags) const {  // :138-251
    int idx[8]; int m = 0;
    for (int i = 0; i < n; ++i) if (bxdfs[i].matches(flags)) idx[m++] = i;
    if (m == 0) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, 0u};
    int comp = (int)f2u_sat(floorf(u.x * (float)m));
    if (comp > m - 1) comp = m - 1;
    const Bxdf& bx = bxdfs[idx[comp]];
    P2 ur{fminf(u.x * (float)m - (float)comp, kOneMinusEpsilon), u.y};
    V3 wo = world_to_local(wo_w);
    if (wo.z == 0.0f) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, bx.get_type()};
    SampleF s = bx.sample_f(wo, ur);
    if (s.pdf == 0.0f) return {rgb(0, 0, 0), v3(0, 0, 0), 0.0f, 0u};
    V3 wi = s.wi;
    V3 wi_w = local_to_world(wi);
    float pdf = s.pdf;
    if (!(bx.get_type() & BSDF_SPECULAR) && m > 1)
      for (int i = 0; i < m; ++i) if (i != comp) pdf += bxdfs[idx[i]].pdf(wo, wi);
    if (m > 1) pdf /= (float)m;
    RGB f = s.f;
    if (!(bx.get_type() & BSDF_SPECULAR)) {
      bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
      f = rgb(0, 0, 0);
      for (int i = 0; i < m; ++i) {
        const Bxdf& b = bxdfs[idx[i]];
        if ((refl && (b.get_type() & BSDF_REFLECTION)) || (!refl && (b.get_type() & BSDF_TRANSMISSION))) f = f + b.f(wo, wi);
      }
    }
    return {f, wi_w, pdf, s.type};
  }
};

}  // namespace orc
