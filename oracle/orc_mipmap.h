// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// MIPMap (rc/mipmap.rs; rc/blockedarray.rs is only a memory layout and is restated as plain
// row-major storage), ImageTexture (rc/texture/imagemap.rs:232-235 + UVMapping2D rc/texture/mod.rs:52-60)
// and InfiniteAreaLight (rc/light/infinite.rs).
// Images whose sides are not powers of two are first re-sampled with the reference's 4-tap Lanczos zoom
// (mipmap.rs:75-139, 362-408).
#pragma once
#include <vector>
#include "orc_scene.h"

namespace orc {

enum WrapMode { WRAP_REPEAT = 0, WRAP_BLACK = 1, WRAP_CLAMP = 2 };

struct MipLevel { int u = 0, v = 0; std::vector<RGB> data; };

struct MipMap {
  bool do_trilinear = false;
  float max_anisotropy = 8.0f;
  int wrap = WRAP_REPEAT;
  int res_x = 0, res_y = 0;
  std::vector<MipLevel> pyramid;
  float weight_lut[128];

  static long modulo(long a, long b) { long r = a % b; return r < 0 ? r + b : r; }  // mipmap.rs:428-435
  RGB texel(int level, long s, long t) const {  // :208-225
    const MipLevel& l = pyramid[level];
    long us = l.u, vs = l.v, ss, tt;
    if (wrap == WRAP_REPEAT) { ss = modulo(s, us); tt = modulo(t, vs); }
    else if (wrap == WRAP_CLAMP) { ss = clamp_t<long>(s, 0, us - 1); tt = clamp_t<long>(t, 0, vs - 1); }
    else { if (s < 0 || s >= us || t < 0 || t >= vs) return rgb(0, 0, 0); ss = s; tt = t; }
    return l.data[(size_t)tt * l.u + ss];
  }
  int levels() const { return (int)pyramid.size(); }

  static float lanczos(float f) {  // mipmap.rs:395-408
    float tau = 2.0f, x = fabsf(f);
    if (x < 1e-5f) return 1.0f;
    if (x > 1.0f) return 0.0f;
    x *= kPi;
    float s = sinf(x * tau) / (x * tau);
    float l = sinf(x) / x;
    return s * l;
  }
  struct ResampleWeight { int first_texel; float w[4]; };
  static std::vector<ResampleWeight> resample_weights(int old_res, int new_res) {  // mipmap.rs:362-393
    std::vector<ResampleWeight> wt((size_t)new_res);
    const float filter_width = 2.0f;
    for (int i = 0; i < new_res; ++i) {
      float center = ((float)i + 0.5f) * (float)old_res / (float)new_res;
      float first = floorf((center - filter_width) + 0.5f);
      float w[4];
      for (int j = 0; j < 4; ++j) { float pos = first + (float)j + 0.5f; w[j] = lanczos((pos - center) / filter_width); }
      float inv = 1.0f / (w[0] + w[1] + w[2] + w[3]);
      wt[(size_t)i].first_texel = f2i_sat(first);
      for (int j = 0; j < 4; ++j) wt[(size_t)i].w[j] = w[j] * inv;
    }
    return wt;
  }
  static int round_up_pow2(int v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }  // lib.rs:215-224
  long wrap_index(long i, long n) const { return wrap == WRAP_REPEAT ? modulo(i, n) : (wrap == WRAP_CLAMP ? clamp_t<long>(i, 0, n - 1) : i); }

  // mipmap.rs:67-194. `img` is row-major, res.x * res.y texels.
  void init(int rx, int ry, const RGB* img_in, bool trilinear, float max_aniso, int wrap_mode) {
    do_trilinear = trilinear; max_anisotropy = max_aniso; wrap = wrap_mode;
    std::vector<RGB> resampled;
    const RGB* img = img_in;
    if ((rx & (rx - 1)) != 0 || (ry & (ry - 1)) != 0) {  // :75-139
      const int px = round_up_pow2(rx), py = round_up_pow2(ry);
      resampled.assign((size_t)px * py, rgb(0, 0, 0));
      std::vector<ResampleWeight> sw = resample_weights(rx, px);
      for (int t = 0; t < ry; ++t)  // zoom in s: only the first res.y rows are filled
        for (int s = 0; s < px; ++s)
          for (int j = 0; j < 4; ++j) {
            long o = wrap_index((long)sw[(size_t)s].first_texel + j, rx);
            if (o >= 0 && o < rx) resampled[(size_t)t * px + s] = resampled[(size_t)t * px + s] + img_in[(size_t)t * rx + o] * sw[(size_t)s].w[j];
          }
      std::vector<ResampleWeight> tw = resample_weights(ry, py);
      std::vector<RGB> work((size_t)py);
      for (int s = 0; s < px; ++s) {  // zoom in t, column by column
        for (int t = 0; t < py; ++t) {
          work[(size_t)t] = rgb(0, 0, 0);
          for (int j = 0; j < 4; ++j) {
            long o = wrap_index((long)tw[(size_t)t].first_texel + j, ry);
            if (o >= 0 && o < ry) work[(size_t)t] = work[(size_t)t] + resampled[(size_t)o * px + s] * tw[(size_t)t].w[j];
          }
        }
        for (int t = 0; t < py; ++t) resampled[(size_t)t * px + s] = clamp_pos(work[(size_t)t]);  // Clampable::clamp(0, inf), lib.rs:287-295
      }
      rx = px; ry = py; img = resampled.data();
    }
    res_x = rx; res_y = ry;
    for (int i = 0; i < 128; ++i) {  // :33-44
      float alpha = 2.0f;
      float r2 = (float)i / (128.0f - 1.0f);
      weight_lut[i] = expf(-alpha * r2) - expf(-alpha);
    }
    int n_levels = 1 + (int)f2u_sat(log2f((float)(rx > ry ? rx : ry)));  // :159
    pyramid.clear();
    MipLevel l0; l0.u = rx; l0.v = ry; l0.data.assign(img, img + (size_t)rx * ry);
    pyramid.push_back(std::move(l0));
    for (int i = 1; i < n_levels; ++i) {  // :168-187
      int s_res = pyramid[i - 1].u / 2 > 1 ? pyramid[i - 1].u / 2 : 1;
      int t_res = pyramid[i - 1].v / 2 > 1 ? pyramid[i - 1].v / 2 : 1;
      MipLevel l; l.u = s_res; l.v = t_res; l.data.resize((size_t)s_res * t_res);
      for (int t = 0; t < t_res; ++t)
        for (int s = 0; s < s_res; ++s)
          l.data[(size_t)t * s_res + s] =
              (texel(i - 1, 2 * s, 2 * t) + texel(i - 1, 2 * s + 1, 2 * t) + texel(i - 1, 2 * s, 2 * t + 1) + texel(i - 1, 2 * s + 1, 2 * t + 1)) * 0.25f;
      pyramid.push_back(std::move(l));
    }
  }
  RGB triangle(int level, P2 st) const {  // :285-308
    level = clamp_t(level, 0, levels() - 1);
    float s = st.x * (float)pyramid[level].u - 0.5f;
    float t = st.y * (float)pyramid[level].v - 0.5f;
    long s0 = (long)f2i_sat(floorf(s)), t0 = (long)f2i_sat(floorf(t));
    float ds = s - (float)s0, dt = t - (float)t0;
    return texel(level, s0, t0) * (1.0f - ds) * (1.0f - dt) + texel(level, s0, t0 + 1) * (1.0f - ds) * dt +
           texel(level, s0 + 1, t0) * ds * (1.0f - dt) + texel(level, s0 + 1, t0 + 1) * ds * dt;
  }
  static RGB lerp_rgb(float t, RGB a, RGB b) { return a * (1.0f - t) + b * t; }  // lib.rs:107-117
  RGB lookup(P2 st, float width) const {  // :227-245
    float level = (float)levels() - 1.0f + log2f(fmaxf(width, 1e-8f));
    if (level < 0.0f) return triangle(0, st);
    if (level >= (float)levels() - 1.0f) return texel(levels() - 1, 0, 0);
    float i_level = floorf(level);
    float delta = level - i_level;
    return lerp_rgb(delta, triangle((int)f2u_sat(i_level), st), triangle((int)f2u_sat(i_level) + 1, st));
  }
  RGB ewa(int level, P2 st, P2 dst0, P2 dst1) const {  // :310-360
    if (level >= levels()) return texel(levels() - 1, 0, 0);
    float us = (float)pyramid[level].u, vs = (float)pyramid[level].v;
    st.x = st.x * us - 0.5f; st.y = st.y * vs - 0.5f;
    dst0.x *= us; dst0.y *= vs; dst1.x *= us; dst1.y *= vs;
    float A = dst0.y * dst0.y + dst1.y * dst1.y + 1.0f;
    float B = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
    float C = dst0.x * dst0.x + dst1.x * dst1.x + 1.0f;
    float invF = 1.0f / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    float det = -B * B + 4.0f * A * C;
    float invDet = 1.0f / det;
    float uSqrt = sqrtf(det * C), vSqrt = sqrtf(A * det);
    long s0 = f2i_sat(ceilf(st.x - 2.0f * invDet * uSqrt)), s1 = f2i_sat(floorf(st.x + 2.0f * invDet * uSqrt));
    long t0 = f2i_sat(ceilf(st.y - 2.0f * invDet * vSqrt)), t1 = f2i_sat(floorf(st.y + 2.0f * invDet * vSqrt));
    RGB sum = rgb(0, 0, 0); float sumWts = 0.0f;
    for (long it = t0; it < t1 + 1; ++it) {
      float tt = (float)it - st.y;
      for (long is = s0; is < s1 + 1; ++is) {
        float ss = (float)is - st.x;
        float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
        if (r2 < 1.0f) {
          size_t index = (size_t)f2u_sat(r2 * 128.0f); if (index > 127) index = 127;
          float weight = weight_lut[index];
          sum = sum + texel(level, is, it) * weight;
          sumWts += weight;
        }
      }
    }
    return sum / sumWts;
  }
  RGB lookup_diff(P2 st, P2 dst0, P2 dst1) const {  // :247-283
    if (do_trilinear) {
      float width = fmaxf(fmaxf(fabsf(dst0.x), fabsf(dst0.y)), fmaxf(fabsf(dst1.x), fabsf(dst1.y)));
      return lookup(st, 2.0f * width);
    }
    auto l2 = [](P2 v) { return v.x * v.x + v.y * v.y; };
    if (l2(dst0) < l2(dst1)) { P2 t = dst0; dst0 = dst1; dst1 = t; }
    float major_length = sqrtf(l2(dst0));
    float minor_length = sqrtf(l2(dst1));
    if ((minor_length * max_anisotropy) < major_length && minor_length > 0.0f) {
      float scale = major_length / (minor_length * max_anisotropy);
      dst1.x *= scale; dst1.y *= scale;
      minor_length *= scale;
    }
    if (minor_length == 0.0f) return triangle(0, st);
    float lod = fmaxf(0.0f, (float)levels() - 1.0f + log2f(minor_length));
    int ilod = (int)f2u_sat(floorf(lod));
    return lerp_rgb(lod - (float)ilod, ewa(ilod, st, dst0, dst1), ewa(ilod + 1, st, dst0, dst1));
  }
};

inline RGB image_tex_eval(const Scene& sc, const Texture& t, const SurfaceInteraction& si) {
  // UVMapping2D::map, texture/mod.rs:52-60
  P2 st{t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv};
  P2 dstdx{t.su * si.dudx, t.sv * si.dvdx};
  P2 dstdy{t.su * si.dudy, t.sv * si.dvdy};
  RGB v = sc.mips[t.mip]->lookup_diff(st, dstdx, dstdy);
  if (t.is_float) { float y = v.r; return rgb(y, y, y); }  // float image textures are stored as Y in .r
  return v;
}

// ---------------------------------------------------------------- InfiniteAreaLight
inline std::shared_ptr<Distribution2D> infinite_build_distribution(const MipMap& lmap) {  // infinite.rs:78-101
  int width = 2 * lmap.res_x, height = 2 * lmap.res_y;
  float filter = 0.5f / fminf((float)width, (float)height);
  std::vector<float> img((size_t)width * height);
  for (int v = 0; v < height; ++v) {
    float vp = ((float)v + 0.5f) / (float)height;
    float sin_theta = sinf(kPi * ((float)v + 0.5f) / (float)height);
    for (int u = 0; u < width; ++u) {
      float up = ((float)u + 0.5f) / (float)width;
      img[(size_t)v * width + u] = lum_y(lmap.lookup(P2{up, vp}, filter)) * sin_theta;
    }
  }
  auto d = std::make_shared<Distribution2D>();
  d->init(img.data(), (size_t)width, (size_t)height);
  return d;
}
inline Scene::LiSample infinite_sample_li(const Scene& sc, const Light& l, const Interaction& it, P2 u) {  // :143-181
  Scene::LiSample s; s.p0 = it;
  float map_pdf;
  P2 uv = l.distribution->sample_continuous(u, &map_pdf);
  if (map_pdf == 0.0f) { s.li = rgb(0, 0, 0); s.wi = v3(0, 0, 0); s.pdf = 0.0f; s.p1 = Interaction{v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)}; return s; }
  float theta = uv.y * kPi, phi = uv.x * 2.0f * kPi;
  float cos_theta_ = cosf(theta), sin_theta_ = sinf(theta), cos_phi_ = cosf(phi), sin_phi_ = sinf(phi);
  V3 wi = xf_vector(l.l2w, v3(sin_theta_ * cos_phi_, sin_theta_ * sin_phi_, cos_theta_));
  s.pdf = sin_theta_ == 0.0f ? 0.0f : map_pdf / (2.0f * kPi * kPi * sin_theta_);
  V3 target = it.p + wi * (2.0f * l.w_radius);
  s.p1 = Interaction{target, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
  s.li = sc.mips[l.mip]->lookup(uv, 0.0f); s.wi = wi;
  return s;
}
inline float infinite_pdf_li(const Scene&, const Light& l, const Interaction&, V3 w) {  // :183-196
  V3 wi = xf_vector(l.w2l, w);
  float theta = spherical_theta(wi), phi = spherical_phi(wi);
  float sin_theta_ = sinf(theta);
  if (sin_theta_ == 0.0f) return 0.0f;
  return l.distribution->pdf(P2{phi * kInvPi * 0.5f, theta * kInvPi}) / (2.0f * kPi * kPi * sin_theta_);
}
inline RGB infinite_le(const Scene& sc, const Light& l, const Ray& ray) {  // :211-219
  V3 w = normalize(xf_vector(l.w2l, ray.d));
  P2 st{spherical_phi(w) * kInvPi * 0.5f, spherical_theta(w) * kInvPi};
  return sc.mips[l.mip]->lookup(st, 0.0f);
}

}  // namespace orc
