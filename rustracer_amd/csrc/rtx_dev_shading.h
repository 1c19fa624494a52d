// rtx_dev_shading.h — textures / MIP lookups, materials -> Bsdf, lights and light distributions.
// Restates rc/texture/*.rs, rc/mipmap.rs:208-360, rc/material/*.rs, rc/light/*.rs,
// rc/lightdistrib.rs, rc/sampling/distribution{1,2}d.rs.
#pragma once
#include "rtx_dev_bsdf.h"
#include "rtx_dev_scene.h"
#include "rtx_dev_sphere.h"

namespace rtx {

// ---------------------------------------------------------------- MIPMap lookups (rc/mipmap.rs)
static __device__ __constant__ float kEwaLut[128];  // filled by the host at start-up: exp(-2 r2) - exp(-2) (:33-44). One copy per translation unit (rtx_hip.hip, rtx_shade.hip): rtx_fill_ewa_lut_here() in each

// MIPMap::texel (:208-225). Level sizes are powers of two (MIPMap::new resamples, :75-139; checked at rt_scene_create), so the Repeat wrap
// `modulo(s, size)` (:428-435) is the low bits of s - for negative s too, in two's complement - instead of a 64-bit signed division per texel.
RT_DEV rgb3 mip_texel(const DImage& im, int level, long s, long t) {
  const int us = im.w[level], vs = im.h[level];
  unsigned ss, tt;
  if (im.wrap == 0) { ss = (unsigned)s & (unsigned)(us - 1); tt = (unsigned)t & (unsigned)(vs - 1); }
  else if (im.wrap == 2) { ss = (unsigned)(s < 0 ? 0 : (s > us - 1 ? us - 1 : s)); tt = (unsigned)(t < 0 ? 0 : (t > vs - 1 ? vs - 1 : t)); }
  else { if (s < 0 || s >= us || t < 0 || t >= vs) return mkc(0, 0, 0); ss = (unsigned)s; tt = (unsigned)t; }
  const float4 v = im.texels[im.off[level] + ((((unsigned long long)(tt >> 1) << im.tshift[level]) + (ss >> 2)) << 3) + ((tt & 1u) << 2) + (ss & 3u)];
  return mkc(v.x, v.y, v.z);
}
RT_DEV rgb3 mip_triangle(const DImage& im, int level, f2 st) {  // :285-308
  level = clampi(level, 0, im.n_levels - 1);
  float s = st.x * (float)im.w[level] - 0.5f;
  float t = st.y * (float)im.h[level] - 0.5f;
  long s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
  float ds = s - (float)s0, dt = t - (float)t0;
  return mip_texel(im, level, s0, t0) * (1.0f - ds) * (1.0f - dt) + mip_texel(im, level, s0, t0 + 1) * (1.0f - ds) * dt +
         mip_texel(im, level, s0 + 1, t0) * ds * (1.0f - dt) + mip_texel(im, level, s0 + 1, t0 + 1) * ds * dt;
}
RT_DEV rgb3 lerp_rgb(float t, rgb3 a, rgb3 b) { return a * (1.0f - t) + b * t; }
RT_DEV rgb3 mip_lookup(const DImage& im, f2 st, float width) {  // :227-245
  float level = (float)im.n_levels - 1.0f + log2f(fmaxf(width, 1e-8f));
  if (level < 0.0f) return mip_triangle(im, 0, st);
  if (level >= (float)im.n_levels - 1.0f) return mip_texel(im, im.n_levels - 1, 0, 0);
  float i_level = floorf(level);
  float delta = level - i_level;
  int il = (int)f2u_sat(i_level);
  rgb3 v[2];
  for (int k = 0; k < 2; ++k) v[k] = mip_triangle(im, il + k, st);  // a loop: one copy of the inlined lookup
  return lerp_rgb(delta, v[0], v[1]);
}
RT_DEV rgb3 mip_ewa(const DImage& im, int level, f2 st, f2 dst0, f2 dst1) {  // :310-360
  if (level >= im.n_levels) return mip_texel(im, im.n_levels - 1, 0, 0);
  float us = (float)im.w[level], vs = (float)im.h[level];
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
  rgb3 sum = mkc(0, 0, 0); float sumWts = 0.0f;
  for (long it = t0; it < t1 + 1; ++it) {
    float tt = (float)it - st.y;
    for (long is = s0; is < s1 + 1; ++is) {
      float ss = (float)is - st.x;
      float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
      if (r2 < 1.0f) {
        unsigned index = f2u_sat(r2 * 128.0f); if (index > 127u) index = 127u;
        float weight = kEwaLut[index];
        sum = sum + mip_texel(im, level, is, it) * weight;
        sumWts += weight;
      }
    }
  }
  return sum / sumWts;
}
RT_DEV rgb3 mip_lookup_diff(const DImage& im, f2 st, f2 dst0, f2 dst1) {  // :247-283
  if (im.trilinear) {
    float width = fmaxf(fmaxf(fabsf(dst0.x), fabsf(dst0.y)), fmaxf(fabsf(dst1.x), fabsf(dst1.y)));
    return mip_lookup(im, st, 2.0f * width);
  }
  float l0 = dst0.x * dst0.x + dst0.y * dst0.y, l1 = dst1.x * dst1.x + dst1.y * dst1.y;
  if (l0 < l1) { f2 t = dst0; dst0 = dst1; dst1 = t; }
  float major_length = sqrtf(dst0.x * dst0.x + dst0.y * dst0.y);
  float minor_length = sqrtf(dst1.x * dst1.x + dst1.y * dst1.y);
  if ((minor_length * im.max_aniso) < major_length && minor_length > 0.0f) {
    float scale = major_length / (minor_length * im.max_aniso);
    dst1.x *= scale; dst1.y *= scale;
    minor_length *= scale;
  }
  if (minor_length == 0.0f) return mip_triangle(im, 0, st);
  float lod = fmaxf(0.0f, (float)im.n_levels - 1.0f + log2f(minor_length));
  int ilod = (int)f2u_sat(floorf(lod));
  rgb3 v[2];
  for (int k = 0; k < 2; ++k) v[k] = mip_ewa(im, ilod + k, st, dst0, dst1);  // a loop: one copy of the inlined filter
  return lerp_rgb(lod - (float)ilod, v[0], v[1]);
}

// ---------------------------------------------------------------- textures (rc/texture/*.rs)
// Scale/Mix operands are resolved iteratively with a tiny explicit stack (no device recursion).
// ---- Perlin noise (rc/noise.rs)
static __device__ __constant__ const unsigned char kNoisePerm[512] = {
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
RT_DEV float noise_grad(int x, int y, int z, float dx, float dy, float dz) {  // noise.rs:68-76
  int h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z];
  h &= 15;
  float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
  float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
  return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
RT_DEV float noise_weight(float t) { float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }  // :78-83
RT_DEVN float noise_perlin(float x, float y, float z) {  // noise.rs:8-43
  int ix = f2i_sat(floorf(x)), iy = f2i_sat(floorf(y)), iz = f2i_sat(floorf(z));
  float dx = x - (float)ix, dy = y - (float)iy, dz = z - (float)iz;
  ix &= 255; iy &= 255; iz &= 255;
  float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
  float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
  float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
  float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
  float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
  float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
  float y0 = lerpf(wy, x00, x10), y1 = lerpf(wy, x01, x11);
  return lerpf(wz, y0, y1);
}
RT_DEV float noise_fbm(f3 p, f3 dpdx, f3 dpdy, float omega, unsigned max_octaves) {  // noise.rs:46-66
  float l2 = fmaxf(len2(dpdx), len2(dpdy));
  float n = clampf(-1.0f - 0.5f * log2f(l2), 0.0f, (float)max_octaves);
  unsigned n_int = f2u_sat(floorf(n));
  float sum = 0.0f, lambda = 1.0f, o = 1.0f;
  for (unsigned i = 0; i < n_int; ++i) {
    sum += o * noise_perlin(lambda * p.x, lambda * p.y, lambda * p.z);
    lambda *= 1.99f;
    o *= omega;
  }
  float n_partial = n - (float)n_int;
  float v = clampf((n_partial - 0.3f) / (0.7f - 0.3f), 0.0f, 1.0f);  // smooth_step, :85-89
  sum += o * (v * v * (-2.0f * v + 3.0f)) * noise_perlin(lambda * p.x, lambda * p.y, lambda * p.z);
  return sum;
}

// What a texture reads of the SurfaceInteraction (texture/mod.rs:52-60 UVMapping2D, noise.rs): 15 dwords, handed to the out-of-line evaluator by value in
// registers - a reference would pin the caller's whole SurfaceInteraction in scratch.
struct TexIn { f2 uv; float dudx, dvdx, dudy, dvdy; f3 p, dpdx, dpdy; };
RT_DEV TexIn tex_in(const SurfaceInteraction& si) {
  TexIn q; q.uv = si.uv; q.dudx = si.dudx; q.dvdx = si.dvdx; q.dudy = si.dudy; q.dvdy = si.dvdy; q.p = si.hit.p; q.dpdx = si.dpdx; q.dpdy = si.dpdy; return q;
}
// Leaves: constant, imagemap, uv, fbm. Combinators (operands = other textures): scale, mix, checkerboard.
RT_DEV bool tex_is_leaf(int kind) { return kind == 0 || kind == 3 || kind == 5 || kind == 6; }
RT_DEV rgb3 tex_leaf(const DImage* images, const DTexture& t, const TexIn& si) {
  if (t.kind == 0) return mkc(t.v[0], t.v[1], t.v[2]);  // constant.rs:35-38
  if (t.kind == 6) { float f = noise_fbm(si.p, si.dpdx, si.dpdy, t.v[0], (unsigned)(t.amount < 0 ? 0 : t.amount)); return mkc(f, f, f); }  // fbm.rs:18-21
  f2 st = mk2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv);  // UVMapping2D (texture/mod.rs:52-60)
  if (t.kind == 5) return mkc(st.x - floorf(st.x), st.y - floorf(st.y), 0.0f);  // uv.rs:50-54
  // imagemap.rs:232-235
  f2 dstdx = mk2(t.su * si.dudx, t.sv * si.dvdx), dstdy = mk2(t.su * si.dudy, t.sv * si.dvdy);
  return mip_lookup_diff(images[t.image], st, dstdx, dstdy);
}
// checkerboard.rs:102-143: which of the two operands, or the box-filtered blend. Returns 0 = tex1, 1 = tex2, 2 = blend with area2.
RT_DEV int checker_select(const DTexture& t, const TexIn& si, float& area2) {
  f2 st = mk2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv);
  if (t.amount == 0) {  // AAMethod::None: `floor() as u32` saturates negatives to 0, the u32 sum wraps
    unsigned a = f2u_sat(floorf(st.x)), b = f2u_sat(floorf(st.y));
    return ((a + b) % 2u == 0u) ? 0 : 1;
  }
  f2 dstdx = mk2(t.su * si.dudx, t.sv * si.dvdx), dstdy = mk2(t.su * si.dudy, t.sv * si.dvdy);
  float ds = fmaxf(fabsf(dstdx.x), fabsf(dstdy.x)), dt = fmaxf(fabsf(dstdx.y), fabsf(dstdy.y));
  float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
  if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1)) {
    int sum = (int)((long long)f2i_sat(floorf(st.x)) + (long long)f2i_sat(floorf(st.y)));
    return (sum % 2 == 0) ? 0 : 1;
  }
  float b1 = floorf(s1 / 2.0f) + 2.0f * fmaxf(s1 / 2.0f - floorf(s1 / 2.0f) - 0.5f, 0.0f);
  float b0 = floorf(s0 / 2.0f) + 2.0f * fmaxf(s0 / 2.0f - floorf(s0 / 2.0f) - 0.5f, 0.0f);
  float c1 = floorf(t1 / 2.0f) + 2.0f * fmaxf(t1 / 2.0f - floorf(t1 / 2.0f) - 0.5f, 0.0f);
  float c0 = floorf(t0 / 2.0f) + 2.0f * fmaxf(t0 / 2.0f - floorf(t0 / 2.0f) - 0.5f, 0.0f);
  float sint = (b1 - b0) / (2.0f * ds), tint = (c1 - c0) / (2.0f * dt);
  area2 = sint + tint - 2.0f * sint * tint;
  if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
  return 2;
}
RT_DEV rgb3 tex_combine(const DTexture* textures, const DImage* images, const DTexture& t, rgb3 a, rgb3 b, const TexIn& si) {
  if (t.kind == 1) return a * b;  // scale.rs:23-25
  if (t.kind == 2) { float amt = tex_leaf(images, textures[t.amount], si).r; return a * (1.0f - amt) + b * amt; }  // mix.rs:24
  float area2 = 0.0f;
  const int sel = checker_select(t, si, area2);
  return sel == 0 ? a : (sel == 1 ? b : a * (1.0f - area2) + b * area2);
}
// a texture whose operands are leaves
RT_DEV rgb3 tex_depth1(const DTexture* textures, const DImage* images, const DTexture& t, const TexIn& si) {
  if (tex_is_leaf(t.kind)) return tex_leaf(images, t, si);
  return tex_combine(textures, images, t, tex_leaf(images, textures[t.tex1], si), tex_leaf(images, textures[t.tex2], si), si);
}
// combinators nest two deep at most (a combinator of combinators of leaves); the host rejects deeper scenes
// The register budget of a call's aggregate arguments is 16 dwords, pointers included; scalars always travel in registers - hence the nine floats.
RT_DEVN rgb3 tex_eval_q(const DTexture* textures, const DImage* images, int id, f2 uv, float dudx, float dvdx, float dudy, float dvdy, float px, float py, float pz,
                        float dpdx_x, float dpdx_y, float dpdx_z, float dpdy_x, float dpdy_y, float dpdy_z) {
  TexIn si; si.uv = uv; si.dudx = dudx; si.dvdx = dvdx; si.dudy = dudy; si.dvdy = dvdy; si.p = mk3(px, py, pz); si.dpdx = mk3(dpdx_x, dpdx_y, dpdx_z); si.dpdy = mk3(dpdy_x, dpdy_y, dpdy_z);
  const DTexture& t = textures[id];
  if (tex_is_leaf(t.kind)) return tex_leaf(images, t, si);
  return tex_combine(textures, images, t, tex_depth1(textures, images, textures[t.tex1], si), tex_depth1(textures, images, textures[t.tex2], si), si);
}
RT_DEV rgb3 tex_eval(const DScene& sc, int id, const SurfaceInteraction& si) {
  return tex_eval_q(sc.textures, sc.images, id, si.uv, si.dudx, si.dvdx, si.dudy, si.dvdy, si.hit.p.x, si.hit.p.y, si.hit.p.z, si.dpdx.x, si.dpdx.y, si.dpdx.z, si.dpdy.x, si.dpdy.y, si.dpdy.z);
}
RT_DEV float tex_eval_f(const DScene& sc, int id, const SurfaceInteraction& si) { return tex_eval(sc, id, si).r; }
// The image-map leaf alone (imagemap.rs:232-235), out of line: what the register-resident shade front-ends call. Their material classes hold constant and
// image-map parameters only (rt_scene_create sends every other texture shape to the generic kernel), so the general evaluator above - combinators two deep,
// four inlined MIP lookups, Perlin noise: 163 VGPRs - is not reachable from them and does not set their register allocation.
RT_DEVN rgb3 tex_image_q(const DTexture* textures, const DImage* images, int id, f2 uv, float dudx, float dvdx, float dudy, float dvdy) {
  const DTexture& t = textures[id];
  f2 st = mk2(t.su * uv.x + t.du, t.sv * uv.y + t.dv);  // UVMapping2D (texture/mod.rs:52-60)
  f2 dstdx = mk2(t.su * dudx, t.sv * dvdx), dstdy = mk2(t.su * dudy, t.sv * dvdy);
  return mip_lookup_diff(images[t.image], st, dstdx, dstdy);
}
RT_DEV rgb3 tex_eval_leaf(const DScene& sc, int id, const SurfaceInteraction& si) {  // constant or image map
  const DTexture& t = sc.textures[id];
  return t.kind == 0 ? mkc(t.v[0], t.v[1], t.v[2]) : tex_image_q(sc.textures, sc.images, id, si.uv, si.dudx, si.dvdx, si.dudy, si.dvdy);
}
// A vertex past the camera ray has no differentials (interaction.rs:245-314: only camera rays carry them), and an image map under zero differentials is
// `triangle(0, st)` whichever filter it was built for (mipmap.rs:227-283: EWA with a zero minor axis, trilinear with width 0): the level-0 bilinear lookup, inline
RT_DEV rgb3 tex_eval_leaf_bounced(const DScene& sc, int id, const SurfaceInteraction& si) {
  const DTexture& t = sc.textures[id];
  if (t.kind == 0) return mkc(t.v[0], t.v[1], t.v[2]);
  return mip_triangle(sc.images[t.image], 0, mk2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv));  // UVMapping2D (texture/mod.rs:52-60), then what mip_lookup_diff returns for zero derivatives
}
// the same values with the constant texture (the common parameter) answered in place instead of through the out-of-line evaluator
RT_DEV rgb3 tex_eval_c(const DScene& sc, int id, const SurfaceInteraction& si) { return tex_eval_leaf(sc, id, si); }  // (front-end classes: constants and image maps)
RT_DEV float tex_eval_cf(const DScene& sc, int id, const SurfaceInteraction& si) { return tex_eval_c(sc, id, si).r; }

// ---------------------------------------------------------------- alpha masks (rc/shapes/mesh.rs:353-370, 534-582)
// An accepted hit test is no hit when the mesh's "alpha" float texture evaluates to 0 at the hit (and, for intersect_p, its "shadowalpha"
// texture). The reference evaluates the texture on a local SurfaceInteraction(p_hit, uv_hit, -ray.d, dpdu, dpdv) without ray differentials
// (dudx .. = 0, dpdx = dpdy = 0: an image map then takes its bilinear level-0 lookup). No texture class reads dpdu / dpdv or the normal, so the
// reference's `/ inv_det` in intersect_p's dpdu (mesh.rs:549-550, where intersect multiplies) cannot change an outcome; they are not formed here.
RT_DEVN bool tri_alpha_rejects(const DScene& sc, int prim, const TriHit& h, bool shadow_ray) {
  const unsigned flags = tri_flags(sc.tri_p, prim);
  const unsigned want = shadow_ray ? 48u : 16u;
  if (!(flags & want)) return false;
  f3 p0, p1, p2; load_tri(sc.tri_p, prim, p0, p1, p2);
  f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);  // mesh.rs:201-211
  if (flags & 4u) { const float* u = sc.tri_uv + 6 * (size_t)prim; uv0 = mk2(u[0], u[1]); uv1 = mk2(u[2], u[3]); uv2 = mk2(u[4], u[5]); }
  SurfaceInteraction si;
  si.hit.p = p0 * h.b0 + p1 * h.b1 + p2 * h.b2;
  si.uv = mk2(uv0.x * h.b0 + uv1.x * h.b1 + uv2.x * h.b2, uv0.y * h.b0 + uv1.y * h.b1 + uv2.y * h.b2);
  si.dudx = si.dvdx = si.dudy = si.dvdy = 0.0f; si.dpdx = si.dpdy = mk3(0, 0, 0);
  si.hit.p_error = si.hit.wo = si.hit.n = si.dpdu = si.dpdv = si.sh_n = si.sh_dpdu = si.sh_dpdv = mk3(0, 0, 0); si.prim = prim;
  const int2 ids = sc.tri_alpha[prim];
  if ((flags & 16u) && tex_eval(sc, ids.x, si).r == 0.0f) return true;
  if (shadow_ray && (flags & 32u) && tex_eval(sc, ids.y, si).r == 0.0f) return true;
  return false;
}

// leaf_prim_test's handler (rtx_dev_scene.h): a quadric, or a triangle of a masked mesh
template <bool MASKS, bool QUADRICS>
RT_DEV bool general_prim_test(const DScene& sc_self, int prim, f3 p0, f3 p1, f3 p2, unsigned flags, const Ray& ray, const RayPre& rp, bool shadow_masks, TriHit& h) {
  if (QUADRICS && (flags & RT_FLAG_SPHERE)) {
    float ts;
    if (!(MASKS ? sphere_test(sc_self.spheres[__float_as_uint(p2.x)], ray.o, ray.d, ray.t_max, ts) : sphere_test_inl(sc_self.spheres[__float_as_uint(p2.x)], ray.o, ray.d, ray.t_max, ts))) return false;
    h.t = ts; h.b0 = h.b1 = 0.0f; h.b2 = ts;
    return true;
  }
  if (!tri_test_pre(p0, p1, p2, ray, rp, h)) return false;
  if (!MASKS) return true;  // (no mesh of the scene carries a mask: rt_scene_create picked this instantiation)
  return !tri_alpha_rejects(sc_self, prim, h, shadow_masks);
}

// ---------------------------------------------------------------- materials (rc/material/*.rs)
RT_DEV Lobe lobe_zero(int kind) {
  Lobe l; l.kind = kind; l.fr_kind = FR_NOOP; l.r = mkc(0, 0, 0); l.t = mkc(0, 0, 0); l.k = mkc(0, 0, 0);
  l.ax = l.ay = 0.0f; l.eta_a = l.eta_b = 1.0f; l.fr_ei = l.fr_et = 1.0f; l.n_scales = 0; l.scale0 = l.scale1 = mkc(1, 1, 1);
  return l;
}
RT_DEV void bsdf_add(Bsdf& b, const Lobe& l) { if (b.n < RT_MAX_LOBES) b.lobes[b.n++] = l; }
RT_DEV Lobe mk_micro_r(rgb3 r, float ax, float ay, int fr_kind, float ei, float et) {
  Lobe l = lobe_zero(LB_MICRO_R); l.r = r; l.ax = ax; l.ay = ay; l.fr_kind = fr_kind; l.fr_ei = ei; l.fr_et = et; return l;
}
RT_DEV Lobe mk_micro_t(rgb3 t, float ax, float ay, float ea, float eb) {
  Lobe l = lobe_zero(LB_MICRO_T); l.r = t; l.ax = ax; l.ay = ay; l.eta_a = ea; l.eta_b = eb; l.fr_kind = FR_DIELECTRIC; l.fr_ei = ea; l.fr_et = eb; return l;
}
RT_DEV Lobe mk_spec_t(rgb3 t, float ea, float eb) {
  Lobe l = lobe_zero(LB_SPEC_T); l.r = t; l.eta_a = ea; l.eta_b = eb; l.fr_kind = FR_DIELECTRIC; l.fr_ei = ea; l.fr_et = eb; return l;
}
RT_DEV Lobe mk_lambert(int kind, rgb3 r) { Lobe l = lobe_zero(kind); l.r = r; return l; }

// Fills the lobes of one non-mix material; returns eta.
RT_DEVN float material_lobes(const DScene& sc, const DMaterial& m, const SurfaceInteraction& si, Bsdf& b) {
  float eta = 1.0f;
  const int* s = m.slot;
  switch (m.kind) {
    case 0: {  // matte.rs:37-62
      rgb3 r = clamp_pos(tex_eval(sc, s[0], si));
      float sigma = clampf(tex_eval_f(sc, s[4], si), 0.0f, 1.0f);
      if (!is_black(r)) {
        if (sigma == 0.0f) bsdf_add(b, mk_lambert(LB_LAMBERT_R, r));
        else {  // OrenNayar::new, oren_nayar.rs:17-27
          Lobe l = lobe_zero(LB_OREN_NAYAR); l.r = r;
          float sigma_rad = sigma * (kPi / 180.0f);
          float sigma2 = sigma_rad * sigma_rad;
          l.ax = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
          l.ay = 0.45f * sigma2 / (sigma2 + 0.09f);
          bsdf_add(b, l);
        }
      }
      break;
    }
    case 1: {  // plastic.rs:45-75
      rgb3 kd = tex_eval(sc, s[0], si), ks = tex_eval(sc, s[1], si);
      if (!is_black(kd)) bsdf_add(b, mk_lambert(LB_LAMBERT_R, kd));
      if (!is_black(ks)) {
        float rough = tex_eval_f(sc, s[5], si);
        if (m.remap) rough = tr_roughness_to_alpha(rough);
        bsdf_add(b, mk_micro_r(ks, rough, rough, FR_DIELECTRIC, 1.5f, 1.0f));
      }
      break;
    }
    case 2: {  // metal.rs:50-82
      float ur = tex_eval_f(sc, s[6] >= 0 ? s[6] : s[5], si), vr = tex_eval_f(sc, s[7] >= 0 ? s[7] : s[5], si);
      if (m.remap) { ur = tr_roughness_to_alpha(ur); vr = tr_roughness_to_alpha(vr); }
      Lobe l = mk_micro_r(mkc(1, 1, 1), ur, vr, FR_CONDUCTOR, 1.0f, 1.0f);
      l.t = tex_eval(sc, s[8], si); l.k = tex_eval(sc, s[9], si);
      bsdf_add(b, l);
      break;
    }
    case 3: {  // mirror.rs:30-48
      rgb3 R = clamp_pos(tex_eval(sc, s[2], si));
      if (!is_black(R)) { Lobe l = lobe_zero(LB_SPEC_R); l.r = R; bsdf_add(b, l); }
      break;
    }
    case 4: {  // glass.rs:53-106, allow_multiple_lobes = true (path.rs:145)
      eta = tex_eval_f(sc, s[8], si);
      float ur = tex_eval_f(sc, s[6], si), vr = tex_eval_f(sc, s[7], si);
      rgb3 r = tex_eval(sc, s[2], si), t = tex_eval(sc, s[3], si);
      if (!is_black(r) || !is_black(t)) {
        if (ur == 0.0f && vr == 0.0f) {
          Lobe l = lobe_zero(LB_FRESNEL_SPEC); l.r = r; l.t = t; l.eta_a = 1.0f; l.eta_b = eta; bsdf_add(b, l);
        } else {
          if (m.remap) { ur = tr_roughness_to_alpha(ur); vr = tr_roughness_to_alpha(vr); }
          if (!is_black(r)) bsdf_add(b, mk_micro_r(r, ur, vr, FR_DIELECTRIC, 1.0f, eta));
          if (!is_black(t)) bsdf_add(b, mk_micro_t(r, ur, vr, 1.0f, eta));  // passes `r` (glass.rs:97)
        }
      }
      break;
    }
    case 5: {  // uber.rs:63-126
      float e = tex_eval_f(sc, s[8], si);
      rgb3 op = clamp_pos(tex_eval(sc, s[10], si));
      rgb3 t = clamp_pos(mkc(1, 1, 1) - op);
      eta = e;
      if (!is_black(t)) { eta = 1.0f; bsdf_add(b, mk_spec_t(t, 1.0f, 1.0f)); }
      rgb3 kd = op * clamp_pos(tex_eval(sc, s[0], si));
      if (!is_black(kd)) bsdf_add(b, mk_lambert(LB_LAMBERT_R, kd));
      rgb3 ks = op * clamp_pos(tex_eval(sc, s[1], si));
      if (!is_black(ks)) {
        float ru = tex_eval_f(sc, s[6] >= 0 ? s[6] : s[5], si), rv = tex_eval_f(sc, s[7] >= 0 ? s[7] : s[5], si);
        if (m.remap) { ru = tr_roughness_to_alpha(ru); rv = tr_roughness_to_alpha(rv); }
        bsdf_add(b, mk_micro_r(ks, ru, rv, FR_DIELECTRIC, 1.0f, e));
      }
      rgb3 kr = op * clamp_pos(tex_eval(sc, s[2], si));
      if (!is_black(kr)) { Lobe l = lobe_zero(LB_SPEC_R); l.r = kr; l.fr_kind = FR_DIELECTRIC; l.fr_ei = 1.0f; l.fr_et = e; bsdf_add(b, l); }
      rgb3 kt = op * clamp_pos(tex_eval(sc, s[3], si));
      if (!is_black(kt)) bsdf_add(b, mk_spec_t(kt, 1.0f, e));
      break;
    }
    case 6: {  // substrate.rs:43-71
      rgb3 d = clamp_pos(tex_eval(sc, s[0], si)), sp = clamp_pos(tex_eval(sc, s[1], si));
      float ru = tex_eval_f(sc, s[6], si), rv = tex_eval_f(sc, s[7], si);
      if (!is_black(d) || !is_black(sp)) {
        if (m.remap) { ru = tr_roughness_to_alpha(ru); rv = tr_roughness_to_alpha(rv); }
        Lobe l = lobe_zero(LB_FRESNEL_BLEND); l.r = d; l.t = sp; l.ax = ru; l.ay = rv; bsdf_add(b, l);
      }
      break;
    }
    case 8: {  // translucent.rs:49-101
      eta = 1.5f;
      rgb3 r = clamp_pos(tex_eval(sc, s[11], si)), t = clamp_pos(tex_eval(sc, s[12], si));
      if (!is_black(r) || !is_black(t)) {
        rgb3 kd = clamp_pos(tex_eval(sc, s[0], si));
        if (!is_black(kd)) {
          if (!is_black(r)) bsdf_add(b, mk_lambert(LB_LAMBERT_R, r * kd));
          if (!is_black(t)) bsdf_add(b, mk_lambert(LB_LAMBERT_T, t * kd));
        }
        rgb3 ks = clamp_pos(tex_eval(sc, s[1], si));
        if (!is_black(ks) && (!is_black(r) || !is_black(t))) {
          float rough = tex_eval_f(sc, s[5], si);
          if (m.remap) rough = tr_roughness_to_alpha(rough);
          if (!is_black(r)) bsdf_add(b, mk_micro_r(r * ks, rough, rough, FR_DIELECTRIC, 1.0f, eta));
          if (!is_black(t)) bsdf_add(b, mk_micro_t(t * ks, rough, rough, 1.0f, eta));
        }
      }
      break;
    }
    case 9: {  // disney.rs:82-213. Slots: 0 color, 1 metallic, 8 eta, 5 roughness, 2 speculartint, 6 anisotropic, 3 sheen, 4 sheentint,
               // 7 clearcoat, 9 clearcoatgloss, 10 spectrans, 11 scatterdistance, 12 flatness, 13 difftrans, 14 thin (0 / 1)
      const bool thin = s[14] != 0;
      rgb3 c = clamp_pos(tex_eval(sc, s[0], si));
      float metallic_weight = tex_eval_f(sc, s[1], si);
      float e = tex_eval_f(sc, s[8], si);
      float strans = tex_eval_f(sc, s[10], si);
      float diffuse_weight = (1.0f - metallic_weight) * (1.0f - strans);
      float dt = tex_eval_f(sc, s[13], si) / 2.0f;
      float rough = tex_eval_f(sc, s[5], si);
      float lum = lum_y(c);
      rgb3 c_tint = lum > 0.0f ? c / lum : mkc(1, 1, 1);
      float sheen_weight = tex_eval_f(sc, s[3], si);
      rgb3 c_sheen = mkc(0, 0, 0);
      if (sheen_weight > 0.0f) { float stint = tex_eval_f(sc, s[4], si); c_sheen = mkc(1, 1, 1) * (1.0f - stint) + c_tint * stint; }
      if (diffuse_weight > 0.0f) {
        if (thin) {
          float flat = tex_eval_f(sc, s[12], si);
          bsdf_add(b, mk_lambert(LB_DISNEY_DIFFUSE, diffuse_weight * (1.0f - flat) * (1.0f - dt) * c));
          { Lobe l = mk_lambert(LB_DISNEY_FAKESS, diffuse_weight * flat * (1.0f - dt) * c); l.ax = rough; bsdf_add(b, l); }
        } else {
          rgb3 sd = tex_eval(sc, s[11], si);
          if (is_black(sd)) bsdf_add(b, mk_lambert(LB_DISNEY_DIFFUSE, diffuse_weight * c));
          else bsdf_add(b, mk_spec_t(mkc(1, 1, 1), 1.0f, e));  // stand-in of the missing BSSRDF (:137-144)
        }
        { Lobe l = mk_lambert(LB_DISNEY_RETRO, diffuse_weight * c); l.ax = rough; bsdf_add(b, l); }
        if (sheen_weight > 0.0f) bsdf_add(b, mk_lambert(LB_DISNEY_SHEEN, diffuse_weight * sheen_weight * c_sheen));
      }
      float aspect = sqrtf(1.0f - tex_eval_f(sc, s[6], si) * 0.9f);
      float ax = fmaxf(0.001f, (rough * rough) / aspect), ay = fmaxf(0.001f, (rough * rough) * aspect);
      float spec_tint = tex_eval_f(sc, s[2], si);
      float r0eta = ((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f));  // schlick_r0_from_eta, :497-503
      rgb3 tinted = r0eta * (mkc(1, 1, 1) * (1.0f - spec_tint) + c_tint * spec_tint);
      rgb3 cspec0 = tinted * (1.0f - metallic_weight) + c * metallic_weight;
      { Lobe l = mk_micro_r(c, ax, ay, FR_DISNEY | FR_SEPARABLE_G, metallic_weight, e); l.t = cspec0; bsdf_add(b, l); }
      float cc = tex_eval_f(sc, s[7], si);
      if (cc > 0.0f) { Lobe l = lobe_zero(LB_DISNEY_CLEARCOAT); l.ax = cc; l.ay = lerpf(tex_eval_f(sc, s[9], si), 0.1f, 0.001f); bsdf_add(b, l); }
      if (strans > 0.0f) {
        rgb3 t = strans * sqrt3(c);
        if (thin) {
          float rscaled = (0.65f * e - 0.35f) * rough;
          bsdf_add(b, mk_micro_t(t, fmaxf(0.001f, (rscaled * rscaled) / aspect), fmaxf(0.001f, (rscaled * rscaled) * aspect), 1.0f, e));
        } else { Lobe l = mk_micro_t(t, ax, ay, 1.0f, e); l.fr_kind |= FR_SEPARABLE_G; bsdf_add(b, l); }
      }
      if (thin) bsdf_add(b, mk_lambert(LB_LAMBERT_T, dt * c));
      eta = 1.0f;  // Bsdf::new(si, 1.0, ..), :212
      break;
    }
    default: break;
  }
  return eta;
}
RT_DEV void scale_lobes(Bsdf& b, int first, int last, rgb3 s) {
  for (int i = first; i < last; ++i) {
    Lobe& l = b.lobes[i];
    if (l.n_scales == 0) l.scale0 = s; else l.scale1 = s;
    l.n_scales += 1;
  }
}
// Material::compute_scattering_functions. MixMaterial (mixmat.rs:34-64) is expanded up to two
// levels (a mix whose operands are plain materials or mixes of plain materials); the host rejects
// deeper nesting at scene creation.
// material::bump (rc/material/mod.rs:50-92). Triangles carry dndu = dndv = 0 (mesh.rs:372-382 passes zero()) and so do their
// shading.dndu / dndv, which keeps the terms below in place with zero vectors.
RT_DEVN void bump_map(const DScene& sc, int tex, SurfaceInteraction& si) {
  const f3 dndu = mk3(0, 0, 0), dndv = mk3(0, 0, 0);
  SurfaceInteraction e = si;
  float du = 0.5f * (fabsf(si.dudx) + fabsf(si.dudy));
  if (du == 0.0f) du = 0.0005f;
  e.hit.p = si.hit.p + du * si.sh_dpdu;
  e.uv = mk2(si.uv.x + du, si.uv.y + 0.0f);
  e.hit.n = normalize(cross(si.sh_dpdu, si.sh_dpdv) + du * dndu);
  const float u_displace = tex_eval(sc, tex, e).r;
  float dv = 0.5f * (fabsf(si.dvdx) + fabsf(si.dvdy));
  if (dv == 0.0f) dv = 0.0005f;
  e.hit.p = si.hit.p + dv * si.sh_dpdv;
  e.uv = mk2(si.uv.x + 0.0f, si.uv.y + dv);
  e.hit.n = normalize(cross(si.sh_dpdu, si.sh_dpdv) + dv * dndv);
  const float v_displace = tex_eval(sc, tex, e).r;
  const float displace = tex_eval(sc, tex, si).r;
  const f3 dpdu = si.sh_dpdu + (u_displace - displace) / du * si.sh_n + displace * dndu;
  const f3 dpdv = si.sh_dpdv + (v_displace - displace) / dv * si.sh_n + displace * dndv;
  // set_shading_geometry(dpdu, dpdv, dndu, dndv, false), interaction.rs:218-242
  f3 n = normalize(cross(dpdu, dpdv));
  if (tri_flags(sc.tri_p, si.prim) & 1u) n = n * -1.0f;  // reverse_orientation ^ transform_swaps_handedness
  si.sh_n = face_forward(n, si.hit.n);
  si.sh_dpdu = dpdu; si.sh_dpdv = dpdv;
}
// one non-mix material: its bump map first (the first statement of its compute_scattering_functions), then its lobes
RT_DEV float material_apply(const DScene& sc, const DMaterial& m, SurfaceInteraction& si, Bsdf& b) {
  if (m.bump >= 0) bump_map(sc, m.bump, si);
  return material_lobes(sc, m, si, b);
}
RT_DEVN void build_bsdf(const DScene& sc, int mat, SurfaceInteraction& si, Bsdf& b) {
  b.n = 0;
  const DMaterial& m = sc.materials[mat];
  float eta;
  if (m.kind != 7) eta = material_apply(sc, m, si, b);
  else {
    rgb3 s1 = clamp_pos(tex_eval(sc, m.slot[13], si)), s2 = clamp_pos(mkc(1, 1, 1) - s1);
    eta = 1.0f;
    SurfaceInteraction si2 = si;  // mat2 works on a clone taken before mat1 touches si (mixmat.rs:43); mat1's changes stay in si
    for (int side = 0; side < 2; ++side) {
      const DMaterial& c = sc.materials[side == 0 ? m.slot[14] : m.slot[15]];
      SurfaceInteraction& cs = side == 0 ? si : si2;
      int first = b.n;
      float e;
      if (c.kind != 7) e = material_apply(sc, c, cs, b);
      else {
        rgb3 c1 = clamp_pos(tex_eval(sc, c.slot[13], cs)), c2 = clamp_pos(mkc(1, 1, 1) - c1);
        SurfaceInteraction cs2 = cs;
        int f1 = b.n;
        e = material_apply(sc, sc.materials[c.slot[14]], cs, b);
        scale_lobes(b, f1, b.n, c1);
        int f2_ = b.n;
        (void)material_apply(sc, sc.materials[c.slot[15]], cs2, b);
        scale_lobes(b, f2_, b.n, c2);
      }
      scale_lobes(b, first, b.n, side == 0 ? s1 : s2);
      if (side == 0) eta = e;  // the Bsdf (frame, eta) of mat1 is kept, only its lobe list is replaced
    }
  }
  b.eta = eta;
  bsdf_init_frame(b, si.sh_n, si.hit.n, si.sh_dpdu);
}

// ---------------------------------------------------------------- distributions
RT_DEV void d1_sample_continuous(const float* func, const float* cdf, float func_int, int n, float u, float& x, float& pdf, int& off) {  // distribution1d.rs:48-68
  int offset = find_interval_le(cdf, n + 1, u);
  float du = u - cdf[offset];
  if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
  pdf = func_int > 0.0f ? vdiv(func[offset], func_int) : 0.0f;
  x = ((float)offset + du) / (float)n;
  off = offset;
}
template <bool EXACT = false>
RT_DEV void d1_sample_continuous_guided(const float* func, const float* cdf, float func_int, int n, float u, const unsigned short* guide, int glog, float& x, float& pdf, int& off) {
  const int G = 1 << glog;
  const int k = clampi((int)(u * (float)G), 0, G - 1);  // u 2^glog is exact
  const int g0 = (int)guide[k], g1 = (int)guide[k + 1];
  int offset = find_interval_le_from(cdf, n + 1, u, g0, g1 - g0);
  float du = u - cdf[offset];
  if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
  pdf = func_int > 0.0f ? vdiv_e<EXACT>(func[offset], func_int) : 0.0f;
  x = ((float)offset + du) / (float)n;
  off = offset;
}
// ... over a row of (cdf, func) pairs
template <bool EXACT = false>
RT_DEV void d1_sample_continuous_pairs(const float* cf, float func_int, int n, float u, const unsigned short* guide, int glog, float& x, float& pdf, int& off) {
  const int G = 1 << glog;
  const int k = clampi((int)(u * (float)G), 0, G - 1);
  const int g0 = (int)guide[k], g1 = (int)guide[k + 1];
  const int offset = find_interval_le_from_pairs(cf, n + 1, u, g0, g1 - g0);
  const float c0 = cf[2 * offset], f0 = cf[2 * offset + 1], c1 = cf[2 * offset + 2];
  float du = u - c0;
  if (c1 - c0 > 0.0f) du /= c1 - c0;
  pdf = func_int > 0.0f ? vdiv_e<EXACT>(f0, func_int) : 0.0f;
  x = ((float)offset + du) / (float)n;
  off = offset;
}
// (Round 6 measured the row's bracket as ONE 32-byte bucket record per (row, guide bucket) - {first index | count, four cdf entries, three func entries}: everything
// sample_continuous reads of the row in one gather instead of a guide entry and then a bracket of pairs. Exact, and slower: S4 shade 2331 -> 2371 ms. The 4 MB of guide entries
// mostly hit L2 and the 17 MB of pairs the Infinity Cache; 64 MB of records do neither better, and the stage is bound by its L2-miss REQUESTS, of which the record saves none.
// Commit ec235f5; MEASUREMENTS R6.)
RT_DEV void d1_sample_discrete_guided(const float* func, const float* cdf, float func_int, int n, float u, const unsigned short* guide, int glog, int& off, float& pdf) {
  const int G = 1 << glog;
  const int k = clampi((int)(u * (float)G), 0, G - 1);
  const int g0 = (int)guide[k], g1 = (int)guide[k + 1];
  int offset = find_interval_le_from(cdf, n + 1, u, g0, g1 - g0);
  pdf = func_int > 0.0f ? vdiv(func[offset], func_int * (float)n) : 0.0f;
  off = offset;
}
// Distribution1D::sample_discrete (distribution1d.rs:70-79) of a distribution of n <= 3 entries held in registers: r0 = {func_int, func[0..2]}, r1 = {cdf[0..3]}
// (DScene::ld_rows8). find_interval over a non-decreasing array is (the number of entries <= u) - 1, clamped: the same index the bisection returns; same quotient.
RT_DEV void d1_sample_discrete_row8(float4 r0, float4 r1, int n, float u, int& off, float& pdf) {
  const int cnt = (r1.x <= u ? 1 : 0) + ((n >= 1 && r1.y <= u) ? 1 : 0) + ((n >= 2 && r1.z <= u) ? 1 : 0) + ((n >= 3 && r1.w <= u) ? 1 : 0);
  const int offset = clampi(cnt - 1, 0, n - 1);
  const float f = offset == 0 ? r0.y : (offset == 1 ? r0.z : r0.w);
  pdf = r0.x > 0.0f ? vdiv(f, r0.x * (float)n) : 0.0f;
  off = offset;
}
RT_DEV void d1_sample_discrete(const float* func, const float* cdf, float func_int, int n, float u, int& off, float& pdf) {  // :70-79
  int offset = find_interval_le(cdf, n + 1, u);
  pdf = func_int > 0.0f ? vdiv(func[offset], func_int * (float)n) : 0.0f;
  off = offset;
}

// ---------------------------------------------------------------- lights
struct LiSample { rgb3 li; f3 wi; float pdf; LightPoint p1; };  // 16 dwords: returned in registers

RT_DEV float tri_area(const DScene& sc, int prim) {  // mesh.rs:588-594
  f3 p0, p1, p2; load_tri(sc.tri_p, prim, p0, p1, p2);
  return 0.5f * len(cross(p1 - p0, p2 - p0));
}
RT_DEV rgb3 area_light_l(const DLight& l, f3 n, f3 w) {  // diffuse.rs:91-97
  if (l.two_sided || dot(n, w) > 0.0f) return mkc(l.rgb[0], l.rgb[1], l.rgb[2]);
  return mkc(0, 0, 0);
}
RT_DEV f3 xf3x4(const float* m, f3 v) {  // Transform * Vector3f, transform.rs:288-303
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
RT_DEV rgb3 infinite_le_inl(const DImage* images, const DLight& l, f3 ray_d) {  // infinite.rs:211-219
  f3 w = normalize(xf3x4(l.w2l, ray_d));
  f2 st = mk2(spherical_phi(w) * kInvPi * 0.5f, spherical_theta(w) * kInvPi);
  return mip_lookup(images[l.image], st, 0.0f);
}
RT_DEVN rgb3 infinite_le_q(const DImage* images, const DLight& l, f3 ray_d) { return infinite_le_inl(images, l, ray_d); }
// INL: the evaluator inline (the BOUNCED form of k_shade<3>, whose register bound then covers it); otherwise the out-of-line copy
template <bool INL = false>
RT_DEV rgb3 infinite_le(const DScene& sc, const DLight& l, f3 ray_d) { return INL ? infinite_le_inl(sc.images, l, ray_d) : infinite_le_q(sc.images, l, ray_d); }
// DiffuseAreaLight::sample_li diffuse.rs:59-70 -> Shape::sample_si shapes/mod.rs:39-53 -> Triangle::sample mesh.rs:610-634
RT_DEV LiSample area_light_sample_li(const DScene& sc, const DLight& l, const Interaction& ref, f2 u) {
  LiSample s;
  f2 b = uniform_sample_triangle(u);
  f3 p0, p1, p2; load_tri(sc.tri_p, l.prim, p0, p1, p2);
  const unsigned flags = tri_flags(sc.tri_p, l.prim);
  float b2 = 1.0f - b.x - b.y;
  f3 p = (b.x * p0) + (b.y * p1) + (b2 * p2);
  f3 normal = mk3(l.nrm[0], l.nrm[1], l.nrm[2]);  // normalize(cross(p1 - p0, p2 - p0)), k_light_consts
  if (flags & 2u) {
    const float* q = sc.tri_n + 9 * (size_t)l.prim;
    f3 ns = b.x * mk3(q[0], q[1], q[2]) + b.y * mk3(q[3], q[4], q[5]) + b2 * mk3(q[6], q[7], q[8]);
    normal = face_forward(normal, ns);
  } else if (flags & 1u) normal = normal * -1.0f;
  f3 p_abs_sum = abs3(b.x * p0) + abs3(b.y * p1) + abs3(b2 * p2);
  s.p1.p = p; s.p1.p_error = gamma_n(6) * p_abs_sum; s.p1.n = normal;
  float pdf = l.inv_area;  // 1.0 / area
  f3 wi = p - ref.p;
  if (len2(wi) == 0.0f) pdf = 0.0f;
  else {
    wi = vnormalize(wi);  // the direction only enters BxDF values and this pdf; the shadow ray is spawned between the two points
    pdf *= vdiv(distance_squared(ref.p, p), fabsf(dot(normal, -wi)));
    if (isinf(pdf)) pdf = 0.0f;
  }
  s.wi = vnormalize(p - ref.p);
  s.pdf = pdf;
  s.li = area_light_l(l, normal, -s.wi);
  return s;
}
// Shape::pdf_wi (shapes/mod.rs:59-68): re-intersects the emitter triangle (Triangle::intersect, alpha mask included)
template <bool GENERAL>  // true only in the shade kernel of scenes with alpha-masked EMITTERS: the mask evaluator is a large out-of-line function
RT_DEV float area_light_pdf_li(const DScene& sc, const DLight& l, const Interaction& ref, f3 wi) {
  if (GENERAL && (tri_flags(sc.tri_p, l.prim) & RT_FLAG_SPHERE)) return sphere_pdf_wi(sc.spheres[prim_sphere_index(sc.tri_p, l.prim)], ref, wi);  // Sphere overrides pdf_wi
  Ray ray = spawn_ray(ref, wi);
  f3 p0, p1, p2; load_tri_rec(sc.tri_rec, l.prim, p0, p1, p2);  // the line tri_hit_point_normal_inl reads as well
  TriHit h;
  if (!tri_test(p0, p1, p2, ray, h)) return 0.0f;
  if (GENERAL && sc.tri_alpha != nullptr && tri_alpha_rejects(sc, l.prim, h, false)) return 0.0f;
  f3 p, n; tri_hit_point_normal_inl(sc, l.prim, h, p, n);
  return vdiv(distance_squared(ref.p, p), fabsf(dot(n, -wi)) * l.area);
}
template <bool GENERAL, bool EXACT = false>  // EXACT: the light-distribution build (k_lightdist_contrib), whose tables are bit-exact: correctly rounded quotients
RT_DEV LiSample light_sample_li_inl(const DScene& sc, const DLight& l, Interaction ref, f2 u) {  // sc: a DScene in device memory (*sc.self); ref by value, in registers
  LiSample s;
  switch (l.kind) {
    case 0:
      if (GENERAL && (tri_flags(sc.tri_p, l.prim) & RT_FLAG_SPHERE)) {  // DiffuseAreaLight::sample_li (diffuse.rs:59-70) over Sphere::sample_si
        float pdf; const SpherePoint sp = sphere_sample_si(sc.spheres[prim_sphere_index(sc.tri_p, l.prim)], ref, u, pdf);
        s.p1.p = sp.p; s.p1.p_error = sp.p_error; s.p1.n = sp.n;
        s.wi = normalize(sp.p - ref.p); s.pdf = pdf; s.li = area_light_l(l, sp.n, -s.wi);
        return s;
      }
      return area_light_sample_li(sc, l, ref, u);
    case 1: {  // PointLight::sample_li point.rs:43-54 (I / (4 pi r^2), reference quirk)
      f3 pos = mk3(l.vec[0], l.vec[1], l.vec[2]);
      f3 wi = pos - ref.p;
      float r2 = len2(wi);
      s.li = mkc(l.rgb[0], l.rgb[1], l.rgb[2]) / (4.0f * kPi * r2);
      s.wi = normalize(wi); s.pdf = 1.0f;
      s.p1.p = pos; s.p1.p_error = mk3(0, 0, 0); s.p1.n = mk3(0, 0, 0);
      return s;
    }
    case 2: {  // DistantLight::sample_li distant.rs:57-70
      f3 dir = mk3(l.vec[0], l.vec[1], l.vec[2]);
      s.li = mkc(l.rgb[0], l.rgb[1], l.rgb[2]); s.wi = dir; s.pdf = 1.0f;
      s.p1.p = ref.p + dir * (2.0f * l.world_radius); s.p1.p_error = mk3(0, 0, 0); s.p1.n = mk3(0, 0, 0);
      return s;
    }
    default: {  // InfiniteAreaLight::sample_li infinite.rs:143-181
      float d1, pdf1, d0, pdf0; int v, dummy;
      if (RT_DBG(sc, 2)) { d1 = u.y; d0 = u.x; pdf0 = pdf1 = 1.0f; } else {
      d1_sample_continuous_guided<EXACT>(l.mfunc, l.mcdf, l.mfunc_int, l.nv, u.y, l.mguide, l.mglog, d1, pdf1, v);  // Distribution2D::sample_continuous
      d1_sample_continuous_pairs<EXACT>(l.cf + (size_t)v * (l.nu + 1) * 2, l.func_int[v], l.nu, u.x, l.guide + ((size_t)v << l.glog) + (size_t)v, l.glog, d0, pdf0, dummy);
      }
      float map_pdf = pdf0 * pdf1;
      s.p1.p_error = mk3(0, 0, 0); s.p1.n = mk3(0, 0, 0);
      if (map_pdf == 0.0f) { s.li = mkc(0, 0, 0); s.wi = mk3(0, 0, 0); s.pdf = 0.0f; s.p1.p = mk3(0, 0, 0); return s; }
      float theta = d1 * kPi, phi = d0 * 2.0f * kPi;
      float cos_theta_, sin_theta_, cos_phi_, sin_phi_; sincosf(theta, &sin_theta_, &cos_theta_); sincosf(phi, &sin_phi_, &cos_phi_);
      f3 wi = xf3x4(l.l2w, mk3(sin_theta_ * cos_phi_, sin_theta_ * sin_phi_, cos_theta_));
      s.pdf = sin_theta_ == 0.0f ? 0.0f : vdiv_e<EXACT>(map_pdf, 2.0f * kPi * kPi * sin_theta_);
      s.p1.p = ref.p + wi * (2.0f * l.world_radius);
      // the radiance lookup `lmap.lookup(uv)` (infinite.rs:176) is left to the caller (light_sample_li_finish: a call of its own instead of 50 more registers
      // in this function): li carries the map coordinates until then
      s.li = mkc(d0, d1, 0.0f); s.wi = wi;
      return s;
    }
  }
}
// The environment map's radiance at map coordinates st (mipmap.rs:227-245 with width 0: the level-0 bilinear lookup), out of line
template <bool GENERAL, bool EXACT = false>
RT_DEVN LiSample light_sample_li(const DScene& sc, const DLight& l, Interaction ref, f2 u) { return light_sample_li_inl<GENERAL, EXACT>(sc, l, ref, u); }
RT_DEVN rgb3 infinite_li_q(const DImage* images, int image, float s0, float s1) { return mip_lookup(images[image], mk2(s0, s1), 0.0f); }
// completes a light sample: an infinite light's radiance is looked up here (light_sample_li returns the map coordinates in li)
template <bool GENERAL, bool EXACT = false, bool INL = false>
RT_DEV LiSample light_sample_li_full(const DScene& sc, const DLight& l, const Interaction& ref, f2 u) {
  LiSample s = INL ? light_sample_li_inl<GENERAL, EXACT>(sc, l, ref, u) : light_sample_li<GENERAL, EXACT>(sc, l, ref, u);
  if (l.kind == 3 && s.pdf != 0.0f) s.li = RT_DBG(sc, 4) ? mkc(1, 1, 1) : (INL ? mip_lookup(sc.images[l.image], mk2(s.li.r, s.li.g), 0.0f) : infinite_li_q(sc.images, l.image, s.li.r, s.li.g));
  return s;
}
// Light::pdf_li. Area lights: Shape::pdf_wi (shapes/mod.rs:59-68) re-intersects the emitter triangle.
template <bool GENERAL>
RT_DEV float light_pdf_li_inl(const DScene& sc, const DLight& l, const Interaction& ref, f3 wi) {
  if (l.kind == 0) return area_light_pdf_li<GENERAL>(sc, l, ref, wi);
  if (l.kind == 3) {  // infinite.rs:183-196
    f3 w = xf3x4(l.w2l, wi);
    float theta = spherical_theta(w), phi = spherical_phi(w);
    float sin_theta_ = sinf(theta);
    if (sin_theta_ == 0.0f) return 0.0f;
    f2 p = mk2(phi * kInvPi * 0.5f, theta * kInvPi);  // Distribution2D::pdf, distribution2d.rs:36-49
    int iu = clampi((int)f2u_sat(p.x * (float)l.nu), 0, l.nu - 1), iv = clampi((int)f2u_sat(p.y * (float)l.nv), 0, l.nv - 1);
    return vdiv(vdiv(l.cf[((size_t)iv * (l.nu + 1) + iu) * 2 + 1], l.mfunc_int), 2.0f * kPi * kPi * sin_theta_);
  }
  return 0.0f;
}
template <bool GENERAL>
RT_DEVN float light_pdf_li_q(const DScene& sc, const DLight& l, Interaction ref, float wi_x, float wi_y, float wi_z) { return light_pdf_li_inl<GENERAL>(sc, l, ref, mk3(wi_x, wi_y, wi_z)); }
template <bool GENERAL, bool INL = false>
RT_DEV float light_pdf_li(const DScene& sc, const DLight& l, const Interaction& ref, f3 wi) { return INL ? light_pdf_li_inl<GENERAL>(sc, l, ref, wi) : light_pdf_li_q<GENERAL>(sc, l, ref, wi.x, wi.y, wi.z); }
RT_DEV bool light_is_delta(const DLight& l) { return l.kind == 1 || l.kind == 2; }  // light/mod.rs:38-40

// ---------------------------------------------------------------- light distribution (rc/lightdistrib.rs)
RT_DEV f3 bounds_offset(f3 mn, f3 mx, f3 p) {  // bounds.rs:177-190
  f3 o = p - mn;
  if (mx.x > mn.x) o.x /= mx.x - mn.x;
  if (mx.y > mn.y) o.y /= mx.y - mn.y;
  if (mx.z > mn.z) o.z /= mx.z - mn.z;
  return o;
}
RT_DEV f3 bounds_lerp(f3 mn, f3 mx, f3 t) { return mk3(lerpf(t.x, mn.x, mx.x), lerpf(t.y, mn.y, mx.y), lerpf(t.z, mn.z, mx.z)); }  // :160-166
RT_DEV long voxel_of(const DScene& sc, f3 p) {  // lightdistrib.rs:187-198
  f3 offset = bounds_offset(sc.wb_min, sc.wb_max, p);
  int px = clampi(f2i_sat(offset.x * (float)sc.nvox[0]), 0, sc.nvox[0] - 1);
  int py = clampi(f2i_sat(offset.y * (float)sc.nvox[1]), 0, sc.nvox[1] - 1);
  int pz = clampi(f2i_sat(offset.z * (float)sc.nvox[2]), 0, sc.nvox[2] - 1);
  return ((long)pz * sc.nvox[1] + py) * sc.nvox[0] + px;
}

}  // namespace rtx
