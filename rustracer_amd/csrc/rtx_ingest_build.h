// rtx_ingest_build.h — texture pyramids and environment-map sampling tables built on the device.
// Included by rtx_hip.hip.
//
// MIPMap::new (rc/mipmap.rs:75-187): Lanczos zoom of a non-power-of-two image (along s, then along t with a clamp at zero), then
// one box-filtered level after another. InfiniteAreaLight::new (rc/light/infinite.rs:78-101): luminance * sin(theta) of a filtered
// lookup at twice the map's resolution, one Distribution1D per row and one over the row integrals. The host layer does the same
// on the CPU (rtx_host.cpp) and is what the oracle is compared with; these kernels perform the identical float operations in the
// identical order (no contraction, the transcendental parts - Lanczos weights, sin(theta), the log2 of the filter width - come
// from the host), so both builds give bit-identical tables. Device build: 1-2 ms for a 2048 x 1024 map instead of ~0.2 s.
#pragma once

namespace rtx {

RT_DEV long ingest_wrap(long i, long n, int wrap) {
  if (wrap == RT_WRAP_REPEAT) { long r = i % n; return r < 0 ? r + n : r; }
  if (wrap == RT_WRAP_CLAMP) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  return i;  // black: taps outside are skipped by the caller
}

// along s: out (px x h) from in (w x h); along t: out (px x py) from in (px x h rows), clamped at zero (mipmap.rs:100-139)
__global__ void k_mip_resample(const float* __restrict__ in, float* __restrict__ out, int in_w, int in_h, int out_w, int out_h, int along_t, int wrap,
                               const int* __restrict__ first, const float* __restrict__ wts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)out_w * out_h) return;
  const int x = (int)(i % out_w), y = (int)(i / out_w);
  const int o = along_t ? y : x, n = along_t ? in_h : in_w;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int j = 0; j < 4; ++j) {
    const long src = ingest_wrap((long)first[o] + j, n, wrap);
    if (src >= 0 && src < n) {
      const float* p = in + 3 * (along_t ? (size_t)src * in_w + x : (size_t)y * in_w + src);
      const float wj = wts[4 * o + j];
      for (int k = 0; k < 3; ++k) acc[k] += p[k] * wj;
    }
  }
  for (int k = 0; k < 3; ++k) out[3 * (size_t)i + k] = along_t ? (acc[k] < 0.0f ? 0.0f : acc[k]) : acc[k];
}

RT_DEV void ingest_texel(const float* __restrict__ lvl, int us, int vs, int wrap, long s, long t, float out[3]) {  // MIPMap::texel, mipmap.rs:208-225
  if (wrap == RT_WRAP_REPEAT) { s = ingest_wrap(s, us, wrap); t = ingest_wrap(t, vs, wrap); }
  else if (wrap == RT_WRAP_CLAMP) { s = ingest_wrap(s, us, wrap); t = ingest_wrap(t, vs, wrap); }
  else if (s < 0 || s >= us || t < 0 || t >= vs) { out[0] = out[1] = out[2] = 0.0f; return; }
  const float* p = lvl + 3 * ((size_t)t * us + s);
  out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}

__global__ void k_mip_downsample(const float* __restrict__ fine, float* __restrict__ coarse, int fw, int fh, int cw, int ch, int wrap) {  // mipmap.rs:170-184
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)cw * ch) return;
  const int sx = (int)(i % cw), t = (int)(i / cw);
  float a[3], b[3], c[3], d[3];
  ingest_texel(fine, fw, fh, wrap, 2 * sx, 2 * t, a); ingest_texel(fine, fw, fh, wrap, 2 * sx + 1, 2 * t, b);
  ingest_texel(fine, fw, fh, wrap, 2 * sx, 2 * t + 1, c); ingest_texel(fine, fw, fh, wrap, 2 * sx + 1, 2 * t + 1, d);
  for (int k = 0; k < 3; ++k) coarse[3 * (size_t)i + k] = (a[k] + b[k] + c[k] + d[k]) * 0.25f;
}

struct IngestPyramid { const float* texels; int n_levels; int w[RT_MAX_MIP_LEVELS], h[RT_MAX_MIP_LEVELS]; unsigned long long off[RT_MAX_MIP_LEVELS]; int wrap; };

RT_DEV void ingest_triangle(const IngestPyramid& m, int level, float sx, float sy, float out[3]) {  // MIPMap::triangle, mipmap.rs:285-308
  level = level < 0 ? 0 : (level > m.n_levels - 1 ? m.n_levels - 1 : level);
  const float s = sx * (float)m.w[level] - 0.5f, t = sy * (float)m.h[level] - 0.5f;
  const long s0 = (long)f2i_sat(floorf(s)), t0 = (long)f2i_sat(floorf(t));
  const float ds = s - (float)s0, dt = t - (float)t0;
  const float* lvl = m.texels + 3 * m.off[level];
  float a[3], b[3], c[3], d[3];
  ingest_texel(lvl, m.w[level], m.h[level], m.wrap, s0, t0, a); ingest_texel(lvl, m.w[level], m.h[level], m.wrap, s0, t0 + 1, b);
  ingest_texel(lvl, m.w[level], m.h[level], m.wrap, s0 + 1, t0, c); ingest_texel(lvl, m.w[level], m.h[level], m.wrap, s0 + 1, t0 + 1, d);
  for (int k = 0; k < 3; ++k) out[k] = a[k] * (1.0f - ds) * (1.0f - dt) + b[k] * (1.0f - ds) * dt + c[k] * ds * (1.0f - dt) + d[k] * ds * dt;
}

// img[v][u] = y(Lmap.lookup((u + .5) / width, (v + .5) / height, filter)) * sin(theta_v)   (infinite.rs:84-95)
// mode: 0 = level < 0 (triangle on level 0), 1 = level beyond the last (its single texel), 2 = blend of levels il and il + 1 by delta
__global__ void k_env_func(IngestPyramid m, int width, int height, int mode, int il, float delta, const float* __restrict__ sin_theta, float* __restrict__ img) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)width * height) return;
  const int u = (int)(i % width), v = (int)(i / width);
  const float up = ((float)u + 0.5f) / (float)width, vp = ((float)v + 0.5f) / (float)height;
  float c[3];
  if (mode == 0) ingest_triangle(m, 0, up, vp, c);
  else if (mode == 1) ingest_texel(m.texels + 3 * m.off[m.n_levels - 1], m.w[m.n_levels - 1], m.h[m.n_levels - 1], m.wrap, 0, 0, c);
  else {
    float a[3], b[3];
    ingest_triangle(m, il, up, vp, a); ingest_triangle(m, il + 1, up, vp, b);
    for (int k = 0; k < 3; ++k) c[k] = a[k] * (1.0f - delta) + b[k] * delta;
  }
  img[i] = (0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]) * sin_theta[v];
}

// Distribution1D::new (distribution1d.rs:11-42) of n_rows functions of n values each: the running sum is sequential by definition, one lane per row
__global__ void k_dist1_rows(const float* __restrict__ func, int n_rows, int n, float* __restrict__ cdf /* n_rows x (n + 1) */, float* __restrict__ func_int) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float* f = func + (size_t)r * n; float* c = cdf + (size_t)r * (n + 1);
  float run = 0.0f; c[0] = 0.0f;
  for (int i = 1; i < n + 1; ++i) { run = run + f[i - 1] / (float)n; c[i] = run; }
  func_int[r] = run;
  if (run == 0.0f) for (int i = 1; i < n + 1; ++i) c[i] = (float)i / (float)n;
  else for (int i = 1; i < n + 1; ++i) c[i] /= run;
}

}  // namespace rtx

// MIPMap::new on the device. rgb: width x height x 3 (host). For a non-power-of-two image the caller passes the zoomed size (px, py) and
// the Lanczos taps of both axes (first texel + 4 weights per output texel, host-computed), else px = width, py = height and NULL taps.
// lvl_w / lvl_h / lvl_off (texel offsets): the level geometry, n_levels entries; texels_out: room for all levels (host).
extern "C" int rt_mip_build(const float* rgb, int32_t width, int32_t height, int32_t px, int32_t py, const int32_t* s_first, const float* s_wts, const int32_t* t_first,
                            const float* t_wts, int32_t wrap, int32_t n_levels, const int32_t* lvl_w, const int32_t* lvl_h, const uint64_t* lvl_off, float* texels_out) {
  using namespace rtx;
  if (!rgb || !lvl_w || !lvl_h || !lvl_off || !texels_out || width <= 0 || height <= 0 || n_levels <= 0 || n_levels > RT_MAX_MIP_LEVELS) return fail(RT_ERR_INVALID, "bad pyramid arguments");
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  const bool zoom = px != width || py != height;
  if (zoom && (!s_first || !s_wts || !t_first || !t_wts)) return fail(RT_ERR_INVALID, "zoom taps missing");
  size_t total = 0; for (int i = 0; i < n_levels; ++i) total = std::max<size_t>(total, (size_t)lvl_off[i] + (size_t)lvl_w[i] * lvl_h[i]);
  DevBuf d_in, d_tmp, d_pyr, d_sf, d_sw, d_tf, d_tw;
  HIP_TRY(d_in.ensure((size_t)width * height * 12)); HIP_TRY(hipMemcpy(d_in.p, rgb, (size_t)width * height * 12, hipMemcpyHostToDevice));
  HIP_TRY(d_pyr.ensure(total * 12));
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  if (zoom) {
    HIP_TRY(d_tmp.ensure((size_t)px * height * 12));
    HIP_TRY(d_sf.ensure((size_t)px * 4)); HIP_TRY(d_sw.ensure((size_t)px * 16)); HIP_TRY(d_tf.ensure((size_t)py * 4)); HIP_TRY(d_tw.ensure((size_t)py * 16));
    HIP_TRY(hipMemcpy(d_sf.p, s_first, (size_t)px * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(d_sw.p, s_wts, (size_t)px * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tf.p, t_first, (size_t)py * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(d_tw.p, t_wts, (size_t)py * 16, hipMemcpyHostToDevice));
    k_mip_resample<<<blocks((size_t)px * height), 256, 0, 0>>>(d_in.as<float>(), d_tmp.as<float>(), width, height, px, height, 0, wrap, d_sf.as<int>(), d_sw.as<float>());
    k_mip_resample<<<blocks((size_t)px * py), 256, 0, 0>>>(d_tmp.as<float>(), d_pyr.as<float>(), px, height, px, py, 1, wrap, d_tf.as<int>(), d_tw.as<float>());
  } else HIP_TRY(hipMemcpy(d_pyr.p, d_in.p, (size_t)width * height * 12, hipMemcpyDeviceToDevice));
  for (int i = 1; i < n_levels; ++i)
    k_mip_downsample<<<blocks((size_t)lvl_w[i] * lvl_h[i]), 256, 0, 0>>>(d_pyr.as<float>() + 3 * lvl_off[i - 1], d_pyr.as<float>() + 3 * lvl_off[i], lvl_w[i - 1], lvl_h[i - 1], lvl_w[i], lvl_h[i], wrap);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(texels_out, d_pyr.p, total * 12, hipMemcpyDeviceToHost));
  return RT_OK;
}

// The sampling tables of InfiniteAreaLight::new on the device. image: the Lmap pyramid (host pointers); width x height = twice its resolution;
// mode / il / delta: MIPMap::lookup's level choice for the constant filter width (see k_env_func); sin_theta: height values (host-computed).
// Outputs (host): func height x width, cdf height x (width + 1), func_int height, marg_cdf height + 1, *marg_func_int; marg_func = func_int.
extern "C" int rt_env_distribution(const rt_image* image, int32_t width, int32_t height, int32_t mode, int32_t il, float delta, const float* sin_theta,
                                   float* func, float* cdf, float* func_int, float* marg_cdf, float* marg_func_int) {
  using namespace rtx;
  if (!image || !sin_theta || !func || !cdf || !func_int || !marg_cdf || !marg_func_int || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad distribution arguments");
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  DevBuf d_tex, d_sin, d_func, d_cdf, d_int, d_mcdf, d_mint;
  HIP_TRY(d_tex.ensure((size_t)image->n_texels * 12)); HIP_TRY(hipMemcpy(d_tex.p, image->texels, (size_t)image->n_texels * 12, hipMemcpyHostToDevice));
  HIP_TRY(d_sin.ensure((size_t)height * 4)); HIP_TRY(hipMemcpy(d_sin.p, sin_theta, (size_t)height * 4, hipMemcpyHostToDevice));
  HIP_TRY(d_func.ensure((size_t)width * height * 4)); HIP_TRY(d_cdf.ensure((size_t)height * (width + 1) * 4)); HIP_TRY(d_int.ensure((size_t)height * 4));
  HIP_TRY(d_mcdf.ensure((size_t)(height + 1) * 4)); HIP_TRY(d_mint.ensure(4));
  IngestPyramid m; m.texels = d_tex.as<float>(); m.n_levels = image->n_levels; m.wrap = image->wrap;
  for (int l = 0; l < RT_MAX_MIP_LEVELS; ++l) { m.w[l] = l < image->n_levels ? image->width[l] : 0; m.h[l] = l < image->n_levels ? image->height[l] : 0; m.off[l] = l < image->n_levels ? image->offset[l] : 0; }
  k_env_func<<<dim3((unsigned)(((size_t)width * height + 255) / 256)), 256, 0, 0>>>(m, width, height, mode, il, delta, d_sin.as<float>(), d_func.as<float>());
  k_dist1_rows<<<dim3((unsigned)((height + 63) / 64)), 64, 0, 0>>>(d_func.as<float>(), height, width, d_cdf.as<float>(), d_int.as<float>());
  k_dist1_rows<<<1, 64, 0, 0>>>(d_int.as<float>(), 1, height, d_mcdf.as<float>(), d_mint.as<float>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(func, d_func.p, (size_t)width * height * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(cdf, d_cdf.p, (size_t)height * (width + 1) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(func_int, d_int.p, (size_t)height * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(marg_cdf, d_mcdf.p, (size_t)(height + 1) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(marg_func_int, d_mint.p, 4, hipMemcpyDeviceToHost));
  return RT_OK;
}
