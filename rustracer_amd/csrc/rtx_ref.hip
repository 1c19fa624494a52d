// rtx_ref.hip - the REFERENCE'S OWN SAMPLER STREAM on the device (round 6; VERDICT r05 missing #2 / item 9): rc/renderer.rs:83-84 reseeds ONE PCG32 stream per 16 x 16 tile
// (set_sequence(tile.y * n_tiles.x + tile.x)) and every pixel and every sample of the tile consumes it in order - start_pixel's tables, then whatever the sample's path draws
// beyond the pre-generated dimensions. The number of draws a sample takes depends on its path, so a tile is one serial chain: the wavefront frame loop (rt_render) is defined on
// the pixel-KEYED variant of the stream instead and equals what rustracer-cli writes only statistically. Here the chain is kept: ONE LANE PER TILE walks its
// pixels and samples in the reference's order - ZeroTwoSequence::start_pixel (rc/sampler/zerotwosequence.rs:67-108), get_camera_sample, PathIntegrator::li
// (rc/integrator/path.rs:96-215) with its rays traced inline, FilmTile::add_sample - with the device functions the wavefront kernels are made of (same BVH walk, same
// interactions, materials, lights, BxDFs). Slow by construction (a few hundred lanes of serial work) and exact in the sense that matters: the same samples reach the same
// pixels with the same weights as the oracle's SAMPLER_REF mode, the radiance inside the image gate. For the configuration the reference itself runs (BASELINE configs[0]:
// 400 x 400 x 64 spp = 625 tiles). Plain-triangle scenes; every material, texture and light of the path. A translation unit of its own, compiled beside the other two.
#include <hip/hip_runtime.h>
#include "../../include/rtx_hip.h"
#include "rtx_shade_kernels.h"
#include "rtx_ref_launch.h"

namespace rtx {

struct RefSampler {  // ZeroTwoSequence over per-tile arrays in device memory: s1[d][i] = samples_1d[d][i], s2[d][i] = samples_2d[d][i]
  Pcg32 rng; float* s1; float2* s2; unsigned spp, dims, cur1, cur2, idx;
  RT_DEV void shuffle1(float* a, unsigned count) {  // shuffle(samp, count, 1, rng), lowdiscrepancy.rs:114-124
    for (unsigned i = 0; i < count; ++i) { const unsigned other = i + rng.bounded(count - i); const float t = a[i]; a[i] = a[other]; a[other] = t; }
  }
  RT_DEV void shuffle2(float2* a, unsigned count) {
    for (unsigned i = 0; i < count; ++i) { const unsigned other = i + rng.bounded(count - i); const float2 t = a[i]; a[i] = a[other]; a[other] = t; }
  }
  RT_DEV void start_pixel() {  // zerotwosequence.rs:67-108 (REF mode: the stream is the tile's, it is not re-keyed)
    for (unsigned d = 0; d < dims; ++d) {  // van_der_corput(1, spp, ..), lowdiscrepancy.rs:4-23
      float* a = s1 + (size_t)d * spp;
      const unsigned scramble = rng.next_u32();
      for (unsigned i = 0; i < spp; ++i) a[i] = u32_to_unit(scramble ^ vdc_bits(i));  // gray_code_sample's value at index i (rtx_dev_math.h: the running XOR in closed form)
      for (unsigned i = 0; i < spp; ++i) (void)rng.bounded(1u);  // shuffle(samples + i, 1, 1): one bounded draw each, the swap is with itself
      shuffle1(a, spp);
    }
    for (unsigned d = 0; d < dims; ++d) {  // sobol_2d(1, spp, ..), :25-50
      float2* a = s2 + (size_t)d * spp;
      const unsigned sc0 = rng.next_u32(); const unsigned sc1 = rng.next_u32();
      for (unsigned i = 0; i < spp; ++i) a[i] = make_float2(u32_to_unit(sc0 ^ vdc_bits(i)), u32_to_unit(sc1 ^ sobol1_bits(i)));  // gray_code_sample_2d over CSOBOL[0], CSOBOL[1]
      for (unsigned i = 0; i < spp; ++i) (void)rng.bounded(1u);
      shuffle2(a, spp);
    }
    idx = 0;  // (cur1 / cur2 were reset by the previous pixel's last start_next_sample, :110-117)
  }
  RT_DEV bool start_next_sample() { cur1 = 0; cur2 = 0; idx += 1; return idx < spp; }
  RT_DEV float get_1d() { if (cur1 < dims) return s1[(size_t)(cur1++) * spp + idx]; return rng.next_f32(); }  // :158-166
  RT_DEV f2 get_2d() {  // :168-180 (the RNG fallback returns (second draw, first draw))
    if (cur2 < dims) { const float2 v = s2[(size_t)(cur2++) * spp + idx]; return mk2(v.x, v.y); }
    const float x = rng.next_f32(); const float y = rng.next_f32();
    return mk2(y, x);
  }
};
struct RefCounts { unsigned long long camera, closest, shadow, mis, scrubbed; };

// estimate_direct (integrator/mod.rs:222-318) with both rays traced where the reference traces them
RT_DEVN rgb3 ref_estimate_direct(const DScene* self, const SurfaceInteraction* sip, const GenericBsdf* bp, float us_x, float us_y, int light_num, float ul_x, float ul_y, int* stack, RefCounts* st) {
  const DScene& sc = *self; const SurfaceInteraction& si = *sip; const GenericBsdf& bsdf = *bp;
  const DLight& light = sc.lights[light_num];
  const unsigned nonspec = BSDF_ALL & ~BSDF_SPECULAR;
  const GlobalSrc src{sc.nodes, sc.tri_p};
  rgb3 ld = mkc(0, 0, 0);
  const LiSample ls = light_sample_li_full<false, false, false>(sc, light, si.hit, mk2(ul_x, ul_y));
  if (ls.pdf > 0.0f && !is_black(ls.li)) {
    const rgb3 f = bsdf.f(si.hit.wo, ls.wi, nonspec) * fabsf(dot(ls.wi, si.sh_n));
    const float scattering_pdf = light_is_delta(light) ? 0.0f : bsdf.pdf(si.hit.wo, ls.wi, nonspec);
    if (!is_black(f)) {
      const Ray sr = spawn_ray_to_interaction(si.hit, ls.p1);  // VisibilityTester::unoccluded, light/mod.rs:52-55
      int prim = -1; TriHit h; unsigned nn = 0, nt = 0;
      st->shadow += 1;
      const bool occluded = traverse<true, false, GlobalSrc, int>(src, sr, stack, 1, prim, h, nn, nt);
      if (!occluded) {
        if (light_is_delta(light)) ld = ld + vdiv(f * ls.li, ls.pdf);
        else ld = ld + vdiv(f * ls.li * power_heuristic1(ls.pdf, scattering_pdf), ls.pdf);
      }
    }
  }
  if (!light_is_delta(light)) {
    const LobeSample bs = bsdf.sample_f(si.hit.wo, mk2(us_x, us_y), nonspec);
    const rgb3 f = bs.f * fabsf(dot(bs.wi, si.sh_n));
    if (!is_black(f) && bs.pdf > 0.0f) {
      float weight = 1.0f;
      if (!(bs.type & BSDF_SPECULAR)) {
        const float lp = light_pdf_li<false, false>(sc, light, si.hit, bs.wi);
        if (lp == 0.0f) return ld;
        weight = power_heuristic1(bs.pdf, lp);
      }
      const Ray mr = spawn_ray(si.hit, bs.wi);
      int prim = -1; TriHit h; h.t = kInf; h.b0 = h.b1 = h.b2 = 0.0f; unsigned nn = 0, nt = 0;
      st->mis += 1;
      rgb3 li2 = mkc(0, 0, 0);
      if (traverse<false, false, GlobalSrc, int>(src, mr, stack, 1, prim, h, nn, nt)) {
        if (rec_light(sc.tri_rec, prim) == light_num) { f3 p, n; tri_hit_point_normal(sc, prim, h, p, n); li2 = area_light_l(light, n, -bs.wi); }  // area_light.id() == light.id(), mod.rs:295-304
      } else if (light.kind == 3) li2 = infinite_le(sc, light, mr.d);  // light.le(ray): only an infinite light has one
      if (!is_black(li2)) ld = ld + vdiv(f * li2 * weight, bs.pdf);
    }
  }
  return ld;
}

// PathIntegrator::li (path.rs:96-215) for one camera ray
RT_DEVN rgb3 ref_li(const DScene* self, const FrameParams* fpp, const CameraRay* crp, RefSampler* smp, int* stack, RefCounts* st) {
  const DScene& sc = *self; const FrameParams& fp = *fpp; const CameraRay& cr = *crp;
  const GlobalSrc src{sc.nodes, sc.tri_p};
  rgb3 L = mkc(0, 0, 0), beta = mkc(1, 1, 1);
  bool specular_bounce = false; int bounces = 0; float eta_scale = 1.0f;
  Ray ray; ray.o = cr.o; ray.d = cr.d; ray.t_max = kInf;
  for (;;) {
    int prim = -1; TriHit h; h.t = kInf; h.b0 = h.b1 = h.b2 = 0.0f; unsigned nn = 0, nt = 0;
    st->closest += 1;
    const bool found = traverse<false, false, GlobalSrc, int>(src, ray, stack, 1, prim, h, nn, nt);
    SurfaceInteraction si;
    if (found) tri_fill_interaction(sc, prim, ray.d, h, si);
    if (bounces == 0 || specular_bounce) {  // path.rs:127-136
      if (found) { const int li = rec_light(sc.tri_rec, prim); if (li >= 0) L = L + beta * area_light_l(sc.lights[li], si.hit.n, -ray.d); }
      else for (int k = 0; k < sc.n_infinite; ++k) L = L + beta * infinite_le(sc, sc.lights[sc.infinite_ids[k]], ray.d);
    }
    if (!found || bounces >= fp.max_depth) break;
    if (bounces == 0 && sc.needs_differentials) compute_differential_call(si, cr.rx_o, cr.ry_o, cr.rx_d, cr.ry_d);  // only the camera ray carries differentials (interaction.rs:245-314)
    GenericBsdf bsdf;
    bsdf.build(sc, rec_material(sc.tri_rec, prim), si);
    // light_distribution.lookup(p) (path.rs:154-158)
    const float* ld_func = sc.ld_func; const float* ld_cdf = sc.ld_cdf; float ld_int;
    if (sc.ld_uniform) ld_int = sc.ld_int[0];
    else {
      const long slot = sc.ld_slot[voxel_of(sc, si.hit.p)];
      if (slot >= 0) { ld_func = sc.ld_func + slot * sc.n_lights; ld_cdf = sc.ld_cdf + slot * (sc.n_lights + 1); ld_int = sc.ld_int[slot]; }
      else ld_int = -1.0f;  // (a voxel the eager build did not mark: cannot hold a surface point)
    }
    if (!(ld_int < 0.0f) && bsdf.num_nonspecular() > 0 && sc.n_lights > 0) {  // uniform_sample_one_light, integrator/mod.rs:186-220
      const float su = smp->get_1d();
      int light_num; float light_pdf;
      d1_sample_discrete(ld_func, ld_cdf, ld_int, sc.n_lights, su, light_num, light_pdf);
      if (light_pdf != 0.0f) {
        const f2 u_light = smp->get_2d(); const f2 u_scattering = smp->get_2d();
        const rgb3 ld = ref_estimate_direct(self, &si, &bsdf, u_scattering.x, u_scattering.y, light_num, u_light.x, u_light.y, stack, st);
        L = L + beta * vdiv(ld, light_pdf);
      }
    }
    const f3 wo = -ray.d;  // not normalised (reference quirk)
    const LobeSample bs = bsdf.sample_f(wo, smp->get_2d(), BSDF_ALL);
    if (is_black(bs.f) || bs.pdf <= 0.0f) break;
    beta = vdiv(beta * bs.f * fabsf(dot(bs.wi, si.sh_n)), bs.pdf);
    specular_bounce = (bs.type & BSDF_SPECULAR) != 0u;
    if ((bs.type & BSDF_SPECULAR) && (bs.type & BSDF_TRANSMISSION)) {
      const float eta = bsdf.eta();
      eta_scale *= dot(wo, si.hit.n) > 0.0f ? eta * eta : vdiv(1.0f, eta * eta);
    }
    ray = spawn_ray(si.hit, bs.wi);
    const rgb3 rr_beta = beta * eta_scale;  // path.rs:201-209
    if (max_component_value(rr_beta) < fp.rr_threshold && bounces > 3) {
      const float q = fmaxf(1.0f - max_component_value(rr_beta), 0.05f);
      if (smp->get_1d() < q) break;
      beta = vdiv(beta, 1.0f - q);
    }
    bounces += 1;
  }
  return L;
}

// one lane per tile of rp.tile x rp.tile sample-bounds pixels, in the reference's order (renderer.rs:73-131)
__global__ void __launch_bounds__(64) k_render_ref(DScene sc, FrameParams fp, RefParams rp) {
  const int tile = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (tile >= rp.ntx * rp.nty) return;
  const int tx = tile % rp.ntx, ty = tile / rp.ntx;
  RefSampler smp; smp.spp = rp.spp; smp.dims = rp.dims; smp.cur1 = smp.cur2 = 0; smp.idx = 0;
  smp.s1 = rp.samples + (size_t)tile * 3u * rp.dims * rp.spp; smp.s2 = (float2*)(smp.s1 + (size_t)rp.dims * rp.spp);
  smp.rng.set_sequence((unsigned long long)(ty * rp.ntx + tx));  // sampler.reseed(tile.y * n_tiles.x + tile.x), renderer.rs:83-84
  int* const stack = rp.stack + (size_t)tile * 64;
  RefCounts st{0, 0, 0, 0, 0};
  const int x0 = fp.sb_x0 + tx * rp.tile, x1 = min(x0 + rp.tile, fp.sb_x1), y0 = fp.sb_y0 + ty * rp.tile, y1 = min(y0 + rp.tile, fp.sb_y1);
  const int cw = fp.crop_x1 - fp.crop_x0;
  const float inv_rx = 1.0f / fp.radius_x, inv_ry = 1.0f / fp.radius_y;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      smp.start_pixel();  // every pixel of the tile advances the stream, the ones pixel_bounds skips too (renderer.rs:96-104)
      if (!(x >= fp.pb_x0 && x < fp.pb_x1 && y >= fp.pb_y0 && y < fp.pb_y1)) continue;
      rgb3 own = mkc(0, 0, 0); float own_w = 0.0f;
      for (;;) {
        const f2 o = smp.get_2d();  // get_camera_sample, zerotwosequence.rs:182-192
        const f2 p_film = mk2((float)x + o.x, (float)y + o.y);
        (void)smp.get_1d();
        const f2 p_lens = smp.get_2d();
        const CameraRay cr = generate_camera_ray(fp, p_film, p_lens, 1.0f / sqrtf((float)rp.spp));
        st.camera += 1;
        rgb3 c = ref_li(sc.self, &fp, &cr, &smp, stack, &st);
        bool bad = false;  // renderer.rs:115-126
        if (has_nan(c)) { c = mkc(0, 0, 0); bad = true; }
        if (lum_y(c) < -1e-5f) { c = mkc(0, 0, 0); bad = true; }
        if (isinf(lum_y(c))) { c = mkc(0, 0, 0); bad = true; }
        st.scrubbed += bad ? 1u : 0u;
        {  // FilmTile::add_sample (film.rs:298-361): the pixel's own sum in registers in sample order, what the filter spreads onto other pixels through float atomics
          const rgb3 Lc = lum_y(c) > fp.max_sample_luminance ? c * fp.max_sample_luminance / lum_y(c) : c;
          const float dx = p_film.x - 0.5f, dy = p_film.y - 0.5f;
          const float p0x = ceilf(dx - fp.radius_x), p0y = ceilf(dy - fp.radius_y), p1x = floorf(dx + fp.radius_x + 1.0f), p1y = floorf(dy + fp.radius_y + 1.0f);
          const int fx0 = f2i_sat(max_po(p0x, (float)fp.crop_x0)), fy0 = f2i_sat(max_po(p0y, (float)fp.crop_y0));
          const int fx1 = f2i_sat(min_po(p1x, (float)fp.crop_x1)), fy1 = f2i_sat(min_po(p1y, (float)fp.crop_y1));
          for (int yy = fy0; yy < fy1; ++yy) {
            const int iy = (int)f2u_sat(fminf(floorf(fabsf(((float)yy - dy) * inv_ry * 16.0f)), 15.0f));
            for (int xx = fx0; xx < fx1; ++xx) {
              const int ix = (int)f2u_sat(fminf(floorf(fabsf(((float)xx - dx) * inv_rx * 16.0f)), 15.0f));
              const float fw = rp.filter_table[iy * 16 + ix];
              if (xx == x && yy == y) { own = own + Lc * fw; own_w += fw; }
              else {
                float* dst = (float*)&rp.film_acc[(size_t)(yy - fp.crop_y0) * cw + (xx - fp.crop_x0)];
                const rgb3 v = Lc * fw;
                atomicAdd(dst + 0, v.r); atomicAdd(dst + 1, v.g); atomicAdd(dst + 2, v.b); atomicAdd(dst + 3, fw);
              }
            }
          }
        }
        if (!smp.start_next_sample()) break;
      }
      if (x >= fp.crop_x0 && x < fp.crop_x1 && y >= fp.crop_y0 && y < fp.crop_y1) {
        float* dst = (float*)&rp.film_acc[(size_t)(y - fp.crop_y0) * cw + (x - fp.crop_x0)];
        atomicAdd(dst + 0, own.r); atomicAdd(dst + 1, own.g); atomicAdd(dst + 2, own.b); atomicAdd(dst + 3, own_w);
      }
    }
  atomicAdd(&rp.stats[0], st.camera); atomicAdd(&rp.stats[1], st.closest); atomicAdd(&rp.stats[2], st.shadow); atomicAdd(&rp.stats[3], st.mis); atomicAdd(&rp.stats[4], st.scrubbed);
}

void rtx_launch_render_ref(const DScene& d, const FrameParams& fp, const RefParams& rp, hipStream_t stream) {
  const unsigned n = (unsigned)(rp.ntx * rp.nty);
  hipLaunchKernelGGL(k_render_ref, dim3((n + 63u) / 64u), dim3(64), 0, stream, d, fp, rp);
}
void rtx_ref_set_ewa_lut(const float* lut128) { (void)hipMemcpyToSymbol(HIP_SYMBOL(kEwaLut), lut128, 128 * sizeof(float)); }
}  // namespace rtx
