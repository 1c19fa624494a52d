// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// PCG32 (rc/rng.rs), low-discrepancy generators (rc/sampler/lowdiscrepancy.rs) and the
// ZeroTwoSequence sampler (rc/sampler/zerotwosequence.rs).
//
// Two seeding modes (SURVEY.md §7 "hard parts", §8c):
//   REF   — bit-faithful to rc/renderer.rs:83-84: one PCG32 stream per 16x16 tile
//           (set_sequence(tile.y*n_tiles.x + tile.x)), consumed sequentially by every pixel and
//           every sample of the tile. Inherently serial; used for the CPU baseline.
//   KEYED — the parity contract with the GPU: before start_pixel(p) the stream is re-seeded with
//           set_sequence(pixel_index) (pixel_index = row-major index of p inside the sample
//           bounds); start_pixel then runs exactly as in the reference; before each sample s the
//           RNG used for draws beyond the pre-generated tables is re-seeded with
//           set_sequence(pixel_index * spp + s + 2^32).
#pragma once
#include <vector>
#include "orc_math.h"

namespace orc {

// ---------------------------------------------------------------- rc/rng.rs
struct Rng {
  uint64_t state = 0x853c49e6748fea9bULL;  // PCG32_DEFAULT_STATE :5
  uint64_t inc = 0xda3e39cb94b95bdbULL;    // PCG32_DEFAULT_STREAM :6
  uint32_t uniform_u32() {                 // :23-30
    uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
  }
  uint32_t uniform_u32_bounded(uint32_t b) {  // :32-40  (quirk 14: threshold = (!b+1)&b)
    uint32_t threshold = (~b + 1u) & b;
    for (;;) {
      uint32_t r = uniform_u32();
      if (r >= threshold) return r % b;
    }
  }
  float uniform_f32() {  // :42-44
    return fminf((float)uniform_u32() * 2.3283064365386963e-10f, kOneMinusEpsilon);
  }
  void set_sequence(uint64_t seed) {  // :46-52
    state = 0;
    inc = (seed << 1u) | 1u;
    (void)uniform_u32();
    state += 0x853c49e6748fea9bULL;
    (void)uniform_u32();
  }
};

// ---------------------------------------------------------------- rc/sampler/lowdiscrepancy.rs
inline uint32_t cvdc(uint32_t i) { return 0x80000000u >> i; }  // CVAN_DER_CORPUT :126-159 (identity matrix)
inline uint32_t csobol0(uint32_t i) { return 0x80000000u >> i; }  // CSOBOL[0] :162-166
static const uint32_t kCSobol1[32] = {  // CSOBOL[1] :167-173
    0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000,
    0x80800000, 0xc0c00000, 0xa0a00000, 0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000,
    0x80008000, 0xc000c000, 0xa000a000, 0xf000f000, 0x88008800, 0xcc00cc00, 0xaa00aa00, 0xff00ff00,
    0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc, 0xaaaaaaaa, 0xffffffff};

struct P2 { float x, y; };

inline uint32_t ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }

// gray_code_sample :96-102
inline void gray_code_sample(uint32_t n, uint32_t scramble, float* p) {
  uint32_t v = scramble;
  for (uint32_t i = 0; i < n; ++i) {
    p[i] = fminf((float)v * 2.3283064365386963e-10f, kOneMinusEpsilon);
    v ^= cvdc(ctz32(i + 1));
  }
}
// gray_code_sample_2d :104-112
inline void gray_code_sample_2d(uint32_t n, uint32_t s0, uint32_t s1, P2* p) {
  uint32_t v0 = s0, v1 = s1;
  for (uint32_t i = 0; i < n; ++i) {
    p[i].x = fminf((float)v0 * 2.3283064365386963e-10f, kOneMinusEpsilon);
    p[i].y = fminf((float)v1 * 2.3283064365386963e-10f, kOneMinusEpsilon);
    v0 ^= csobol0(ctz32(i + 1));
    v1 ^= kCSobol1[ctz32(i + 1)];
  }
}
// shuffle :114-124
template <class T>
inline void shuffle(T* samp, uint32_t count, uint32_t n_dimensions, Rng& rng) {
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t other = i + rng.uniform_u32_bounded(count - i);
    for (uint32_t j = 0; j < n_dimensions; ++j) {
      T t = samp[n_dimensions * i + j];
      samp[n_dimensions * i + j] = samp[n_dimensions * other + j];
      samp[n_dimensions * other + j] = t;
    }
  }
}
// van_der_corput :4-23
inline void van_der_corput(uint32_t n_per_pixel_sample, uint32_t n_pixel_samples, float* samples, Rng& rng) {
  uint32_t scramble = rng.uniform_u32();
  uint32_t total = n_per_pixel_sample * n_pixel_samples;
  gray_code_sample(total, scramble, samples);
  for (uint32_t i = 0; i < n_pixel_samples; ++i) shuffle(samples + (size_t)i * n_per_pixel_sample, n_per_pixel_sample, 1u, rng);
  shuffle(samples, n_pixel_samples, n_per_pixel_sample, rng);
}
// sobol_2d :25-50
inline void sobol_2d(uint32_t n_per_pixel_sample, uint32_t n_pixel_samples, P2* samples, Rng& rng) {
  uint32_t s0 = rng.uniform_u32();
  uint32_t s1 = rng.uniform_u32();
  gray_code_sample_2d(n_per_pixel_sample * n_pixel_samples, s0, s1, samples);
  for (uint32_t i = 0; i < n_pixel_samples; ++i) shuffle(samples + (size_t)i * n_per_pixel_sample, n_per_pixel_sample, 1u, rng);
  shuffle(samples, n_pixel_samples, n_per_pixel_sample, rng);
}

// radical_inverse :52-94
inline uint32_t reverse_bits_32(uint32_t n) {
  n = (n << 16) | (n >> 16);
  n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
  n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
  n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
  n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
  return n;
}
inline uint64_t reverse_bits_64(uint64_t n) {
  uint64_t n0 = reverse_bits_32((uint32_t)n), n1 = reverse_bits_32((uint32_t)(n >> 32));
  return (n0 << 32) | n1;
}
inline float radical_inverse_specialized(uint32_t base, uint64_t a) {
  float inv_base = 1.0f / (float)base;
  uint64_t reversed = 0;
  float inv_base_n = 1.0f;
  while (a != 0) {
    uint64_t next = a / base;
    uint64_t digit = a - next * base;
    reversed = reversed * base + digit;
    inv_base_n *= inv_base;
    a = next;
  }
  return fminf((float)reversed * inv_base_n, kOneMinusEpsilon);
}
inline float radical_inverse(uint32_t base_index, uint64_t a) {
  switch (base_index) {
    case 0: return (float)reverse_bits_64(a) * 5.4210108624275222e-20f;
    case 1: return radical_inverse_specialized(3, a);
    case 2: return radical_inverse_specialized(5, a);
    case 3: return radical_inverse_specialized(7, a);
    case 4: return radical_inverse_specialized(11, a);
    default: return radical_inverse_specialized(13, a);
  }
}

// ---------------------------------------------------------------- rc/sampler/zerotwosequence.rs
enum SamplerMode { SAMPLER_REF = 0, SAMPLER_KEYED = 1 };

struct ZeroTwoSequence {
  uint32_t spp = 1;   // rounded up to pow2 (:32)
  uint32_t ndims = 4; // "dimensions" default 4 (:60)
  std::vector<std::vector<float>> samples_1d;
  std::vector<std::vector<P2>> samples_2d;
  uint32_t cur_1d = 0, cur_2d = 0, sample_index = 0;
  Rng rng;
  SamplerMode mode = SAMPLER_REF;
  uint64_t pixel_key = 0;  // KEYED only

  static uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }
  void init(uint32_t spp_, uint32_t nd, SamplerMode m) {
    spp = next_pow2(spp_ == 0 ? 1 : spp_);
    ndims = nd; mode = m;
    samples_1d.assign(nd, std::vector<float>(spp, 0.0f));
    samples_2d.assign(nd, std::vector<P2>(spp, P2{0.0f, 0.0f}));
    cur_1d = cur_2d = sample_index = 0;
    rng = Rng();
  }
  void reseed(uint64_t seed) { rng.set_sequence(seed); }  // :197-199
  // :67-108 ; `pixel_index` only used in KEYED mode
  void start_pixel(uint64_t pixel_index) {
    if (mode == SAMPLER_KEYED) { pixel_key = pixel_index; rng.set_sequence(pixel_index); }
    for (uint32_t i = 0; i < ndims; ++i) van_der_corput(1, spp, samples_1d[i].data(), rng);
    for (uint32_t i = 0; i < ndims; ++i) sobol_2d(1, spp, samples_2d[i].data(), rng);
    sample_index = 0;
    // NB: current_{1d,2d}_dimension are NOT reset here in the reference; they are 0 because
    // start_next_sample() of the previous pixel's last sample reset them (:110-117).
    if (mode == SAMPLER_KEYED) begin_sample_keyed();
  }
  void begin_sample_keyed() { rng.set_sequence(pixel_key * (uint64_t)spp + sample_index + (1ULL << 32)); }
  bool start_next_sample() {  // :110-117
    cur_1d = 0; cur_2d = 0;
    sample_index += 1;
    bool more = sample_index < spp;
    if (more && mode == SAMPLER_KEYED) begin_sample_keyed();
    return more;
  }
  float get_1d() {  // :158-166
    if (cur_1d < ndims) return samples_1d[cur_1d++][sample_index];
    return rng.uniform_f32();
  }
  P2 get_2d() {  // :168-180 — RNG fallback returns (second draw, first draw)
    if (cur_2d < ndims) return samples_2d[cur_2d++][sample_index];
    float x = rng.uniform_f32();
    float y = rng.uniform_f32();
    return P2{y, x};
  }
};

}  // namespace orc
