// Microbenchmark (GPU box): what rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts in the access patterns of the shade kernels (VERDICT r05 item 1a;
// MI355X_MICROARCH.md "HBM": the x2 correction is calibrated for 16-byte-per-lane streaming reads only). Every kernel has a name of its own so that the per-dispatch rows of a
// --pmc pass can be matched with the byte counts this program prints. Patterns:
//   cal_stream<W>            coalesced streaming read of a 2 GiB buffer, W dwords per lane (W = 1, 4)
//   cal_gather<W, SZ>        every lane loads W dwords at an independent random record of a table of SZ MB (16 / 150 / 1024), one load per record (the shade kernels' gathers)
//   cal_mix<NT>              the shade kernels' mix: per lane 2 x 16 B streamed in, 2 x 16 B streamed out, 4 random 4-byte gathers from a 150 MB table; NT = the streams carry the
//                            non-temporal hint. Tells whether the hint keeps the table in L2 / Infinity Cache while > 256 MiB of records stream past (time and FETCH_SIZE)
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/fetch_calibrate scripts/micro/fetch_calibrate.hip     Run: scripts/micro/fetch_calibrate [json-out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int W>
__global__ void __launch_bounds__(256) cal_stream(const float* __restrict__ src, float* __restrict__ out, size_t n_vec) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    if (W == 4) { const float4 v = ((const float4*)src)[i]; acc += v.x + v.y + v.z + v.w; }
    else acc += src[i];
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int W>
__global__ void __launch_bounds__(256) cal_stream_write(float* __restrict__ dst, size_t n_vec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    if (W == 4) ((float4*)dst)[i] = make_float4(1.0f, 2.0f, 3.0f, (float)i);
    else dst[i] = (float)i;
  }
}
// SZ only names the kernel (the table size in MB); n_rec = records of 64 bytes
template <int W, int SZ>
__global__ void __launch_bounds__(256) cal_gather(const float* __restrict__ tab, float* __restrict__ out, unsigned n_rec, unsigned per_lane, unsigned seed) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  for (unsigned k = 0; k < per_lane; ++k) {
    const unsigned r = hash32(gid * 9781u + k * 0x9e3779b9u + seed) % n_rec;
    const float* p = tab + 16 * (size_t)r + (W == 4 ? 4 * (hash32(r + k) & 3u) : (hash32(r + k) & 15u));
    if (W == 4) { const float4 v = *(const float4*)p; acc += v.x + v.w; } else acc += *p;
  }
  if (acc == 123.456f) out[0] = acc;
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) { const v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store4(float4 a, float4* p) { v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, (v4f*)p); }
template <int NT>
__global__ void __launch_bounds__(256) cal_mix(const float4* __restrict__ in_a, const float4* __restrict__ in_b, float4* __restrict__ out_a, float4* __restrict__ out_b,
                                                const float* __restrict__ tab, unsigned n_rec, size_t n_vec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    float4 a, b;
    if (NT) { a = nt_load4(in_a + i); b = nt_load4(in_b + i); } else { a = in_a[i]; b = in_b[i]; }
    float g = 0.0f;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) { const unsigned r = hash32((unsigned)i * 4u + k) % n_rec; g += tab[16 * (size_t)r + (hash32(r) & 15u)]; }
    a.x += g; b.y += g;
    if (NT) { nt_store4(a, out_a + i); nt_store4(b, out_b + i); } else { out_a[i] = a; out_b[i] = b; }
  }
}

static FILE* g_json = nullptr; static bool g_first = true;
static void report(const char* name, const char* state, double ms, double bytes_requested, double lines64, double lines128, double stores) {
  printf("%-28s %-5s %9.3f ms  requested %10.1f MB (%7.1f GB/s)  distinct 64-B sectors %10.1f MB  128-B lines %10.1f MB  stored %8.1f MB\n", name, state, ms, bytes_requested / 1e6,
         bytes_requested / ms / 1e6, lines64 / 1e6, lines128 / 1e6, stores / 1e6);
  if (g_json) { fprintf(g_json, "%s\n  {\"kernel\": \"%s\", \"state\": \"%s\", \"ms\": %.4f, \"bytes_requested\": %.0f, \"bytes_sectors64\": %.0f, \"bytes_lines128\": %.0f, \"bytes_stored\": %.0f}", g_first ? "" : ",", name, state, ms, bytes_requested, lines64, lines128, stores); g_first = false; }
}
// expected distinct lines of size LB touched by G uniform random gathers into a table of T bytes, in bytes
static double distinct(double G, double T, double LB) { const double L = T / LB; return LB * L * (1.0 - exp(-G / L)); }

int main(int argc, char** argv) {
  if (argc > 1) { g_json = fopen(argv[1], "w"); if (g_json) fprintf(g_json, "{\"runs\": ["); }
  const size_t big = (size_t)2 << 30;
  float *s0, *s1, *s2, *s3, *out, *tab;
  CK(hipMalloc(&s0, big)); CK(hipMalloc(&s1, big)); CK(hipMalloc(&s2, big)); CK(hipMalloc(&s3, big)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&tab, (size_t)1 << 30));
  CK(hipMemset(s0, 0, big)); CK(hipMemset(s1, 0, big)); CK(hipMemset(s2, 0, big)); CK(hipMemset(s3, 0, big)); CK(hipMemset(tab, 0, (size_t)1 << 30));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8, block = 256;
  float ms;
#define TIMED(launch) do { CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); } while (0)
  // streams
  TIMED((cal_stream<4><<<grid, block>>>(s0, out, big / 16))); report("cal_stream<4>", "cold", ms, (double)big, (double)big, (double)big, 0);
  TIMED((cal_stream<1><<<grid, block>>>(s1, out, big / 4)));  report("cal_stream<1>", "cold", ms, (double)big, (double)big, (double)big, 0);
  TIMED((cal_stream_write<4><<<grid, block>>>(s2, big / 16))); report("cal_stream_write<4>", "-", ms, 0, 0, 0, (double)big);
  TIMED((cal_stream_write<1><<<grid, block>>>(s3, big / 4)));  report("cal_stream_write<1>", "-", ms, 0, 0, 0, (double)big);
  // gathers: cold = after 4 GiB of streaming through the caches, warm = the same launch again (another seed: other records, same table)
  const unsigned per_lane = 32;
  const double G = (double)grid * block * per_lane;
#define GATHER(W, SZ) do { const double T = (double)SZ * 1e6; const unsigned n_rec = (unsigned)(T / 64.0); \
    cal_stream<4><<<grid, block>>>(s0, out, big / 16); cal_stream<4><<<grid, block>>>(s1, out, big / 16); \
    TIMED((cal_gather<W, SZ><<<grid, block>>>(tab, out, n_rec, per_lane, 1u))); report("cal_gather<" #W ", " #SZ ">", "cold", ms, G * 4 * W, distinct(G, T, 64), distinct(G, T, 128), 0); \
    TIMED((cal_gather<W, SZ><<<grid, block>>>(tab, out, n_rec, per_lane, 2u))); report("cal_gather<" #W ", " #SZ ">", "warm", ms, G * 4 * W, distinct(G, T, 64), distinct(G, T, 128), 0); \
    TIMED((cal_gather<W, SZ><<<grid, block>>>(tab, out, n_rec, per_lane, 3u))); report("cal_gather<" #W ", " #SZ ">", "warm2", ms, G * 4 * W, distinct(G, T, 64), distinct(G, T, 128), 0); } while (0)
  GATHER(1, 16); GATHER(4, 16); GATHER(1, 150); GATHER(4, 150); GATHER(1, 1024); GATHER(4, 1024);
  // the mix: 150 MB table, 2 GiB per stream = 8 GiB streamed per launch; the table is warmed first
  {
    const unsigned n_rec = (unsigned)(150e6 / 64.0); const size_t n_vec = big / 16;
    const double Gm = 4.0 * (double)n_vec;
    for (int nt = 0; nt < 2; ++nt) for (int rep = 0; rep < 2; ++rep) {
      cal_gather<1, 150><<<grid, block>>>(tab, out, n_rec, 64, 7u);
      if (nt) TIMED((cal_mix<1><<<grid, block>>>((const float4*)s0, (const float4*)s1, (float4*)s2, (float4*)s3, tab, n_rec, n_vec)));
      else TIMED((cal_mix<0><<<grid, block>>>((const float4*)s0, (const float4*)s1, (float4*)s2, (float4*)s3, tab, n_rec, n_vec)));
      report(nt ? "cal_mix<1>" : "cal_mix<0>", rep ? "rep1" : "rep0", ms, 2.0 * big + Gm * 4, 2.0 * big + distinct(Gm, 150e6, 64), 2.0 * big + distinct(Gm, 150e6, 128), 2.0 * big);
    }
  }
  if (g_json) { fprintf(g_json, "\n]}\n"); fclose(g_json); }
  return 0;
}
