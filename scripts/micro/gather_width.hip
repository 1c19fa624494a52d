// Microbenchmark (GPU box): the cost of a gather by width and count - every lane reads N loads of W dwords from its own random 64-byte record of an L2-resident
// table, the next record depending on the data (a walk). Prints lane-loads per clock per CU. Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gw scripts/micro/gather_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
template <int N, int W>
__global__ void k(const float* __restrict__ tab, const unsigned* __restrict__ idx, float* out, int iters, unsigned n_rec) {
  unsigned r = idx[blockIdx.x * blockDim.x + threadIdx.x];
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const float* p = tab + 16 * (size_t)r;
    float first = 0.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (W == 4) { const float4 v = ((const float4*)p)[k]; acc += v.x + v.w; if (k == 0) first = v.x; }
      else if (W == 2) { const float2 v = ((const float2*)p)[k]; acc += v.x + v.y; if (k == 0) first = v.x; }
      else { const float v = p[k]; acc += v; if (k == 0) first = v; }
    }
    r = (r * 1664525u + 1013904223u + __float_as_uint(first)) % n_rec;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int N, int W> void run(const float* tab, const unsigned* idx, float* out, unsigned n_rec, int grid, int block, int iters, double clk) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, W>), dim3(grid), dim3(block), 0, 0, tab, idx, out, iters, n_rec);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double lane_loads = (double)grid * block * iters * N;
  printf("%d x %2d-byte loads per record: %7.2f ms  %6.2f lane-loads / clk / CU  %6.1f bytes / clk / CU  %6.0f ns per round\n", N, 4 * W, ms, lane_loads / (ms * 1e-3) / clk / 256.0,
         lane_loads * 4 * W / (ms * 1e-3) / clk / 256.0, ms * 1e6 / iters);
}
int main() {
  const unsigned n_rec = 16u * (1u << 20) / 64u;
  std::vector<float> h((size_t)n_rec * 16, 0.0f);
  std::mt19937 g(1); for (auto& v : h) v = (float)(g() & 1023u) * 1e-9f;
  const int grid = 256 * 6, block = 256, iters = 2000;
  std::vector<unsigned> hi((size_t)grid * block); for (auto& v : hi) v = g() % n_rec;
  float* tab; unsigned* idx; float* out;
  (void)hipMalloc(&tab, h.size() * 4); (void)hipMalloc(&idx, hi.size() * 4); (void)hipMalloc(&out, hi.size() * 4);
  (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  const double clk = 2.35e9;
  run<1, 4>(tab, idx, out, n_rec, grid, block, iters, clk); run<2, 4>(tab, idx, out, n_rec, grid, block, iters, clk); run<4, 4>(tab, idx, out, n_rec, grid, block, iters, clk);
  run<1, 1>(tab, idx, out, n_rec, grid, block, iters, clk); run<4, 1>(tab, idx, out, n_rec, grid, block, iters, clk); run<8, 1>(tab, idx, out, n_rec, grid, block, iters, clk);
  run<2, 2>(tab, idx, out, n_rec, grid, block, iters, clk); run<8, 2>(tab, idx, out, n_rec, grid, block, iters, clk);
  return 0;
}
