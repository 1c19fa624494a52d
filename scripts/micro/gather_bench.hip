// Microbenchmark (GPU box): what a wave pays for gathering one 64-byte record per lane from an L2-resident table -
// (A) every lane loads its own record with four 16-byte loads (64 distinct lines per instruction), against
// (B) the same 64 records fetched cooperatively: four lanes share a record, each instruction covers 16 records (16 distinct 64-byte segments), values
//     exchanged through LDS. Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench scripts/micro/gather_bench.hip ; run: /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__global__ void k_own(const float4* __restrict__ tab, const unsigned* __restrict__ idx, float* out, int iters, unsigned n_rec) {
  unsigned r = idx[blockIdx.x * blockDim.x + threadIdx.x];
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const float4* p = tab + 4 * (size_t)r;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
    r = (r * 1664525u + 1013904223u + __float_as_uint(a.x)) % n_rec;  // next record depends on the data (a dependent chain, as in a tree walk)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_coop(const float4* __restrict__ tab, const unsigned* __restrict__ idx, float* out, int iters, unsigned n_rec) {
  __shared__ float4 sh[256 * 4];
  __shared__ unsigned sr[256];
  const unsigned lane = threadIdx.x & 63u, wbase = threadIdx.x & ~63u;
  unsigned r = idx[blockIdx.x * blockDim.x + threadIdx.x];
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    sr[threadIdx.x] = r;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // instruction k: lane l fetches quarter (l & 3) of the record of lane 16 k + (l >> 2)
      const unsigned owner = 16u * k + (lane >> 2);
      const unsigned rr = sr[wbase + owner];
      sh[(wbase + owner) * 4 + (lane & 3u)] = tab[4 * (size_t)rr + (lane & 3u)];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const float4 a = sh[threadIdx.x * 4], b = sh[threadIdx.x * 4 + 1], c = sh[threadIdx.x * 4 + 2], d = sh[threadIdx.x * 4 + 3];
    acc += a.x + b.y + c.z + d.w;
    r = (r * 1664525u + 1013904223u + __float_as_uint(a.x)) % n_rec;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  for (unsigned mb : {1u, 16u, 128u}) {
    const unsigned n_rec = mb * (1u << 20) / 64u;
    std::vector<float> h((size_t)n_rec * 16, 0.0f);
    std::mt19937 g(1); for (auto& v : h) v = (float)(g() & 1023u) * 1e-9f;
    const int grid = 256 * 6, block = 256, iters = 2000;
    std::vector<unsigned> hi((size_t)grid * block); for (auto& v : hi) v = g() % n_rec;
    float4* tab; unsigned* idx; float* out;
    hipMalloc(&tab, h.size() * 4); hipMalloc(&idx, hi.size() * 4); hipMalloc(&out, hi.size() * 4);
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_own, dim3(grid), dim3(block), 0, 0, tab, idx, out, iters, n_rec);
        else hipLaunchKernelGGL(k_coop, dim3(grid), dim3(block), 0, 0, tab, idx, out, iters, n_rec);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) printf("table %4u MB  %s  %8.2f ms  %7.1f G records/s  %6.2f TB/s of records\n", mb, which == 0 ? "own-record gathers " : "cooperative + LDS   ", ms,
                             (double)grid * block * iters / ms / 1e6, (double)grid * block * iters * 64.0 / ms / 1e9);
      }
    }
    hipFree(tab); hipFree(idx); hipFree(out);
  }
  return 0;
}
