// Is a reciprocal built from v_rcp_f32 and two fused Newton steps the correctly rounded 1 / x for EVERY float? Exhaustive over all 2^32 bit patterns against the
// compiler's IEEE division (-ffp-contract=off, no fast-math). Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/rcp_ex scripts/micro/rcp_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float rcp2(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
  return r;
}
__device__ __forceinline__ float rcp1(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
  return r;
}
__global__ void k(unsigned long long* out) {  // out[0..3]: mismatches of rcp2 in the normal range / outside it; of rcp1 likewise; out[4..5]: first mismatching pattern of rcp2 in range, count of in-range inputs
  unsigned long long bad2 = 0, bad2_out = 0, bad1 = 0, bad1_out = 0, n_in = 0; unsigned first = 0;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const unsigned u = (unsigned)i; float x = __uint_as_float(u);
    volatile float one = 1.0f;
    const float ref = one / x;
    const float a = rcp2(x), b = rcp1(x);
    const float ax = fabsf(x);
    const bool in_range = ax >= 1.1754944e-38f * 4.0f && ax <= 8.5070592e37f * 0.5f;  // [2^-124, 2^125]: x and 1 / x both normal with room
    const bool eq2 = __float_as_uint(a) == __float_as_uint(ref) || (a != a && ref != ref);
    const bool eq1 = __float_as_uint(b) == __float_as_uint(ref) || (b != b && ref != ref);
    if (in_range) { n_in++; if (!eq2) { bad2++; if (!first) first = u; } if (!eq1) bad1++; }
    else { if (!eq2) bad2_out++; if (!eq1) bad1_out++; }
  }
  atomicAdd(&out[0], bad2); atomicAdd(&out[1], bad2_out); atomicAdd(&out[2], bad1); atomicAdd(&out[3], bad1_out); atomicAdd(&out[5], n_in);
  if (first) atomicMax(&out[4], (unsigned long long)first);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 48); hipMemset(d, 0, 48);
  hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, d);
  unsigned long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
  printf("rcp + 2 Newton steps: %llu mismatches among %llu inputs with |x| in [2^-124, 2^125], %llu outside; one step: %llu / %llu; a mismatching pattern: 0x%08llx\n", h[0], h[5], h[1], h[2], h[3], h[4]);
  return 0;
}
