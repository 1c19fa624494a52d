// rtx_shade.hip - the k_shade instantiations (K3, rtx_kernels.h) and their launches: a translation unit of its own, compiled beside rtx_hip.hip.
#include <hip/hip_runtime.h>
#include "../../include/rtx_hip.h"
#include "rtx_shade_kernels.h"
#include "rtx_shade_launch.h"

namespace rtx {
// one shade launch of front-end MODE: the GENERAL form (quadric / instance hits, masked emitters), the LEAN form (area lights and constant textures only;
// front-ends 3 / 5 / 6), or the plain one
template <int MODE>
static void launch_shade_t(bool general, bool lean, bool bounced, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p, bool qlights = false, int lds = 0) {
  // sphere lights over constant textures: the front-end ranges hold triangle vertices only (quadric hits are binned into the generic range)
  // lds (LEAN / QLIGHTS forms): 1 = the scene's records, lights, materials and textures fit the kernel's LDS, 2 = its materials and textures do (k_shade's LDSREC)
  if constexpr (MODE != 0) {
    if (qlights) {
      if (lds == 1) hipLaunchKernelGGL((k_shade<MODE, false, true, false, true, 1>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      else if (lds == 2) hipLaunchKernelGGL((k_shade<MODE, false, true, false, true, 2>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      else hipLaunchKernelGGL((k_shade<MODE, false, true, false, true>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      return;
    }
  }
  if (general) { hipLaunchKernelGGL((k_shade<MODE, true>), dim3(grid), dim3(block), 0, stream, d, fp, p); return; }
  if constexpr (MODE != 0) {
    if (lean) {
      if (lds == 2) hipLaunchKernelGGL((k_shade<MODE, false, true, false, false, 2>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      else hipLaunchKernelGGL((k_shade<MODE, false, true>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      return;
    }
  }
  // the Lambert front-end past the camera vertices: no differentials, bilinear image lookups, everything inline under a three-wave bound
  if constexpr (MODE == 3) {
    if (bounced) {
      if (lds == 3) hipLaunchKernelGGL((k_shade<3, false, false, true, false, 3>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      else hipLaunchKernelGGL((k_shade<3, false, false, true>), dim3(grid), dim3(block), 0, stream, d, fp, p);
      return;
    }
  }
  if constexpr (MODE == 3 || MODE == 5 || MODE == 6) { if (lds == 3) { hipLaunchKernelGGL((k_shade<MODE, false, false, false, false, 3>), dim3(grid), dim3(block), 0, stream, d, fp, p); return; } }
  hipLaunchKernelGGL((k_shade<MODE, false>), dim3(grid), dim3(block), 0, stream, d, fp, p);
}

void rtx_launch_shade(int mode, bool general, bool lean, bool bounced, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p, bool qlights, int lds) {
  switch (mode) {
    case 3: launch_shade_t<3>(general, lean, bounced, grid, block, stream, d, fp, p, qlights, lds); break;
    case 5: launch_shade_t<5>(general, lean, bounced, grid, block, stream, d, fp, p, qlights, lds); break;
    case 6: launch_shade_t<6>(general, lean, bounced, grid, block, stream, d, fp, p, qlights, lds); break;
    default: launch_shade_t<0>(general, lean, bounced, grid, block, stream, d, fp, p, qlights, lds); break;
  }
}
void rtx_launch_shade_const(int ldsrec, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p) {
  if (ldsrec == 1) hipLaunchKernelGGL((k_shade<1, false, false, false, false, 1>), dim3(grid), dim3(block), 0, stream, d, fp, p);
  else if (ldsrec == 3) hipLaunchKernelGGL((k_shade<1, false, false, false, false, 3>), dim3(grid), dim3(block), 0, stream, d, fp, p);
  else hipLaunchKernelGGL(k_shade<1>, dim3(grid), dim3(block), 0, stream, d, fp, p);
}
void rtx_shade_set_ewa_lut(const float* lut128) { (void)hipMemcpyToSymbol(HIP_SYMBOL(kEwaLut), lut128, 128 * sizeof(float)); }
}  // namespace rtx
