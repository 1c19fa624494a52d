// rtx_shade_launch.h - the launches of the k_shade instantiations, which live in a translation unit of their own (rtx_shade.hip) so that the library's two halves
// compile side by side (a clean build: ~1 min instead of ~2). Included after rtx_kernels.h.
#pragma once
namespace rtx {
// one shade launch of front-end MODE (0 generic, 3 Lambert, 5 / 6 two-lobe): the GENERAL form (quadric / instance hits, masked emitters), the LEAN form (area lights and
// constant textures only), its QLIGHTS form (area lights on analytic spheres), the BOUNCED form of front-end 3, or the plain one; lds = k_shade's LDSREC
void rtx_launch_shade(int mode, bool general, bool lean, bool bounced, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p, bool qlights, int lds);
// k_shade<1> (constant matte, area lights): ldsrec 0 / 1 / 3
void rtx_launch_shade_const(int ldsrec, unsigned grid, unsigned block, hipStream_t stream, const DScene& d, const FrameParams& fp, const PassState& p);
void rtx_shade_set_ewa_lut(const float* lut128);  // kEwaLut of that translation unit
}
