// rtx_ref_launch.h - the reference-stream frame (rtx_ref.hip: one lane per 16 x 16 tile, the reference's one RNG stream per tile): what rt_render hands its kernel.
// Included after rtx_kernels.h.
#pragma once
namespace rtx {
struct RefParams {
  unsigned spp, dims;        // samples per pixel (a power of two), pre-generated dimensions
  int ntx, nty, tile;        // tiles of `tile` x `tile` sample-bounds pixels, row-major (renderer.rs:73-84)
  float* samples;            // [tile][3 * dims * spp]: the tile's samples_1d, then samples_2d (ZeroTwoSequence's arrays)
  int* stack;                // [tile][64]: the BVH walk's to-visit stack (bvh/mod.rs:375)
  float4* film_acc;          // the cropped film's (R, G, B, weight) sums
  const float* filter_table; // Film::filter_table, 16 x 16
  unsigned long long* stats; // camera rays, closest-hit rays, shadow rays, MIS rays, scrubbed samples
};
void rtx_launch_render_ref(const DScene& d, const FrameParams& fp, const RefParams& rp, hipStream_t stream);
void rtx_ref_set_ewa_lut(const float* lut128);  // kEwaLut of that translation unit
}
