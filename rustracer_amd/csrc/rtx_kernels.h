// rtx_kernels.h — the wavefront kernels (gfx950). One frame = for each pass (a chunk of pixels x all
// spp): sampler_tables -> raygen -> { trace_closest -> shade -> trace_any(shadow) ->
// trace_closest(MIS) -> resolve } x (max_depth + 1) -> film_accumulate; then film_finalize.
// Every kernel is launched with a fixed persistent grid and strides over a device-side queue
// whose length is read from device memory, so a frame needs no host round trip.
#pragma once
#include <type_traits>
#include "rtx_dev_shading.h"

namespace rtx {


// ---- per-pass path state in HBM. A frame is cut into pixel BATCHES (sampler tables are built once per batch
//      for all spp) and each batch into PASSES of n_samples consecutive samples of every batch pixel.
//      SoA arrays are indexed by path id = local_sample * n_pixels + batch_pixel, so that the 64 lanes of a
//      wave hold the same sample of 64 neighbouring pixels and every access coalesces.
// Round 4: the records of a path TRAVEL WITH ITS QUEUE ENTRY. A bounce's producer (k_raygen, or k_shade of the previous bounce) writes the ray, the throughput and
// the packed state of every path it appends at the path's SLOT in the sharded queue (slot = shard * shard_cap + position in the shard), into planar 16-byte
// arrays; the bounce's consumers - the closest-hit kernel, k_shade - walk the queue's entries in order and read slot after slot: every load of a wave is one
// contiguous 1 KB stream issued without a dependent id fetch, paths that ended leave no holes, and the order rays are appended in (e.g. by direction octant)
// costs the next shade launch nothing. Two generations (PathGen) alternate between bounces. What is addressed by PATH ID is only what outlives the bounce:
// the radiance sum `lacc` (added to by the any-hit epilogue, k_resolve and - for emitters seen directly - k_shade) and the film position `pfilm`.
// Before: RayRec / VertRec / PathAcc arrays indexed by path id and queues of ids, so a consumer's loads were a gather through the id (two dependent round
// trips, lines shared with dead paths; S1 k_shade<1> moved 1.7x its algorithmic bytes at 0.68 wait cycles per wave cycle - VERDICT r03 weak #3).
// ---- Cache policy of the once-through streams (round 6). With RT_NT_STREAMS (the default; -DRT_NT_STREAMS=0: SPtr<T> is T*) SPtr<T> is a pointer whose element reads and
// writes carry the non-temporal hint (global_load / global_store ... nt): the travelling path records, ray and hit records, ray queues and occlusion bytes are written
// once and read once, and should not push the gathered tables (environment rows, texels, triangle records) out of L2 / Infinity Cache. Same values either way; measured in
// MEASUREMENTS R6 (scripts/micro/fetch_calibrate.hip has the same mix in isolation).
#ifndef RT_NT_STREAMS
#define RT_NT_STREAMS 1
#endif
#if RT_NT_STREAMS
typedef float nt_v4f __attribute__((ext_vector_type(4)));
typedef unsigned nt_v4u __attribute__((ext_vector_type(4)));
typedef float nt_v2f __attribute__((ext_vector_type(2)));
RT_DEV float4 nt_ld(const float4* p) { const nt_v4f v = __builtin_nontemporal_load((const nt_v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
RT_DEV uint4 nt_ld(const uint4* p) { const nt_v4u v = __builtin_nontemporal_load((const nt_v4u*)p); return make_uint4(v.x, v.y, v.z, v.w); }
RT_DEV float2 nt_ld(const float2* p) { const nt_v2f v = __builtin_nontemporal_load((const nt_v2f*)p); return make_float2(v.x, v.y); }
RT_DEV unsigned nt_ld(const unsigned* p) { return __builtin_nontemporal_load(p); }
RT_DEV unsigned char nt_ld(const unsigned char* p) { return __builtin_nontemporal_load(p); }
RT_DEV void nt_st(float4 a, float4* p) { nt_v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, (nt_v4f*)p); }
RT_DEV void nt_st(uint4 a, uint4* p) { nt_v4u v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, (nt_v4u*)p); }
RT_DEV void nt_st(float2 a, float2* p) { nt_v2f v; v.x = a.x; v.y = a.y; __builtin_nontemporal_store(v, (nt_v2f*)p); }
RT_DEV void nt_st(unsigned a, unsigned* p) { __builtin_nontemporal_store(a, p); }
RT_DEV void nt_st(unsigned char a, unsigned char* p) { __builtin_nontemporal_store(a, p); }
template <class T> struct NtRef {
  T* p;
  RT_DEV operator T() const { return nt_ld((const T*)p); }
  RT_DEV void operator=(const T& v) const { nt_st(v, p); }
};
template <class T> struct NtPtr {
  typedef typename std::remove_const<T>::type V;
  T* p;
  __host__ __device__ NtPtr() : p(nullptr) {}
  __host__ __device__ NtPtr(T* q) : p(q) {}
  template <class U> __host__ __device__ NtPtr(const NtPtr<U>& q) : p(q.p) {}  // NtPtr<T> -> NtPtr<const T>
  RT_DEV typename std::conditional<std::is_const<T>::value, V, NtRef<V>>::type operator[](size_t i) const { if constexpr (std::is_const<T>::value) return nt_ld(p + i); else return NtRef<V>{p + i}; }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
  __host__ __device__ bool operator!=(std::nullptr_t) const { return p != nullptr; }
  __host__ __device__ bool operator==(std::nullptr_t) const { return p == nullptr; }
};
template <class T> using SPtr = NtPtr<T>;
#define RT_SPTR_R(T) NtPtr<T>
template <class T> __host__ __device__ T* sraw(NtPtr<T> q) { return q.p; }
// sp<NT>(q): the stream pointer as a kernel wants it - with the hint (q itself) or without (the plain pointer)
template <bool NT, class T> RT_DEV typename std::conditional<NT, NtPtr<T>, T*>::type sp(NtPtr<T> q) { if constexpr (NT) return q; else return q.p; }
#else
template <class T> using SPtr = T*;
#define RT_SPTR_R(T) T* __restrict__
template <class T> __host__ __device__ T* sraw(T* q) { return q; }
template <bool NT, class T> RT_DEV T* sp(T* q) { return q; }
#endif
struct PathGen { SPtr<float4> o; SPtr<float4> d; SPtr<float4> beta; SPtr<uint4> st; };  // [slot]: ray (o | t_max), (d | -), throughput (rgb | eta_scale), (packed state, path id, RNG state lo, hi)
// The records of a vertex's shadow ray and BSDF-sampled MIS ray, PLANAR (round 4; 64- and 128-byte structs with 16 + 44 unused bytes before): a shade wave's
// stores are contiguous 1 KB runs per field, the trace kernels read the two ray fields and nothing else, k_resolve reads what it needs of a vertex that
// contributes. 48 B per shadow ray, 100 B per MIS ray.
struct ShadowPlanes { SPtr<float4> o; SPtr<float4> d; SPtr<float4> add; };  // (o | t_max), (d | complete-here flag << 31 | path id), beta * Ld / pick_pdf
struct MisPlanes { SPtr<float4> o; SPtr<float4> d; SPtr<float4> hit; SPtr<float4> a; SPtr<float4> b; SPtr<float4> c; SPtr<unsigned> flags; };  // d.w = path id; hit: of closest-hit MIS rays
// 128 B, one line: everything k_resolve needs of a vertex with a BSDF-sampled MIS ray. hit.y = prim of the closest hit, or - for rays that only
// need occlusion (sampled light infinite) - 1 / 0 from the any-hit kernel
#define RT_PEND_SHADOW 1u           // MisPlanes::flags: a shadow ray is out (its result is in occ_sh)
#define RT_PEND_MIS_ANY 0x40000000u // the MIS ray was traced for occlusion only; bits 2-29: index of the sampled light
struct PassState {
  unsigned cap;            // paths in this pass = n_pixels * n_samples
  unsigned spp, spp_log2, dims;
  unsigned n_pixels;       // batch pixels
  unsigned n_pixels_recip; // floor(2^32 / n_pixels) (2^32 - 1 for one pixel): path id -> (sample, pixel) by a multiply (split_path_id)
  unsigned s0, n_samples;  // this pass renders samples [s0, s0 + n_samples) of every batch pixel
  float4* own_acc;         // [batch pixel] running (R, G, B, weight) sum of the pixel's own samples across passes
  // sampler tables of the chunk
  const unsigned* scrambles;        // [pixel][3*dims]
  const unsigned short* perms;      // [pixel][2*dims][spp]
  PathGen in, out;         // records of the paths of this bounce (read) and of the next (written), by queue slot
  SPtr<float4> hit;        // [slot] closest hit of the bounce's ray: (b2, prim, b0, b1)
  float4* lacc;            // [path id] (L rgb | flags: RT_STATE_OUT_OF_BOUNDS)
  SPtr<float2> pfilm;      // [path id] film position of the camera sample
  // The records of a vertex's shadow ray and BSDF-sampled MIS ray are indexed by the vertex's POSITION IN THE SHADE LAUNCH'S QUEUE (entry i of the - possibly
  // material-sorted - queue; distinct for every vertex of a bounce), not by its path id: the 64 lanes of a shade wave then write 64 neighbouring records, the
  // ray queues (which hold these indices) list them in nearly ascending order for the trace kernels and k_resolve, and occ_sh / occ_mi below are indexed the
  // same way. On a material-sorted queue path ids are scattered; 192 B of records per vertex written and re-read by id were a quarter of S4's memory traffic.
  // Each record carries its path: ShadowRec::d.w = (no MIS ray in flight: complete at the any-hit epilogue) << 31 | path id, MisRec::d.w = path id.
  ShadowPlanes sh; MisPlanes mi;
  // one byte per path each: the shadow ray of a vertex that also has an MIS ray in flight was blocked / its occlusion-only MIS ray was blocked. Dense, so
  // that k_resolve learns from two bytes that a vertex contributes nothing (most vertices of an interior) without touching its 128-byte MisRec
  unsigned char* occ_sh; unsigned char* occ_mi;
  // q_in: NULL = the bounce's entries are the slots of the sharded queue themselves (cnt_in: shard counts; also NULL: bounce 0 of a pass whose samples are all
  // traced - entry i is slot i is path i), or the material-sorted list of slots (k_bin_scatter) as a one-shard queue. q_out is gone: the out records ARE the queue.
  // Ray queues (shadow, mis, mis-any) hold record indices = positions in the shade launch's queue, each split into RT_QSHARDS shards (shard = blockIdx & 7 of the producer, region
  // [shard * shard_cap, ...)) with its own counter word: a single word sustains only ~88 returning
  // atomics per microsecond. Every bounce has its own zero-initialised block of counters, so nothing has to be
  // reset or rotated between bounces: cnt_in = the 8 shard counts of q_in (written by raygen / the previous
  // bounce), cnt_out[q * RT_QSHARDS + shard] with q: 0 = continuing paths (q_out), 1 = shadow, 2 = mis (closest hit),
  // 3 = mis rays that only need occlusion.
  const unsigned* q_in; unsigned* q_shadow; unsigned* q_mis; unsigned* q_misany;
  const unsigned* cnt_in; unsigned* cnt_out; unsigned shard_cap;
  int all_in_bounds;  // every sample of the pass is traced (no crop by pixel_bounds): bounce 0 needs no queue, path i is entry i
  // ... and then bounce 0's throughput and state records are known without being stored: beta = (1, 1, 1 | eta_scale 1), st = (pack_state(0, false, 1, 2), path id = slot,
  // the keyed RNG stream's first state). fresh: bit 0 - k_raygen does not write the throughput record and the launches of bounce 0 do not read it, bit 1 - the same for the
  // state record (16 + 16 of the 72 bytes a camera sample wrote and of the 80 its first vertex read; round 6, end). During k_raygen it names what the frame leaves out;
  // in the bounce loop it is zero from bounce 1 on.
  int fresh;
#ifndef RT_FRESH_RECORDS
#define RT_FRESH_RECORDS 3      // which of the two records a frame leaves out (0: the A/B control - every record written and read)
#endif
#ifndef RT_FRESH_RECORDS_LDS
#define RT_FRESH_RECORDS_LDS 2  // ... a frame of k_shade<1, .., LDSREC = 1> (S1): the state record only. One box, interleaved, three rounds (profiles/r06_ab_fresh_records.txt):
#endif                          // S1 0 / 1 / 2 / 3 = 2003 / 1978 / 2020 / 1977 Msamples/s (a constant throughput costs that kernel 12 ms of shade); the other forms
                                // 0 / 1 / 2 / 3: S2 1509 / 1533 / 1523 / 1556, S3 1816 / 1825 / 1828 / 1842, S4 447.0 / 445.3 / 448.2 / 446.6
#define RT_FRESH_BETA(ps) (((ps).fresh & 1) != 0)
#define RT_FRESH_ST(ps) (((ps).fresh & 2) != 0)
  int mis_any;        // BSDF-sampled MIS rays toward an infinite light go to q_misany (off on frames that count node visits: reference walk)
  // Sphere::pdf_wi answers with the cone's uniform density for ANY direction (sphere.rs:310-334 never tests wi against the cone), so estimate_direct casts the
  // BSDF-sampled ray of every vertex whose picked light is a sphere - 150x the MIS rays of the tessellated S3 - and drops all that do not end on the sphere. A ray
  // that misses the sphere's (slightly widened) world box cannot end on it: with this flag such a ray is not cast (its term is the zero it would have been);
  // it is counted in ST_MIS_UNREACHED and stays in rt_stats::rays_mis. Off on frames that count the reference's walk.
  int skip_unreachable_mis;
  // PathIntegrator::li traces a path's next ray BEFORE it tests the depth (path.rs:100-137: intersect, emitted light if bounces == 0 or after a specular
  // bounce, then `if !found || bounces >= max_depth { break }`): the ray cast at bounces == max_depth is read only after a specular bounce. With this flag a
  // path whose last bounce was not specular ends there (counted in ST_TAIL_UNCAST, part of rt_stats::rays_closest): same film, no trace, no vertex. Off on
  // frames that count the reference's walk.
  int skip_dead_tail;
  const unsigned* range;  // k_shade: shade entries [range[0], range[1]) of q_in only (NULL = all): class-wise dispatch over the binned queue
  unsigned long long* stats;  // device-side u64 counters, see ST_* below
};
#define RT_NQ 4  // queues a bounce fills
enum { ST_CAMERA = 0, ST_RAYS_CLOSEST, ST_RAYS_SHADOW, ST_RAYS_MIS, ST_NODES_CLOSEST, ST_NODES_SHADOW, ST_NODES_MIS,
       ST_TRIS_CLOSEST, ST_TRIS_SHADOW, ST_TRIS_MIS, ST_SCRUBBED, ST_UNBUILT_VOXEL,
       ST_RAYS_MISANY, ST_NODES_MISANY, ST_TRIS_MISANY,  // the occlusion-only MIS rays (also counted in the _MIS entries)
       ST_MIS_UNREACHED,  // MIS rays toward a quadric light that miss its box: not cast (PassState::skip_unreachable_mis)
       ST_TAIL_UNCAST,    // path rays at the depth limit after a non-specular bounce: not cast (PassState::skip_dead_tail)
       ST_SHADED,  // + {0: k_shade<1>, 1: k_shade<3>, 2: k_shade<5>, 3: k_shade<0>}: path vertices shaded by each front-end (misses included)
       ST_STAMP = ST_SHADED + 4,  // + 8 * front-end + section: wave cycles of the sections of k_shade (measurement builds only, make ABLATE=1)
       ST_COUNT = ST_STAMP + 32 };
#ifdef RT_ABLATE
#define RT_STAMP(k) do { const unsigned long long t_ = clock64(); stamp_acc[k] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define RT_STAMP(k) do { } while (0)
#endif

// path id = local sample * n_pixels + batch pixel. q = mulhi(pid, floor(2^32 / n)) is floor(pid / n) or one less: one compare fixes it (an integer division is
// ~30 instructions, twice per path vertex otherwise).
RT_DEV void split_path_id(const PassState& ps, unsigned pid, unsigned& sl, unsigned& pix) {
  unsigned q = __umulhi(pid, ps.n_pixels_recip), r = pid - q * ps.n_pixels;
  if (r >= ps.n_pixels) { q += 1u; r -= ps.n_pixels; }
  sl = q; pix = r;
}

struct FrameParams {
  // camera (rc/camera.rs)
  float r2c[16]; float c2w[16]; f3 dx_camera, dy_camera; float lens_radius, focal_distance;
  // film (rc/film.rs)
  int crop_x0, crop_y0, crop_x1, crop_y1; int sb_x0, sb_y0, sb_x1, sb_y1;
  float radius_x, radius_y, max_sample_luminance;
  // integrator
  int max_depth; float rr_threshold; int pb_x0, pb_y0, pb_x1, pb_y1;
  // sharding: owned sample rows j -> y = sb_y0 + (((j >> L) * world + rank) << L) + (j & (2^L - 1)), L = shard_log2 (bands of RT_SHARD_ROWS rows)
  int rank, world, shard_log2;
  // pass
  unsigned long long chunk_first;  // first owned-pixel index of this batch
  unsigned w_recip;                // floor(2^32 / W), W = sample-bounds width (0 when W == 1): division by W without a divide
};

// packed per-path state (PathGen::st.x): bits 0-7 bounces, bit 8 specular_bounce, bits 9-12 cur_1d, bits 13-16 cur_2d. lacc.w, bit 17: sample outside
// pixel_bounds (never traced, skipped by the film)
#define RT_STATE_OUT_OF_BOUNDS (1u << 17)
RT_DEV unsigned pack_state(int bounces, bool spec, int c1, int c2) { return (unsigned)bounces | ((unsigned)spec << 8) | ((unsigned)c1 << 9) | ((unsigned)c2 << 13); }

RT_DEV unsigned wave_push(unsigned* counter, bool pred) {
  unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return 0u;
  unsigned lane = __lane_id();
  int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
  base = __shfl(base, leader);
  return base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
}

#define RT_QSHARDS 8
// Measured in round 4 and not kept: an XCD-aware chunk order (workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MB L2; with chunk =
// (b % 8) * (grid / 8) + b / 8 the blocks of one XCD take one contiguous eighth of a launch's queue instead of every eighth chunk) for k_shade and for the waves of
// the persistent trace kernels: S4 shade 3136 -> 3155 / 3161 ms, closest hit 1521 -> 1514; S3, S2, S1 within noise. The shade kernels' misses go to tables of
// 16 - 64 MB (environment rows, texels, triangle records) that no 4 MB L2 holds however its blocks are chosen.
// Words between two counters that receive atomics (the shard counters of a queue, the bin cursors of the counting sort). Packed (1), the 24 - 32 shard counters
// a bounce adds to sit in ONE 128-byte line, and atomics on one line are serviced one after the other whichever word they name: every workgroup of a shade
// launch waits on that line three times per iteration (measured: 64-lane workgroups, four times the atomics, made S1's shade launch 1.9x slower). 64 words =
// 256 bytes puts every counter on a line - and a memory channel - of its own.
#ifndef RT_CNT_STRIDE
#define RT_CNT_STRIDE 64
#endif
// Consumer view of a sharded queue: entry i of the concatenation of the shards' filled prefixes. ids == NULL: the entries are the queue's SLOTS
// themselves (the travelling path records live at their slot, PassState::in) - get(i) is then arithmetic on the eight counts, no load.
struct QView {
  const unsigned* ids; unsigned pre[RT_QSHARDS + 1]; unsigned shard_cap;
  RT_DEV void init(const unsigned* ids_, const unsigned* counts, unsigned cap) {
    ids = ids_; shard_cap = cap; pre[0] = 0;
#pragma unroll
    for (int k = 0; k < RT_QSHARDS; ++k) pre[k + 1] = pre[k] + counts[k * RT_CNT_STRIDE];
  }
  RT_DEV unsigned total() const { return pre[RT_QSHARDS]; }
  RT_DEV unsigned get(unsigned i) const {
    unsigned k = 0, base = 0;  // pre[k] selected on the way: indexed afterwards, the prefix array would live in scratch
#pragma unroll
    for (int j = 1; j < RT_QSHARDS; ++j) { const bool ge = i >= pre[j]; k += ge ? 1u : 0u; base = ge ? pre[j] : base; }
    const unsigned slot = k * shard_cap + (i - base);
    // (measured, no effect on any stage: the shard of the wave's first entry found on the scalar unit and shared by the lanes inside it)
    return ids ? ids[slot] : slot;
  }
};
// Round 6 measured the continuing paths GROUPED BY THE DIRECTION OCTANT of their rays (an LDS-resident scene's closest-hit walk reads one link row per octant and meets its
// leaves in one order per octant, rc/bvh/mod.rs:366-433) - VERDICT r05 item 2, twice: as runs of ONE octant across a workgroup's iterations (commit 1c259ca: every wave of the
// next bounce holds one octant; unfilled slots as dead entries) and as octant order inside one iteration's run of 256 slots (two or three octants per wave). S1, one box,
// interleaved: runs - closest hit 184 -> 168 ms, occlusion rays 90 -> 114, shade 220 -> 236; order inside the run - closest hit 189 -> 184, occlusion rays 94 -> 100 - 117,
// shade +1 - 4. What the order costs is what is still addressed by PATH ID: the any-hit epilogue's read-modify-write of lacc[path] (1.6 G per S1 frame) touches 8x / 4x the
// lines once the paths of a wave are no longer neighbours (1.6 G x 2 x 64 B = 26 ms at 8 TB/s: the loss), and a shade wave's record stores go to eight runs. Not kept
// (MEASUREMENTS R6); S3 / mis-spheres +0.3 / +1.6 %.
// Block-aggregated append to up to three sharded device queues: one returning atomic per queue per
// 256-lane iteration, on the counter word of this block's shard. Must be reached by every thread of
// the block. Returns the absolute slot for each queue (valid where pred).
template <int NQ>
RT_DEV void block_push(unsigned* counters, unsigned shard_cap, const int* queue_idx, const bool* pred, unsigned* slot) {
  __shared__ unsigned s_cnt[NQ][16];
  __shared__ unsigned s_base[NQ];
  const unsigned lane = __lane_id(), wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
  const unsigned shard = blockIdx.x & (RT_QSHARDS - 1);
  unsigned long long mask[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    mask[q] = __ballot(pred[q]);
    if (lane == 0) s_cnt[q][wave] = (unsigned)__popcll(mask[q]);
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    unsigned total = 0;
    for (unsigned w = 0; w < n_waves; ++w) { unsigned c = s_cnt[threadIdx.x][w]; s_cnt[threadIdx.x][w] = total; total += c; }
    s_base[threadIdx.x] = shard * shard_cap + (total ? atomicAdd(&counters[(queue_idx[threadIdx.x] * RT_QSHARDS + shard) * RT_CNT_STRIDE], total) : 0u);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) slot[q] = s_base[q] + s_cnt[q][wave] + (unsigned)__popcll(mask[q] & ((1ull << lane) - 1ull));
  __syncthreads();
}

// owned-pixel index -> raster pixel (x, y) and keyed pixel index inside the sample bounds
RT_DEV void owned_pixel(const FrameParams& fp, unsigned long long k, int& x, int& y, unsigned long long& pixel_index) {
  const unsigned W = (unsigned)(fp.sb_x1 - fp.sb_x0);
  unsigned long long j; unsigned xi;
  if (k <= 0xffffffffull && fp.w_recip != 0u) {
    // k / W by a multiply: q = mulhi(k, floor(2^32 / W)) is floor(k / W) or one less (k < 2^32), fixed by one compare
    const unsigned k32 = (unsigned)k;
    unsigned q = __umulhi(k32, fp.w_recip), r = k32 - q * W;
    if (r >= W) { q += 1u; r -= W; }
    j = q; xi = r;
  } else { j = k / W; xi = (unsigned)(k - j * W); }
  const unsigned L = (unsigned)fp.shard_log2;
  unsigned long long row = (((j >> L) * (unsigned long long)fp.world + (unsigned long long)fp.rank) << L) + (j & ((1ull << L) - 1ull));
  x = fp.sb_x0 + (int)xi;
  y = fp.sb_y0 + (int)row;
  pixel_index = row * W + xi;
}

// ================================================================================ K0 sampler tables
// ZeroTwoSequence::start_pixel (rc/sampler/zerotwosequence.rs:67-108) in keyed mode. Per pixel the reference draws,
// for each of the 2*dims tables in turn and from ONE PCG32 stream: 1 (van_der_corput, lowdiscrepancy.rs:10) or 2
// (sobol_2d, :31) scrambles, spp draws uniform_u32_bounded(1) for the per-pixel-sample shuffles of one element
// (:14-21), and spp draws bounded(spp - i) of the Fisher-Yates shuffle of the sample order (:22, :114-124). The
// (0,2) values are a closed form of (scramble, shuffled index) and are evaluated by the consumers (table_1d/2d).
//
// The stream is cut at fixed positions: a bounded draw retries only when r < (!b+1)&b = lowest set bit of b
// (rng.rs:32-40), which over a whole pixel happens with probability ~1e-5. So:
//   K0a  one lane per (pixel, table, segment of <= 256 draws): PCG32 jump-ahead to the segment's nominal stream
//        position (state = A^n * s0 + S_n * inc, (A^n, S_n) from a host-built table), then sequential draws;
//        the shuffle's swap partners other_i = i + r % (spp - i) go to a scratch array [table][pixel][i]; any
//        retry marks the pixel dirty. The divisor is wave-uniform: r % b by a multiply with floor(2^32 / b).
//   K0c  the (rare) dirty pixels' draws are redone in stream order, one lane per pixel.
//   K0b  one lane per (pixel, table) replays the swaps on a permutation kept in LDS ([lane][spp+2] u16, odd word
//        stride => conflict-free); a dependent LDS chain, so blocks are small to put a wave on every SIMD.
// Output, pixel-minor so that a wave's accesses coalesce: scrambles[k][pixel], perms[table][sample][pixel].
struct SamplerSeg { unsigned long long mul, sum; unsigned table, half, i0, n_scr; };  // half 0: consume draws, 1: swap draws
#define RT_DIRTY_CAP 1024u

RT_DEV unsigned long long pixel_index_of(const FrameParams& fp, unsigned pix, unsigned long long explicit_pixel0, int use_explicit) {
  if (use_explicit) return explicit_pixel0 + pix;
  int x, y; unsigned long long pixel_index; owned_pixel(fp, fp.chunk_first + pix, x, y, pixel_index);
  return pixel_index;
}
RT_DEV void mark_dirty(unsigned* dirty, unsigned pix) {  // dirty[0] = count, dirty[1..] = pixels (duplicates are harmless)
  unsigned k = atomicAdd(&dirty[0], 1u);
  if (k < RT_DIRTY_CAP) dirty[1 + k] = pix;
}

static __global__ void __launch_bounds__(256) k_sampler_draws(FrameParams fp, unsigned n_pixels, unsigned spp, unsigned dims, unsigned seg_len,
                                                       unsigned long long explicit_pixel0, int use_explicit, const SamplerSeg* __restrict__ segs,
                                                       const unsigned* __restrict__ magic, unsigned* __restrict__ scrambles, unsigned short* __restrict__ partners,
                                                       unsigned* dirty, int chain_major) {
  const unsigned pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n_pixels) return;
  const SamplerSeg sg = segs[blockIdx.y];
  const unsigned long long pixel_index = pixel_index_of(fp, pix, explicit_pixel0, use_explicit);
  Pcg32 rng;
  rng.inc = (pixel_index << 1u) | 1ull;
  const unsigned long long s0 = (rng.inc + 0x853c49e6748fea9bULL) * 0x5851f42d4c957f2dULL + rng.inc;  // state after set_sequence
  rng.state = sg.mul * s0 + sg.sum * rng.inc;
  const unsigned t = sg.table;
  if (sg.n_scr == 1u) scrambles[(size_t)t * n_pixels + pix] = rng.next_u32();
  else if (sg.n_scr == 2u) {
    unsigned a = rng.next_u32(), b = rng.next_u32();
    scrambles[(size_t)(dims + 2u * (t - dims)) * n_pixels + pix] = a;
    scrambles[(size_t)(dims + 2u * (t - dims) + 1u) * n_pixels + pix] = b;
  }
  bool retry = false;
  if (sg.half == 0u) {
    // bounded(1): threshold 1, retried iff the draw is 0; the output function maps exactly xorshifted == 0 to 0
    for (unsigned i = 0; i < seg_len; ++i) {
      unsigned long long old = rng.state;
      rng.state = old * 0x5851f42d4c957f2dULL + rng.inc;
      retry |= (unsigned)(((old >> 18u) ^ old) >> 27u) == 0u;
    }
  } else {
    // partners[table][i][pixel]: a wave's stores coalesce. chain_major (the parallel replay, k_sampler_shuffle_par): partners[table][pixel][i] - a lane
    // writes its segment as 16-byte stores of eight partners each; a wave's lanes work on 64 lines of their own for the whole segment, which L2 merges
    unsigned short* dst = chain_major ? partners + ((size_t)t * n_pixels + pix) * spp + sg.i0 : partners + ((size_t)t * spp + sg.i0) * n_pixels + pix;
    unsigned w8[4] = {0u, 0u, 0u, 0u};
    for (unsigned k = 0; k < seg_len; ++k) {
      const unsigned i = sg.i0 + k, b = spp - i;  // wave-uniform
      const unsigned r = rng.next_u32();
      retry |= r < ((~b + 1u) & b);
      unsigned rem = 0u;
      if (b > 1u) {
        const unsigned q = __umulhi(r, magic[b]);  // floor(r / b) or one less
        rem = r - q * b;
        rem = rem >= b ? rem - b : rem;
      }
      if (!chain_major) dst[(size_t)k * n_pixels] = (unsigned short)(i + rem);
      else {  // (seg_len is a multiple of 8 wherever chain_major is used: spp >= 64)
        const unsigned v = i + rem, h = (k & 7u) >> 1;
        const unsigned add = (k & 1u) ? (v << 16) : v;
        w8[0] = h == 0u ? (w8[0] | add) : w8[0]; w8[1] = h == 1u ? (w8[1] | add) : w8[1]; w8[2] = h == 2u ? (w8[2] | add) : w8[2]; w8[3] = h == 3u ? (w8[3] | add) : w8[3];
        if ((k & 7u) == 7u) { *(uint4*)(dst + (k - 7u)) = make_uint4(w8[0], w8[1], w8[2], w8[3]); w8[0] = w8[1] = w8[2] = w8[3] = 0u; }
      }
    }
  }
  if (retry) mark_dirty(dirty, pix);
}

// K0b. Step i of the shuffle: swap(a[i], a[other_i]). The LDS round trip of one step would serialise the chain,
// so G steps are issued together: all 2G reads go out at once and the values are then corrected in registers for
// the writes of the earlier steps of the same group (a later step may read a slot an earlier one has just
// overwritten). a[i] is final after step i, so the permutation is flushed from LDS once at the end.
// LDS layout of K0b: lane l keeps its permutation in [l * spp, (l + 1) * spp) u16 with the index XOR-ed by
// (2 l) mod spp: equal indices of different lanes then fall into different banks although the lane stride is a
// power of two, and a block of 16384 / spp lanes fills exactly 32 KB (five blocks per CU).
struct PermLds {
  unsigned short* base; unsigned swz;
  RT_DEV unsigned get(unsigned i) const { return base[i ^ swz]; }
  RT_DEV void set(unsigned i, unsigned v) const { base[i ^ swz] = (unsigned short)v; }
};
RT_DEV PermLds perm_lds(unsigned short* lds, unsigned lane, unsigned spp) { PermLds p; p.base = lds + (size_t)lane * spp; p.swz = (lane << 1) & (spp - 1u); return p; }

template <int G>
RT_DEV void shuffle_group(const PermLds& a, unsigned i, const unsigned* o) {
  unsigned va[G], vb[G];  // 32-bit registers: 16-bit VALU operands cost pack/unpack instructions
#pragma unroll
  for (int k = 0; k < G; ++k) { va[k] = a.get(i + k); vb[k] = a.get(o[k]); }
#pragma unroll
  for (int k = 0; k < G; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      if (o[j] == i + k) va[k] = va[j];
      if (o[j] == o[k]) vb[k] = va[j];
    }
    a.set(o[k], va[k]);
    a.set(i + k, vb[k]);  // after the write above: if other_k == i + k both hold the same value
  }
}
RT_DEV void unpack8(uint4 v, unsigned* o) {
  o[0] = v.x & 0xffffu; o[1] = v.x >> 16; o[2] = v.y & 0xffffu; o[3] = v.y >> 16;
  o[4] = v.z & 0xffffu; o[5] = v.z >> 16; o[6] = v.w & 0xffffu; o[7] = v.w >> 16;
}
// blockDim.x = L lanes (one wave at most), blockIdx.y = table
// `tables`: the launch's tables, 4 bits each (blockIdx.y-th nibble): a frame builds its tables in the order it needs them (rt_render)
static __global__ void __launch_bounds__(64) k_sampler_shuffle(unsigned n_pixels, unsigned spp, const unsigned short* __restrict__ partners, unsigned short* __restrict__ perms, unsigned long long tables) {
  const unsigned table = (unsigned)(tables >> (4u * blockIdx.y)) & 15u;
  extern __shared__ unsigned short lds_perm[];
  const unsigned lane = threadIdx.x, L = blockDim.x;
  const unsigned pix0 = blockIdx.x * L, pix = pix0 + lane;
  const PermLds a = perm_lds(lds_perm, lane, spp);
  if (pix < n_pixels) {
    const unsigned short* in = partners + (size_t)table * spp * n_pixels + pix;
    for (unsigned i = 0; i < spp; ++i) a.set(i, i);
    constexpr int G = 8;
    if (spp >= 4u * G) {
      // partners are fetched three groups ahead (global latency >> one group) into a ring of four register
      // buffers; the loop is unrolled over the ring so that it needs no register moves
      unsigned b0[G], b1[G], b2[G], b3[G];
      auto fetch = [&](unsigned* dst, unsigned first) {
        const unsigned f = first < spp ? first : 0u;  // past the end: a harmless re-read
#pragma unroll
        for (int k = 0; k < G; ++k) dst[k] = in[(size_t)(f + k) * n_pixels];
      };
      fetch(b0, 0u); fetch(b1, G); fetch(b2, 2u * G);
      for (unsigned i = 0; i < spp; i += 4u * G) {
        fetch(b3, i + 3u * G); shuffle_group<G>(a, i, b0);
        fetch(b0, i + 4u * G); shuffle_group<G>(a, i + G, b1);
        fetch(b1, i + 5u * G); shuffle_group<G>(a, i + 2u * G, b2);
        fetch(b2, i + 6u * G); shuffle_group<G>(a, i + 3u * G, b3);
      }
    } else {
      for (unsigned i = 0; i < spp; ++i) {
        unsigned o = in[(size_t)i * n_pixels];
        unsigned va = a.get(i), vb = a.get(o);
        a.set(i, vb); a.set(o, va);
      }
    }
  }
  __syncthreads();
  // flush to perms[table][sample][pixel]. A vector memory instruction costs the same issue slot whatever it
  // carries, so each lane gathers one sample of 8 neighbouring pixels from LDS and stores 16 bytes.
  unsigned short* out = perms + (size_t)table * spp * n_pixels;
  if ((n_pixels & 7u) == 0u && L >= 8u) {
    const unsigned groups = L >> 3, g = lane % groups, r0 = lane / groups;  // 8 rows per pass
    const unsigned gp = pix0 + 8u * g;
    if (gp < n_pixels) {
      PermLds c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = perm_lds(lds_perm, 8u * g + (unsigned)j, spp);
      for (unsigned i = r0; i < spp; i += 8u) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = c[2 * j].get(i) | (c[2 * j + 1].get(i) << 16);
        *(uint4*)(out + (size_t)i * n_pixels + gp) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  } else if (pix < n_pixels) {
    for (unsigned i = 0; i < spp; ++i) out[(size_t)i * n_pixels + pix] = (unsigned short)a.get(i);
  }
}

// K0b', the exact PARALLEL replay of the Fisher-Yates chain (round 4). k_sampler_shuffle walks a chain of spp dependent swaps with one
// lane and a 2 * spp-byte permutation in LDS: 64 chains fill 128 KB, so a CU holds one such wave and runs it at the latency of one LDS round trip per eight
// swaps - 57 ms per S1 frame alone, and while it overlaps path kernels its 128 KB workgroups evict theirs. Here ONE WAVE replays ONE chain, 64 steps at a time:
//   step i does swap(a[i], a[o_i]), o_i >= i. Position p only changes when a step w < p with o_w = p (a "writer" of p) puts there what a[w] held before step w,
//   R(w); at step p itself a[p] becomes final. So, with p's writers in order w_1 < w_2 < ... < w_k (the self-swap w = p, if any, last):
//       perm[w_1] = p,   perm[w_{m+1}] = R(w_m),   R(j) = j if j has no writer below j, else R(L(j)) with L(j) = the last writer of j below j.
//   Every step is a writer of exactly one position, so this names every perm[i] once. The writers are grouped by target with a counting sort in LDS (packed
//   16-bit counters, one atomic per step; the groups come out in ascending order, which is checked), L is read off the groups and R is found by pointer
//   jumping (a chain of L links is ~2 long: 3 - 5 rounds). Everything is straight-line code over the lane's E steps: a dozen dependent LDS round trips per
//   chain instead of 1024. (A first version with per-target loops - insertion sort, pointer chase - took 131 ms per S1 frame: loops of dependent LDS reads.) The result goes through an LDS tile so that 16 chains leave as 32-byte runs of perms[table][sample][pixel].
// MEASURED (S1's batches, 524 288 pixels x 8 tables, alone on the GPU): 15.4 ms against the chain kernel's 23.2 - and 4.4 ms more in k_sampler_draws for the
// chain-major partner layout it needs; under path kernels, where a frame builds all tables but the first batch's, it costs the frame more than the chain
// kernel does at 64 ... 512 spp (rtx_hip.hip, launch_sampler_tables). DEFAULT: used for every batch of a 1024-spp frame (together with a short leading batch: S1 679.5 ->
// 667.5 ms, end of round 4), not used at other sample counts; RTX_K0_PARALLEL=1 forces it on for 64 <= spp <= 1024, =0 forces the chain kernel everywhere.
// Exact: the same permutation as the sequential replay for every partner sequence (tests: the sampler tables stay bit-equal to the oracle's, retry pixels
// included). E = spp / 64 steps per lane: instantiated for spp 64 ... 1024; other sample counts keep k_sampler_shuffle.
#define RT_SHUF_PIX 16
RT_DEV void wave_sync_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
template <int E>
__global__ void __launch_bounds__(256) k_sampler_shuffle_par(unsigned n_pixels, const unsigned short* __restrict__ partners, unsigned short* __restrict__ perms, unsigned* __restrict__ n_resorted, unsigned long long tables, unsigned force_resort) {
  constexpr unsigned N = 64u * E, NONE = 0xffffu;
  __shared__ unsigned short tile[RT_SHUF_PIX][N + 2];  // (+2: rows an odd number of words apart, the transposed read-out is conflict-free)
  __shared__ unsigned short s_o[4][N];                 // per wave: R (the pointer-jumping array)
  __shared__ unsigned s_cnt[4][N / 2 + 64 + 1];        // per wave: packed 16-bit counters per target (target p at index CI(p) = p + 2 * (p / E)), then exclusive bases
  __shared__ unsigned short s_wl[4][N];                // per wave: writers grouped by target
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, t = (unsigned)(tables >> (4u * blockIdx.y)) & 15u, pix0 = blockIdx.x * RT_SHUF_PIX;
  unsigned short* const o = s_o[wv]; unsigned* const cnt = s_cnt[wv]; unsigned short* const wl = s_wl[wv];
  const unsigned short* const cnt16 = (const unsigned short*)cnt;
  // the partners of the workgroup's 16 chains, all requested at once (one memory latency per workgroup, not one per chain): each chain's row of the tile holds
  // its partners until the chain is replayed and its permutation afterwards
  {  // 16-byte loads, all of a thread's in flight together: thread tid takes 16-byte piece tid, tid + 256, ... of the workgroup's 16 x 2 N bytes
    constexpr unsigned PIECES = RT_SHUF_PIX * N * 2u / 16u / 256u;  // per thread (N >= 64: at least 1... N = 64: 128 pieces for 256 threads)
    constexpr unsigned NP = PIECES ? PIECES : 1u;
    uint4 v[NP];
#pragma unroll
    for (unsigned k = 0; k < NP; ++k) {
      const unsigned piece = threadIdx.x + 256u * k, c = piece / (N / 8u), w = piece - c * (N / 8u);
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (piece < RT_SHUF_PIX * (N / 8u) && pix0 + c < n_pixels) v[k] = ((const uint4*)(partners + ((size_t)t * n_pixels + pix0 + c) * N))[w];
    }
#pragma unroll
    for (unsigned k = 0; k < NP; ++k) {
      const unsigned piece = threadIdx.x + 256u * k, c = piece / (N / 8u), w = piece - c * (N / 8u);
      if (piece < RT_SHUF_PIX * (N / 8u)) { unsigned* d = (unsigned*)tile[c] + 4u * w; d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w; }
    }
  }
  __syncthreads();
  for (unsigned c = wv; c < RT_SHUF_PIX; c += 4u) {
    const unsigned pix = pix0 + c;
    if (pix >= n_pixels) break;  // (wave-uniform)
    for (unsigned k = lane; k < N / 2u + 64u; k += 64u) cnt[k] = 0u;
    wave_sync_lds();
    // Step i = 64 m + lane. Its rank among the writers of its target is the value its atomic returns: iterations run in order of m, and the lanes of ONE
    // atomic instruction that name the same counter are served in lane order on this hardware - so the rank is the writer's position in ascending order and
    // the groups come out sorted. That order is not architected: it is CHECKED below (every writer against its predecessor) and a wave that finds it broken
    // sorts its groups the slow way.
    // Every phase below is written in STAGES over the lane's E steps - all loads of a stage first, then what depends on them - so that a stage costs one LDS
    // round trip, not E of them (written step by step the compiler kept the steps' dependent reads in order: 19 500 cycles per chain, all of it LDS latency
    // at the two waves per SIMD the workgroup's 57 KB allow).
    unsigned pk[E], rk[E];
#pragma unroll
    for (int m = 0; m < E; ++m) pk[m] = tile[c][64u * (unsigned)m + lane];
#pragma unroll
    for (int m = 0; m < E; ++m) {
      const unsigned ci = pk[m] + 2u * (pk[m] / (unsigned)E);
      rk[m] = atomicAdd(&cnt[ci >> 1], (pk[m] & 1u) ? 0x10000u : 1u);
    }
#pragma unroll
    for (int m = 0; m < E; ++m) rk[m] = (pk[m] & 1u) ? (rk[m] >> 16) : (rk[m] & 0xffffu);
    wave_sync_lds();
    if (force_resort) {  // test knob (RTX_K0_FORCE_RESORT=1): hand every group its ranks in DESCENDING order, what an atomic unit that served lanes the other way round would
                         // return - the check below then fails for every group of two or more writers and the wave takes the re-sort branch (that branch is
                         // "never seen" on this hardware and would otherwise never be exercised)
#pragma unroll
      for (int m = 0; m < E; ++m) rk[m] = (unsigned)cnt16[pk[m] + 2u * (pk[m] / (unsigned)E)] - 1u - rk[m];
    }
    {  // exclusive scan of the counts over the targets: lane l scans its targets [l * E, (l + 1) * E) (CI skews the counters by one word per E targets, so
       // that the 64 lanes' chunks start in different banks), one wave scan joins the lanes' sums
      unsigned x[E], local[E]; unsigned sum = 0u;
#pragma unroll
      for (int m = 0; m < E; ++m) x[m] = cnt16[lane * (E + 2u) + (unsigned)m];
#pragma unroll
      for (int m = 0; m < E; ++m) { local[m] = sum; sum += x[m]; }
      unsigned incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
      const unsigned excl = incl - sum;
      wave_sync_lds();  // every lane has read its counts
#pragma unroll
      for (int m = 0; m < E; ++m) ((unsigned short*)cnt)[lane * (E + 2u) + (unsigned)m] = (unsigned short)(excl + local[m]);
    }
    wave_sync_lds();
    unsigned bs[E];
#pragma unroll
    for (int m = 0; m < E; ++m) bs[m] = cnt16[pk[m] + 2u * (pk[m] / (unsigned)E)];
#pragma unroll
    for (int m = 0; m < E; ++m) wl[bs[m] + rk[m]] = (unsigned short)(64u * (unsigned)m + lane);
    wave_sync_lds();
    unsigned pred[E];  // the writer before step i in its group (sorted: its predecessor in arrival order), NONE for the first
#pragma unroll
    for (int m = 0; m < E; ++m) pred[m] = wl[bs[m] + (rk[m] > 0u ? rk[m] - 1u : 0u)];
    bool unsorted = false;
#pragma unroll
    for (int m = 0; m < E; ++m) { if (rk[m] == 0u) pred[m] = NONE; else unsorted |= pred[m] >= 64u * (unsigned)m + lane; }
    if (__ballot(unsorted) != 0ull) {  // never seen; kept exact: sort every group, then every step finds its place
      if (lane == 0u && n_resorted) atomicAdd(n_resorted, 1u);
#pragma unroll 1
      for (int m = 0; m < E; ++m) {
        const unsigned p = 64u * (unsigned)m + lane, g = cnt16[p + 2u * (p / (unsigned)E)], k = (p + 1u < N ? (unsigned)cnt16[p + 1u + 2u * ((p + 1u) / (unsigned)E)] : N) - g;
        for (unsigned a = 1; a < k; ++a) {
          const unsigned v = wl[g + a]; unsigned b = a;
          while (b > 0u && wl[g + b - 1u] > v) { wl[g + b] = wl[g + b - 1u]; --b; }
          wl[g + b] = (unsigned short)v;
        }
      }
      wave_sync_lds();
#pragma unroll 1
      for (int m = 0; m < E; ++m) { unsigned a = 0; while (wl[bs[m] + a] != 64u * (unsigned)m + lane) ++a; pred[m] = a > 0u ? (unsigned)wl[bs[m] + a - 1u] : NONE; }
    }
    {  // R, first as "the last writer of j below j, or j itself" (its array: o, free once the partners are in registers)
      unsigned g[E], gn[E], last[E], prev[E];
#pragma unroll
      for (int m = 0; m < E; ++m) { const unsigned p = 64u * (unsigned)m + lane; g[m] = cnt16[p + 2u * (p / (unsigned)E)]; gn[m] = p + 1u < N ? (unsigned)cnt16[p + 1u + 2u * ((p + 1u) / (unsigned)E)] : N; }
#pragma unroll
      for (int m = 0; m < E; ++m) { const unsigned k = gn[m] - g[m]; last[m] = wl[k > 0u ? gn[m] - 1u : 0u]; prev[m] = wl[k > 1u ? gn[m] - 2u : 0u]; }
#pragma unroll
      for (int m = 0; m < E; ++m) {
        const unsigned p = 64u * (unsigned)m + lane, k = gn[m] - g[m];
        const unsigned L = (k > 0u && last[m] < p) ? last[m] : ((k > 1u) ? prev[m] : NONE);  // (last == p: the self-swap; the writer before it)
        o[p] = (unsigned short)(L == NONE ? p : L);
      }
    }
    wave_sync_lds();
    // ... then by pointer jumping (in place: an entry only ever moves to an ancestor) until every entry names the start of its chain
    for (;;) {
      unsigned r[E], rr[E];
#pragma unroll
      for (int m = 0; m < E; ++m) r[m] = o[64u * (unsigned)m + lane];
#pragma unroll
      for (int m = 0; m < E; ++m) rr[m] = o[r[m]];
      bool changed = false;
#pragma unroll
      for (int m = 0; m < E; ++m) { changed |= rr[m] != r[m]; o[64u * (unsigned)m + lane] = (unsigned short)rr[m]; }
      wave_sync_lds();
      if (__ballot(changed) == 0ull) break;
    }
    {  // perm[i]: the target's own index for its first writer, else what the writer before it moved there
      unsigned val[E];
#pragma unroll
      for (int m = 0; m < E; ++m) val[m] = o[pred[m] != NONE ? pred[m] : 0u];
#pragma unroll
      for (int m = 0; m < E; ++m) tile[c][64u * (unsigned)m + lane] = (unsigned short)(pred[m] != NONE ? val[m] : pk[m]);
    }
    wave_sync_lds();
  }
  __syncthreads();
  // perms[table][sample][pixel]: a thread gathers sample i of the 16 chains (neighbouring lanes read neighbouring entries of a row) and stores 32 contiguous bytes
  unsigned short* const out = perms + (size_t)t * N * n_pixels;
  if ((n_pixels & 7u) == 0u && pix0 + RT_SHUF_PIX <= n_pixels) {
    for (unsigned i = threadIdx.x; i < N; i += 256u) {
      unsigned w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = (unsigned)tile[2 * q][i] | ((unsigned)tile[2 * q + 1][i] << 16);
      uint4* dst = (uint4*)(out + (size_t)i * n_pixels + pix0);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]); dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  } else {
    for (unsigned i = threadIdx.x; i < N; i += 256u)
      for (unsigned c = 0; c < RT_SHUF_PIX; ++c) if (pix0 + c < n_pixels) out[(size_t)i * n_pixels + pix0 + c] = tile[c][i];
  }
}

// K0c: the swap partners and scrambles of the (rare) pixels whose stream holds a retry, redone from the start of
// the stream in order, one lane per pixel, before K0b consumes them.
static __global__ void __launch_bounds__(64) k_sampler_redo(FrameParams fp, unsigned n_pixels, unsigned spp, unsigned dims,
                                                     unsigned long long explicit_pixel0, int use_explicit, unsigned* dirty,
                                                     const unsigned* __restrict__ magic, unsigned* scrambles, unsigned short* partners, int chain_major) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && dirty[0] > RT_DIRTY_CAP) dirty[1 + RT_DIRTY_CAP] = 1u;  // sticky: the host fails the frame
  const unsigned n = dirty[0] < RT_DIRTY_CAP ? dirty[0] : RT_DIRTY_CAP;
  if (k >= n) return;
  const unsigned pix = dirty[1 + k];
  Pcg32 rng; rng.set_sequence(pixel_index_of(fp, pix, explicit_pixel0, use_explicit));
  for (unsigned t = 0; t < 2u * dims; ++t) {
    if (t < dims) scrambles[(size_t)t * n_pixels + pix] = rng.next_u32();
    else {
      unsigned s0 = rng.next_u32(), s1 = rng.next_u32();
      scrambles[(size_t)(dims + 2u * (t - dims)) * n_pixels + pix] = s0;
      scrambles[(size_t)(dims + 2u * (t - dims) + 1u) * n_pixels + pix] = s1;
    }
    for (unsigned i = 0; i < spp; ++i) (void)rng.bounded(1u);
    unsigned short* dst = chain_major ? partners + ((size_t)t * n_pixels + pix) * spp : partners + (size_t)t * spp * n_pixels + pix;
    const size_t step = chain_major ? 1 : n_pixels;
    for (unsigned i = 0; i < spp; ++i) {
      const unsigned b = spp - i, threshold = (~b + 1u) & b;
      unsigned r; do { r = rng.next_u32(); } while (r < threshold);  // bounded(b), rng.rs:32-40
      unsigned rem = 0u;
      if (b > 1u) { const unsigned q = __umulhi(r, magic[b]); rem = r - q * b; rem = rem >= b ? rem - b : rem; }  // r % b
      dst[(size_t)i * step] = (unsigned short)(i + rem);
    }
  }
}

// The plain statement of the whole algorithm, one lane per pixel (kept as the in-tree cross-check of K0a-c:
// rt_sampler_tables(..., RT_SAMPLER_PLAIN)).
static __global__ void k_sampler_tables(FrameParams fp, unsigned n_pixels, unsigned spp, unsigned dims,
                                 unsigned long long explicit_pixel0, int use_explicit, unsigned* scrambles, unsigned short* perms) {
  extern __shared__ unsigned short lds_perm[];
  const unsigned lane = threadIdx.x;
  const unsigned stride = spp + 2;
  unsigned short* mine = lds_perm + (size_t)lane * stride;
  const unsigned pix = blockIdx.x * blockDim.x + lane;
  if (pix >= n_pixels) return;
  const unsigned long long pixel_index = pixel_index_of(fp, pix, explicit_pixel0, use_explicit);
  Pcg32 rng; rng.set_sequence(pixel_index);
  for (unsigned t = 0; t < 2u * dims; ++t) {
    if (t < dims) scrambles[(size_t)t * n_pixels + pix] = rng.next_u32();  // van_der_corput scramble (:10)
    else {  // sobol_2d scramble pair (:31)
      unsigned s0 = rng.next_u32(), s1 = rng.next_u32();
      scrambles[(size_t)(dims + 2u * (t - dims)) * n_pixels + pix] = s0;
      scrambles[(size_t)(dims + 2u * (t - dims) + 1u) * n_pixels + pix] = s1;
    }
    for (unsigned i = 0; i < spp; ++i) mine[i] = (unsigned short)i;
    // per-pixel-sample shuffles of n_samples_per_pixel_sample = 1 element only consume RNG draws (:14-21)
    for (unsigned i = 0; i < spp; ++i) (void)rng.bounded(1u);
    // shuffle(samples, n_pixel_samples, 1) (:22, :114-124)
    for (unsigned i = 0; i < spp; ++i) {
      unsigned other = i + rng.bounded(spp - i);
      unsigned short a = mine[i], b = mine[other];
      mine[i] = b; mine[other] = a;
    }
    unsigned short* dst = perms + (size_t)t * spp * n_pixels + pix;
    for (unsigned i = 0; i < spp; ++i) dst[(size_t)i * n_pixels] = mine[i];
  }
}

// table look-ups used by raygen / shade: value of dimension d for sample s of chunk pixel pix
struct Tables { const unsigned* scrambles; const unsigned short* perms; unsigned spp, dims, n_pixels; };
RT_DEV Tables tables_of(const PassState& ps) { Tables t; t.scrambles = ps.scrambles; t.perms = ps.perms; t.spp = ps.spp; t.dims = ps.dims; t.n_pixels = ps.n_pixels; return t; }
RT_DEV float table_1d(const Tables& tb, unsigned pix, unsigned d, unsigned s) {
  unsigned k = tb.perms[((size_t)d * tb.spp + s) * tb.n_pixels + pix];
  return u32_to_unit(tb.scrambles[(size_t)d * tb.n_pixels + pix] ^ vdc_bits(k));
}
RT_DEV f2 table_2d(const Tables& tb, unsigned pix, unsigned d, unsigned s) {
  unsigned k = tb.perms[((size_t)(tb.dims + d) * tb.spp + s) * tb.n_pixels + pix];
  unsigned s0 = tb.scrambles[(size_t)(tb.dims + 2u * d) * tb.n_pixels + pix], s1 = tb.scrambles[(size_t)(tb.dims + 2u * d + 1u) * tb.n_pixels + pix];
  return mk2(u32_to_unit(s0 ^ vdc_bits(k)), u32_to_unit(s1 ^ sobol1_bits(k)));
}

// ================================================================================ K1 raygen
RT_DEV f3 xf_point44(const float* m, f3 p) {  // transform.rs:264-286
  float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
  float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
  float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
  float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
  if (wp == 1.0f) return mk3(xp, yp, zp);
  return mk3(xp, yp, zp) / wp;
}
RT_DEV f3 xf_vector44(const float* m, f3 v) {
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
struct CameraRay { f3 o, d, rx_o, ry_o, rx_d, ry_d; };
// PerspectiveCamera::generate_ray_differential (camera.rs:150-202) + Ray::transform (ray.rs:46-71)
// + scale_differentials (ray.rs:73-80)
RT_DEV CameraRay generate_camera_ray(const FrameParams& fp, f2 p_film, f2 p_lens, float diff_scale) {
  CameraRay r;
  f3 p_camera = xf_point44(fp.r2c, mk3(p_film.x, p_film.y, 0.0f));
  f3 o = mk3(0, 0, 0), d = normalize(p_camera);
  f3 rx_o, ry_o, rx_d, ry_d;
  if (fp.lens_radius > 0.0f) {
    f2 pl = concentric_sample_disk(p_lens); pl.x = fp.lens_radius * pl.x; pl.y = fp.lens_radius * pl.y;
    float ft = fp.focal_distance / d.z;
    f3 p_focus = o + ft * d;
    o = mk3(pl.x, pl.y, 0.0f);
    d = normalize(p_focus - o);
    f3 origin = mk3(pl.x, pl.y, 0.0f);
    f3 dx = normalize(p_camera + fp.dx_camera);
    float ft_x = fp.focal_distance / dx.z;
    f3 dy = normalize(p_camera + fp.dy_camera);
    float ft_y = fp.focal_distance / dy.z;
    rx_o = origin; ry_o = origin;
    rx_d = normalize(ft_x * dx - origin); ry_d = normalize(ft_y * dy - origin);
  } else {
    rx_o = o; ry_o = o;
    rx_d = normalize(p_camera + fp.dx_camera); ry_d = normalize(p_camera + fp.dy_camera);
  }
  // Ray::transform with the origin error nudge (transform.rs:175-188)
  const float* m = fp.c2w;
  f3 to = xf_point44(m, o);
  f3 o_error = gamma_n(3) * mk3(fabsf(m[0] * o.x) + fabsf(m[1] * o.y) + fabsf(m[2] * o.z) + fabsf(m[3]),
                                fabsf(m[4] * o.x) + fabsf(m[5] * o.y) + fabsf(m[6] * o.z) + fabsf(m[7]),
                                fabsf(m[8] * o.x) + fabsf(m[9] * o.y) + fabsf(m[10] * o.z) + fabsf(m[11]));
  f3 td = xf_vector44(m, d);
  float l2 = len2(td);
  if (l2 > 0.0f) { float dt = dot(abs3(td), o_error) / l2; to = to + td * dt; }
  r.o = to; r.d = td;
  r.rx_o = xf_point44(m, rx_o); r.ry_o = xf_point44(m, ry_o);
  r.rx_d = xf_vector44(m, rx_d); r.ry_d = xf_vector44(m, ry_d);
  r.rx_o = r.o + (r.rx_o - r.o) * diff_scale; r.ry_o = r.o + (r.ry_o - r.o) * diff_scale;
  r.rx_d = r.d + (r.rx_d - r.d) * diff_scale; r.ry_d = r.d + (r.ry_d - r.d) * diff_scale;
  return r;
}

static __global__ void __launch_bounds__(256) k_raygen(FrameParams fp, PassState ps) {
  const unsigned stride = gridDim.x * blockDim.x;
  const Tables tb = tables_of(ps);
  unsigned n_camera = 0;
  for (unsigned base = blockIdx.x * blockDim.x; base < ps.cap; base += stride) {
    const unsigned pid = base + threadIdx.x;
    bool in_bounds = false;
    CameraRay cr; unsigned long long rng_state = 0ull;
    if (pid < ps.cap) {
      unsigned sl, pix; split_path_id(ps, pid, sl, pix); const unsigned s = ps.s0 + sl;
      int x, y; unsigned long long pixel_index;
      owned_pixel(fp, fp.chunk_first + pix, x, y, pixel_index);
      in_bounds = y < fp.sb_y1 && x >= fp.pb_x0 && x < fp.pb_x1 && y >= fp.pb_y0 && y < fp.pb_y1;  // renderer.rs:103
      // get_camera_sample (zerotwosequence.rs:182-192): 2D#0 film, 1D#0 time, 2D#1 lens
      f2 o = table_2d(tb, pix, 0, s);
      f2 p_film = mk2((float)x + o.x, (float)y + o.y);
      const f2 p_lens = fp.lens_radius > 0.0f ? table_2d(tb, pix, 1, s) : mk2(0.0f, 0.0f);  // (a pinhole camera never reads it, and a frame then does not build it: table_groups_frame)
      cr = generate_camera_ray(fp, p_film, p_lens, 1.0f / sqrtf((float)ps.spp));
      Pcg32 rng; rng.set_sequence(pixel_index * (unsigned long long)ps.spp + s + (1ull << 32));  // keyed per-sample stream
      rng_state = rng.state;
      ps.lacc[pid] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(in_bounds ? 0u : RT_STATE_OUT_OF_BOUNDS));
      ps.pfilm[pid] = make_float2(p_film.x, p_film.y);
    }
    n_camera += in_bounds ? 1u : 0u;
    unsigned slot = pid;  // every sample traced: path i is entry i is slot i
    if (!ps.all_in_bounds) {
      const int ci[1] = {0}; const bool pr[1] = {in_bounds}; unsigned sl_[1];
      block_push<1>(ps.cnt_out, ps.shard_cap, ci, pr, sl_);
      slot = sl_[0];
    }
    if (in_bounds) {  // the path's travelling records, at its slot of bounce 0's queue
      ps.out.o[slot] = make_float4(cr.o.x, cr.o.y, cr.o.z, kInf);
      ps.out.d[slot] = make_float4(cr.d.x, cr.d.y, cr.d.z, 0.0f);
      if (!RT_FRESH_BETA(ps)) ps.out.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // (left out: bounce 0 rebuilds the record from the slot, PassState::fresh)
      if (!RT_FRESH_ST(ps)) ps.out.st[slot] = make_uint4(pack_state(0, false, 1, 2), pid, (unsigned)rng_state, (unsigned)(rng_state >> 32));
    }
  }
  // camera samples actually generated: calls of PathIntegrator::li (samples outside pixel_bounds are skipped, renderer.rs:103)
  for (int off = 32; off > 0; off >>= 1) n_camera += __shfl_down(n_camera, off);
  if ((threadIdx.x & 63u) == 0u && n_camera) atomicAdd(&ps.stats[ST_CAMERA], (unsigned long long)n_camera);
}

// Without the mask evaluator a GENERAL trace kernel's largest part is the quadric test (134 VGPRs inline) or the nested instance walk (133): bound to four
// waves per SIMD they take 128 / 127 with at most two spilled dwords (measured: three waves instead of two were worth 13 % / 27 % on mis-spheres / instances-10k).
#define RT_GEN_MIN_WAVES(G) ((G) >= RT_GEN_NO_MASKS ? 4 : 1)
// ================================================================================ K2/K4 trace
// rays indexed through `queue` (NULL => identity). Small scenes are copied into LDS first
// (SMALL): nodes and triangle records are then read from LDS for the whole kernel.
#define RT_SMALL_NODES 256
#define RT_SMALL_TRIS 128
// Planar in LDS: plane k of the nodes holds dword k of every 32-byte node, plane k of the triangles coordinate k of every triangle (plane 9: its flags word).
// A lane reads node `cur`; the 64 lanes of a wave sit at different nodes. As 32-byte records two b128 reads per node hit the banks (cur * 8 + k) mod 32: only
// FOUR distinct bank groups, so the ~20 distinct nodes a wave of bounced rays holds were served one after the other, eight to a group - the kernel was bound
// by LDS bank conflicts (SQ_LDS_BANK_CONFLICT 1.26e10 > SQ_ACTIVE_INST_LDS 8.4e9 cycles on S1, round 3). Planar, the dword k of node cur sits in bank cur mod 32:
// distinct nodes of a small scene fall into distinct banks, equal nodes are a broadcast.
// Plane 7 holds a node's control word in the form the finite-ray walks read it (lds_node_ctl): a leaf's primitive count in bits 0-15, an interior node's split
// axis as ONE bit, 1 << (16 + axis) - `ctl & sign mask of the ray` is then the reference's "direction negative along the split axis" in two instructions
// instead of six. LdsSrcT::node hands the other walks the node record as it is in HBM (n_prims | axis << 16).
RT_DEV unsigned lds_node_ctl(unsigned packed) { const unsigned n = packed & 0xffffu, axis = (packed >> 16) & 0xffu; return n != 0u ? n : (1u << (16u + (axis < 2u ? axis : 2u))); }  // (an axis other than 0 / 1 reads as z, as in traverse())
template <int N, int T>
struct LdsSrcT {
  const float* nodes; const float* tris;
  RT_DEV void node(int i, float4& a, float4& b) const {  // node planes: min.x max.x min.y max.y min.z max.z offset ctl (an axis' two bounds in neighbouring planes: one two-register read feeds one packed subtraction)
    a = make_float4(nodes[i], nodes[2 * N + i], nodes[4 * N + i], nodes[N + i]);
    const unsigned ctl = __float_as_uint(nodes[7 * N + i]);
    b = make_float4(nodes[3 * N + i], nodes[5 * N + i], nodes[6 * N + i], __uint_as_float((ctl & 0xffffu) | (((ctl >> 17) & 3u) << 16)));
  }
  RT_DEV void tri(int i, f3& p0, f3& p1, f3& p2) const {
    p0 = mk3(tris[i], tris[T + i], tris[2 * T + i]); p1 = mk3(tris[3 * T + i], tris[4 * T + i], tris[5 * T + i]); p2 = mk3(tris[6 * T + i], tris[7 * T + i], tris[8 * T + i]);
  }
  RT_DEV void tri_flags(int i, f3& p0, f3& p1, f3& p2, unsigned& flags) const { tri(i, p0, p1, p2); flags = __float_as_uint(tris[9 * T + i]); }
};
typedef LdsSrcT<RT_SMALL_NODES, RT_SMALL_TRIS> LdsSrc;
// (LDS arrays sized for S1's 63 nodes / 32 triangles - 11.5 KB per workgroup, eight resident instead of seven, 62 VGPRs - were measured in round 4: closest hit
// 328 -> 341 ms, shadow rays 155 -> 166. Seven waves per SIMD is this kernel's optimum.)
// copies the scene's nodes and leaf-ordered triangle records into the planar LDS arrays (every thread of the workgroup; followed by a barrier at the caller)
// BOUNDS_ONLY: the six bound planes and no triangles (a mid-size scene's closest-hit walk: its leaves' primitives are in the link words, its triangles stay in HBM)
template <int BLOCK, int N = RT_SMALL_NODES, int T = RT_SMALL_TRIS, bool BOUNDS_ONLY = false>
RT_DEV void stage_small_scene(const DScene& sc, float* s_nodes, float* s_tris) {
  for (unsigned i = threadIdx.x; i < 2u * sc.n_nodes; i += BLOCK) {
    const float4 v = sc.nodes[i]; const unsigned n = i >> 1, h = (i & 1u) * 4u;
    if (h == 0u) { s_nodes[n] = v.x; s_nodes[2 * N + n] = v.y; s_nodes[4 * N + n] = v.z; s_nodes[N + n] = v.w; }
    else if (BOUNDS_ONLY) { s_nodes[3 * N + n] = v.x; s_nodes[5 * N + n] = v.y; }
    else { s_nodes[3 * N + n] = v.x; s_nodes[5 * N + n] = v.y; s_nodes[6 * N + n] = v.z; s_nodes[7 * N + n] = __uint_as_float(lds_node_ctl(__float_as_uint(v.w))); }
  }
  if (BOUNDS_ONLY) return;
  for (unsigned i = threadIdx.x; i < 3u * sc.n_tris; i += BLOCK) {
    const float4 v = sc.tri_p[i]; const unsigned t = i / 3u, r = i - 3u * t;
    s_tris[(3u * r) * T + t] = v.x; s_tris[(3u * r + 1u) * T + t] = v.y; s_tris[(3u * r + 2u) * T + t] = v.z;
    if (r == 2u) s_tris[9 * T + t] = v.w;
  }
}

// ---- The walks of an LDS-resident scene WITHOUT a stack, over LINK tables (FINITE: the node test of a finite ray, slab_test_finite; otherwise the reference's selects).
// The order in which BVH::intersect reaches the nodes depends on the ray only through the signs of its direction, so "where the walk goes when node i's box passes" and
// "where it carries on after node i and everything below it" are constants of (octant, i): one word per (octant, node), DScene::link8, staged into LDS. A node step is the box
// test and one select - no push, no pop, no stack memory. Per ray the sequence of leaf-box tests, primitive tests and t_max updates is the reference's: a far child is tested
// when the reference pops it, with the t_max of that moment. Rounds 4 - 5 walked over a skip table first (closest_small / occluded_small: next node from the node record's
// split axis and second child) - replaced by the link words, which also let the tables LEAVE NODES OUT:
// a node's box CONTAINS its children's boxes (bvh/mod.rs:279-287: an interior node's bounds are the union), so for a finite ray an interior node's test fails only if both
// children's tests fail too (minima / maxima are monotone in the box, slab_test_finite) - the test of an interior node decides nothing, it only saves work when it fails.
// rt_scene_create measures, on synthetic path-like rays, which interior nodes' tests save less than they cost and leaves those out of link8 (S1: 8 of 19 interior nodes stay;
// 18.5 -> 12.3 node tests per ray); the walk visits the same leaves in the same order with the same t_max - every leaf is still tested against its own box at its own time,
// which is all BVH::intersect's result depends on. Rays with a zero direction component keep every node (link8_full, read from HBM by the few hundred waves that hold one):
// the literal node test's NaN rules are not the minima / maxima the argument needs.
// Occlusion rays: intersect_p's answer is a property of the tree and the ray alone - "some primitive passes its test and every box above it passes the node test" (t_max never
// shrinks; bvh/mod.rs:443-500 returns at the first accepted primitive) - so they all walk ONE row (row 8: an order the calibration chose) and the kernel stages one row of links.
// Quadrics and masked triangles in the leaves (GENERAL) go through leaf_prim_test inside the same walks.
#ifndef RT_ANY_EARLY_SIGN
#define RT_ANY_EARLY_SIGN 1
#endif
// a leaf's primitives in its link word: first primitive from bit 16 up (7 bits for the 128-primitive LDS scenes, 11 for the mid-size ones), the count above it
#define RT_LINK_OFF_BITS(N) ((N) > 256 ? 11 : 7)
#define RT_LINK_LEAF_OFF(lk, N) ((int)(((lk) >> 16) & ((1u << RT_LINK_OFF_BITS(N)) - 1u)))
#define RT_LINK_LEAF_N(lk, N) ((int)(((lk) & 0x7fffffffu) >> (16 + RT_LINK_OFF_BITS(N))))
// (a leaf's link word carries its primitives in the half an interior node's uses for "enter": bit 31, count above the first primitive's bits)
// GENERAL != 0 (round 5): leaves may hold quadrics and masked triangles (leaf_prim_test; a plain triangle of such a scene takes tri_test_pre as in the stack walk it replaces)
template <int N, int T, bool FINITE, int GENERAL = 0>
RT_DEV bool occluded_small_links(const float* __restrict__ s_nodes, const float* __restrict__ s_tris, const unsigned* __restrict__ link /* the occlusion walk's row */, const int n_nodes, const int start, const Ray ray, const GeneralCtx gen = GeneralCtx{nullptr, false}) {
  const f3 inv_dir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  const RayPre rp = ray_pre(ray);
  const float* const tpx = s_tris + rp.kx * T; const float* const tpy = s_tris + rp.ky * T; const float* const tpz = s_tris + rp.kz * T;
  const f3 op = permute(ray.o, rp.kx, rp.ky, rp.kz);
  bool found = false;
  int cur = start;  // n_nodes: done
  for (;;) {
    if (cur < n_nodes) {
      const float* nd = s_nodes + cur;
      const float4 n0 = make_float4(nd[0], nd[2 * N], nd[4 * N], nd[N]), n1 = make_float4(nd[3 * N], nd[5 * N], 0.0f, 0.0f);
      const unsigned lk = link[cur];
      const bool hit = slab_test_t<FINITE>(n0, n1, ray, inv_dir, inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f);
      int next = (int)(lk & 0xffffu);
      if (hit) {
        const int n_prims = (int)(lk >> 31) * RT_LINK_LEAF_N(lk, N);
        if (n_prims != 0) {
          const int off = RT_LINK_LEAF_OFF(lk, N);
          for (int i = 0; i < n_prims; ++i) {
            const int t = off + i;
            TriHit h;
            if (GENERAL) { if (leaf_prim_test<GENERAL>(LdsSrcT<N, T>{s_nodes, s_tris}, gen, t, ray, rp, h)) { found = true; next = n_nodes; break; } }
            else { const f3 p0t = mk3(tpx[t] - op.x, tpy[t] - op.y, tpz[t] - op.z), p1t = mk3(tpx[3 * T + t] - op.x, tpy[3 * T + t] - op.y, tpz[3 * T + t] - op.z), p2t = mk3(tpx[6 * T + t] - op.x, tpy[6 * T + t] - op.y, tpz[6 * T + t] - op.z);  /* (the LDS planes of plain triangles: read where they are tested - the GENERAL and TRIS_GLOBAL instantiations never index s_tris here) */
              if (tri_test_permuted<RT_ANY_EARLY_SIGN != 0>(p0t, p1t, p2t, rp.sx, rp.sy, rp.sz, ray.t_max, h)) { found = true; next = n_nodes; break; } }
          }
        } else next = (int)(lk >> 16);
      }
      cur = next;
    }
    if (__ballot(cur < n_nodes) == 0ull) break;
  }
  return found;
}
// TRIS_GLOBAL: the triangles are read from HBM (tri_p; s_tris unused) - the mid-size scenes' closest-hit walk, whose LDS holds the bounds and eight link rows
template <int N, int T, bool FINITE, int LEAF_MIN, int GENERAL = 0, bool TRIS_GLOBAL = false>
RT_DEV bool closest_small_links(const float* __restrict__ s_nodes, const float* __restrict__ s_tris, const unsigned* __restrict__ link8 /* rows N (LDS) or n_nodes (HBM) apart */, const int row, const int n_nodes, Ray ray, int& prim_out, TriHit& hit_out, const GeneralCtx gen = GeneralCtx{nullptr, false}, const float4* __restrict__ tri_p = nullptr) {
  const f3 inv_dir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  const int neg_x = inv_dir.x < 0.0f, neg_y = inv_dir.y < 0.0f, neg_z = inv_dir.z < 0.0f;
  const unsigned oct = (neg_x ? 1u : 0u) | (neg_y ? 2u : 0u) | (neg_z ? 4u : 0u);
  const unsigned* const link = link8 + oct * row;
  const RayPre rp = ray_pre(ray);
  const float* const tpx = s_tris + rp.kx * T; const float* const tpy = s_tris + rp.ky * T; const float* const tpz = s_tris + rp.kz * T;
  const f3 op = permute(ray.o, rp.kx, rp.ky, rp.kz);
  bool found = false;
  int cur = (int)link8[8 * row + oct], leaf_off = 0, leaf_n = 0;  // the octant's first tested node
  for (;;) {
    unsigned long long holders = 0ull;
    for (;;) {
      if (leaf_n == 0 && cur < n_nodes) {
        const float* nd = s_nodes + cur;
        const float4 n0 = make_float4(nd[0], nd[2 * N], nd[4 * N], nd[N]), n1 = make_float4(nd[3 * N], nd[5 * N], 0.0f, 0.0f);
        const unsigned lk = link[cur];
        int next = (int)(lk & 0xffffu);
        if (slab_test_t<FINITE>(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z)) {
          if ((int)lk < 0) { leaf_off = RT_LINK_LEAF_OFF(lk, N); leaf_n = RT_LINK_LEAF_N(lk, N); }
          else next = (int)(lk >> 16);
        }
        cur = next;
      }
      holders = __ballot(leaf_n > 0);
      if (__ballot(leaf_n == 0 && cur < n_nodes) == 0ull) break;
      if (__builtin_popcount((unsigned)holders) + __builtin_popcount((unsigned)(holders >> 32)) >= LEAF_MIN) break;
    }
    if (holders == 0ull) break;
    if (leaf_n > 0) {
      const int t = leaf_off;
      TriHit h;
      if (GENERAL) { if (leaf_prim_test<GENERAL>(LdsSrcT<N, T>{s_nodes, s_tris}, gen, t, ray, rp, h)) { found = true; ray.t_max = h.t; prim_out = t; hit_out = h; } }
      else if (TRIS_GLOBAL) { f3 q0, q1, q2; load_tri(tri_p, t, q0, q1, q2); if (tri_test_pre(q0, q1, q2, ray, rp, h)) { found = true; ray.t_max = h.t; prim_out = t; hit_out = h; } }
      else { const f3 p0t = mk3(tpx[t] - op.x, tpy[t] - op.y, tpz[t] - op.z), p1t = mk3(tpx[3 * T + t] - op.x, tpy[3 * T + t] - op.y, tpz[3 * T + t] - op.z), p2t = mk3(tpx[6 * T + t] - op.x, tpy[6 * T + t] - op.y, tpz[6 * T + t] - op.z);  /* (the LDS planes of plain triangles: read where they are tested - the GENERAL and TRIS_GLOBAL instantiations never index s_tris here) */
        if (tri_test_permuted(p0t, p1t, p2t, rp.sx, rp.sy, rp.sz, ray.t_max, h)) { found = true; ray.t_max = h.t; prim_out = t; hit_out = h; } }
      leaf_off += 1; leaf_n -= 1;
    }
  }
  return found;
}
// occluded_small_links in rounds (RT_LDS_LEAF_MIN_ANY > 1): a lane that reaches a leaf holds it until LEAF_MIN lanes hold one, then every holder tests one primitive
template <int N, int T, bool FINITE, int LEAF_MIN, int GENERAL = 0>
RT_DEV bool occluded_small_links_rounds(const float* __restrict__ s_nodes, const float* __restrict__ s_tris, const unsigned* __restrict__ link, const int n_nodes, const int start, const Ray ray, const GeneralCtx gen = GeneralCtx{nullptr, false}) {
  const f3 inv_dir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
  const int neg_x = inv_dir.x < 0.0f, neg_y = inv_dir.y < 0.0f, neg_z = inv_dir.z < 0.0f;
  const RayPre rp = ray_pre(ray);
  const float* const tpx = s_tris + rp.kx * T; const float* const tpy = s_tris + rp.ky * T; const float* const tpz = s_tris + rp.kz * T;
  const f3 op = permute(ray.o, rp.kx, rp.ky, rp.kz);
  bool found = false;
  int cur = start, leaf_off = 0, leaf_n = 0;
  for (;;) {
    unsigned long long holders = 0ull;
    for (;;) {
      if (leaf_n == 0 && cur < n_nodes) {
        const float* nd = s_nodes + cur;
        const float4 n0 = make_float4(nd[0], nd[2 * N], nd[4 * N], nd[N]), n1 = make_float4(nd[3 * N], nd[5 * N], 0.0f, 0.0f);
        const unsigned lk = link[cur];
        int next = (int)(lk & 0xffffu);
        if (slab_test_t<FINITE>(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z)) {
          if ((int)lk < 0) { leaf_off = RT_LINK_LEAF_OFF(lk, N); leaf_n = RT_LINK_LEAF_N(lk, N); }
          else next = (int)(lk >> 16);
        }
        cur = next;
      }
      holders = __ballot(leaf_n > 0);
      if (__ballot(leaf_n == 0 && cur < n_nodes) == 0ull) break;
      if (__builtin_popcount((unsigned)holders) + __builtin_popcount((unsigned)(holders >> 32)) >= LEAF_MIN) break;
    }
    if (holders == 0ull) break;
    if (leaf_n > 0) {
      const int t = leaf_off;
      TriHit h;
      leaf_off += 1; leaf_n -= 1;
      if (GENERAL) { if (leaf_prim_test<GENERAL>(LdsSrcT<N, T>{s_nodes, s_tris}, gen, t, ray, rp, h)) { found = true; cur = n_nodes; leaf_n = 0; } }
      else { const f3 p0t = mk3(tpx[t] - op.x, tpy[t] - op.y, tpz[t] - op.z), p1t = mk3(tpx[3 * T + t] - op.x, tpy[3 * T + t] - op.y, tpz[3 * T + t] - op.z), p2t = mk3(tpx[6 * T + t] - op.x, tpy[6 * T + t] - op.y, tpz[6 * T + t] - op.z);  /* (the LDS planes of plain triangles: read where they are tested - the GENERAL and TRIS_GLOBAL instantiations never index s_tris here) */
        if (tri_test_permuted<RT_ANY_EARLY_SIGN != 0>(p0t, p1t, p2t, rp.sx, rp.sy, rp.sz, ray.t_max, h)) { found = true; cur = n_nodes; leaf_n = 0; } }
    }
  }
  return found;
}


// Where a trace launch reads its rays and writes its results: element [pid * stride] of each pointer (strides in elements of the pointer's
// type), so that the same kernels serve the records of a frame (RayRec / VertRec / ShadowRec / MisRec) and the planar arrays of the batch entry points.
struct TraceIO {
  SPtr<const float4> ray_o; SPtr<const float4> ray_d; unsigned ray_stride;  // (o | t_max), (d | flag)
  SPtr<float4> hits; unsigned hit_stride; int hit_b2;  // closest hit: (t, prim, b0, b1), or (b2, prim, b0, b1) inside a frame (shade needs the three barycentrics, not t)
  unsigned* occluded; unsigned occ_stride;        // any hit: 1 / 0 (occ_stride 0: one BYTE per ray at ((unsigned char*)occluded)[pid] - the frame's dense result arrays, PassState::occ_sh / occ_mi)
  // any hit inside a frame: a shadow ray with d.w != 0 belongs to a vertex without MIS ray - its `direct_add` goes into lacc right here if unoccluded
  float4* lacc; unsigned lacc_stride; SPtr<const float4> direct_add; unsigned add_stride;
  // any hit on scenes with masked meshes: 1 = Triangle::intersect_p's test (alpha and shadowalpha, mesh.rs:534-582: shadow rays), 0 = Triangle::intersect's
  // (alpha only, mesh.rs:353-370): a BSDF-sampled MIS ray toward an infinite light is traced by scene.intersect (integrator/mod.rs:291-309) although only
  // its occlusion is read
  int shadow_masks;
  // the launch's rays are the entries of a sharded queue whose records sit at their slots (the path rays of a bounce, PassState::in): entry i -> slot by the
  // shard counts alone, `queue` is then only a non-NULL marker
  int queue_is_slots;
};
template <class AddPtr>
RT_DEV void trace_write_any(float4* __restrict__ lacc, size_t ls, const AddPtr direct_add, size_t as, unsigned* __restrict__ occluded, size_t os,
                            unsigned pid, float dw, bool found) {
  // Shadow rays of the frame loop carry d.w = 1 when the vertex has no MIS ray in flight: the light-sampling
  // term of estimate_direct is then complete and `L += beta * (Ld / pick_pdf)` (precomputed by k_shade into
  // direct_add) is applied right here if the ray is unoccluded. Otherwise the flag is left for k_resolve.
  // Inside a frame (lacc != nullptr) `pid` is the shadow RECORD's index and d.w carries (complete-here flag << 31) | path id.
  if (lacc != nullptr && (__float_as_uint(dw) >> 31) != 0u) {
    if (!found) { const size_t path = __float_as_uint(dw) & 0x7fffffffu; float4 a = direct_add[pid * as]; float4 l = lacc[path * ls]; lacc[path * ls] = make_float4(l.x + a.x, l.y + a.y, l.z + a.z, l.w); }
  } else if (os == 0) ((unsigned char*)occluded)[pid] = found ? (unsigned char)1 : (unsigned char)0;
  else occluded[pid * os] = found ? 1u : 0u;
}

// BLOCK threads per workgroup, DEPTH = to-visit stack entries per lane (the host picks the
// smallest of 16/32/64 that covers the tree height; the reference's fixed 64 is the maximum).
#ifndef RT_LDS_TRACE_WAVES  // the LDS-resident kernels' register bound in waves per SIMD: their LDS (21.5 KB per 256 lanes) lets seven workgroups share a CU
#define RT_LDS_TRACE_WAVES 7
#endif
#ifndef RT_LDS_ANY_WAVES      // ... the occlusion kernel, 13.9 KB of LDS per 256 lanes, is bound by its registers
#define RT_LDS_ANY_WAVES 8
#endif
// MID (round 5): a scene of <= RT_MID_NODES nodes and <= RT_MID_TRIS triangles (S3: 2461 / 1294) in the LDS of ONE 1024-lane workgroup per CU (157 KB): occlusion rays only -
// closest hit needs eight link rows. The walk is the 256-node kernel's (occluded_small_links_rounds over the occlusion row, pruned by the same calibration); leaf phases at 8
// waiting lanes. S3 occlusion rays: k_trace_quad 100.7 ms -> 82.4 (all nodes) -> 76.1 (pruned: 35.4 -> 26.2 node tests per calibration ray) -> 65.4 (leaf phases at 8; 16: 66.7,
// 32: 72.0). Measured and not kept: the same walk as persistent waves with refill (idle lanes take the wave's next rays at 8 / 16 / 24 / 32 idle, leaf phases at 8 / 12 / 16):
// 104.4 / 96.0 / 94.5 / 98.5 ms - 74 registers cost nothing here (LDS bounds the CU at four waves per SIMD), but set-up at a quarter of the lanes and the longer round do, as
// in k_trace_pool; SQ of the kept kernel before leaf batching: 18.2 lanes per VALU instruction, waits 35 %, LDS bank conflicts 32 % of its LDS cycles, no LDS waits to speak of.
#ifndef RT_MID_LEAF_MIN_ANY
#define RT_MID_LEAF_MIN_ANY 8  // (> 1: the occlusion walk in rounds, occluded_small_links_rounds)
#endif
#ifndef RT_LDS_LEAF_MIN_CLOSEST_GENERAL
#define RT_LDS_LEAF_MIN_CLOSEST_GENERAL 16
#endif
#ifndef RT_LDS_LEAF_MIN_ANY_GENERAL
#define RT_LDS_LEAF_MIN_ANY_GENERAL 56  // occlusion rays of an LDS-resident scene with quadrics / masks: leaf phases at this many waiting lanes (1: every lane tests its leaf at once)
#endif
#ifndef RT_MID_LEAF_MIN_CLOSEST
#define RT_MID_LEAF_MIN_CLOSEST 16
#endif
#define RT_MID_NODES 2816
#define RT_MID_TRIS 1408
template <bool ANY, bool COUNT, int BLOCK, int DEPTH, int GENERAL = 0, int MID = 0>  // LDS-resident and mid-size scenes. GENERAL: quadrics / alpha-masked triangles in the leaves (no object instances)
__global__ void __launch_bounds__(BLOCK, MID ? 4 : ((GENERAL == 0 && !COUNT) ? (ANY ? RT_LDS_ANY_WAVES : RT_LDS_TRACE_WAVES) : RT_GEN_MIN_WAVES(GENERAL))) k_trace(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                 unsigned count_static, unsigned long long* stats, int st_rays, int st_nodes, int st_tris) {
  // (plain loads and stores, no non-temporal hint: the scene is in LDS, the ray streams are all these launches read)
  const float4* __restrict__ ray_o = sraw(io.ray_o); const float4* __restrict__ ray_d = sraw(io.ray_d); const size_t rs = io.ray_stride;
  float4* __restrict__ hits = sraw(io.hits); const size_t hs = io.hit_stride; const bool hit_b2 = io.hit_b2 != 0;
  unsigned* __restrict__ occluded = io.occluded; const size_t os = io.occ_stride;
  float4* __restrict__ lacc = io.lacc; const size_t ls = io.lacc_stride; const float4* __restrict__ direct_add = sraw(io.direct_add); const size_t as = io.add_stride;
  // node indices of a tiny scene fit 16 bits: half the stack bytes => more resident waves per CU
  typedef unsigned short StackT;
  // LINKS: an LDS-resident scene's rays that do not count visits walk WITHOUT a stack over the link tables (closest_small_links, occluded_small_links[_rounds]) -
  // plain triangles, quadrics and masked triangles alike. Counting launches (the reference's walk, visit by visit) and HBM scenes keep the stack walk (traverse).
  constexpr bool LINKS = !COUNT;
  __shared__ StackT stack[LINKS ? 1 : DEPTH * BLOCK];
  constexpr int NN = MID ? RT_MID_NODES : RT_SMALL_NODES, NT = MID ? RT_MID_TRIS : RT_SMALL_TRIS;
  static_assert(!MID || (!COUNT && GENERAL == 0), "mid-size LDS scenes: plain triangles, no visit counts");
  constexpr bool MIDC = MID != 0 && !ANY;  // closest hit of a mid-size scene: six bound planes + eight link rows fill the LDS, the triangles stay in HBM
  typedef LdsSrcT<NN, NT> LdsS;
  __shared__ float s_nodes[(MIDC ? 6 : 8) * NN];
  __shared__ float s_tris[MIDC ? 1 : 10 * NT];
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  if (blockIdx.x * BLOCK >= count) return;  // short queues (MIS rays, late bounces): most blocks of the persistent grid have nothing to stage for
  __shared__ unsigned s_link[LINKS ? (ANY ? 1 : 8) * NN + 8 : 1];  // DScene::link8 (the tested nodes' links), rows NN apart, then the 8 start nodes
  stage_small_scene<BLOCK, NN, NT, MIDC>(sc, s_nodes, s_tris); __syncthreads();
  if (LINKS) {
    // DScene::link8: rows 0 - 7 (closest hit, by octant), their 8 starts, row 8 (occlusion rays), its start
    if (ANY) { for (unsigned k = threadIdx.x; k < sc.n_nodes; k += BLOCK) s_link[k] = sc.link8[8u * sc.n_nodes + 8u + k]; if (threadIdx.x == 0u) s_link[NN] = sc.link8[9u * sc.n_nodes + 8u]; }
    else {
      for (unsigned i = threadIdx.x; i < 8u * sc.n_nodes; i += BLOCK) { const unsigned o = i / sc.n_nodes, k = i - o * sc.n_nodes; s_link[o * NN + k] = sc.link8[i]; }
      if (threadIdx.x < 8u) s_link[8 * NN + threadIdx.x] = sc.link8[8u * sc.n_nodes + threadIdx.x];
    }
    __syncthreads();
  }
  const unsigned stride = gridDim.x * BLOCK;
  unsigned n_nodes = 0, n_tris = 0, n_rays = 0;
  auto trace_one = [&](unsigned pid) {
    float4 o4 = ray_o[pid * rs], d4 = ray_d[pid * rs];
    Ray ray; ray.o = mk3(o4.x, o4.y, o4.z); ray.d = mk3(d4.x, d4.y, d4.z); ray.t_max = o4.w;
    int prim = -1; TriHit h; h.t = kInf; h.b0 = h.b1 = h.b2 = 0.0f;
    bool found;
    constexpr int LM = ANY ? (MID ? RT_MID_LEAF_MIN_ANY : (GENERAL ? RT_LDS_LEAF_MIN_ANY_GENERAL : RT_LDS_LEAF_MIN_ANY)) : (MID ? RT_MID_LEAF_MIN_CLOSEST : (GENERAL ? RT_LDS_LEAF_MIN_CLOSEST_GENERAL : RT_LDS_LEAF_MIN_CLOSEST));
    const GeneralCtx gen{sc.self, ANY && io.shadow_masks != 0};
    // Plain-triangle launches that do not count visits take the min / max node test (slab_test_finite) when every ray of the wave has a finite reciprocal
    // direction - all but a few hundred waves of a frame; a wave that holds one ray with a zero direction component walks with the reference's selects.
    constexpr bool FIN_FORMS = !COUNT;
    const bool fin = FIN_FORMS && __ballot(!inv_dir_finite(mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z))) == 0ull;
    if (LINKS) {
      static_assert(!LINKS || ANY || (LM > 1 && LM < 64), "the closest-hit link walk tests its leaves in phases: 1 < LEAF_MIN < 64");
      if (!ANY) {
        if (fin) found = closest_small_links<NN, NT, true, LM, GENERAL, MIDC>(s_nodes, s_tris, s_link, NN, (int)sc.n_nodes, ray, prim, h, gen, sc.tri_p);  // (s_link: 8 rows NN apart, the starts behind them)
        else found = closest_small_links<NN, NT, false, LM, GENERAL, MIDC>(s_nodes, s_tris, sc.link8_full, (int)sc.n_nodes, (int)sc.n_nodes, ray, prim, h, gen, sc.tri_p);
      } else if (LM > 1 && LM < 64) {
        found = fin ? occluded_small_links_rounds<NN, NT, true, LM, GENERAL>(s_nodes, s_tris, s_link, (int)sc.n_nodes, (int)s_link[NN], ray, gen)
                    : occluded_small_links_rounds<NN, NT, false, LM, GENERAL>(s_nodes, s_tris, sc.link8_full + 8u * sc.n_nodes + 8u, (int)sc.n_nodes, (int)sc.link8_full[9u * sc.n_nodes + 8u], ray, gen);
      } else {
        found = fin ? occluded_small_links<NN, NT, true, GENERAL>(s_nodes, s_tris, s_link, (int)sc.n_nodes, (int)s_link[NN], ray, gen)
                    : occluded_small_links<NN, NT, false, GENERAL>(s_nodes, s_tris, sc.link8_full + 8u * sc.n_nodes + 8u, (int)sc.n_nodes, (int)sc.link8_full[9u * sc.n_nodes + 8u], ray, gen);
      }
    }
    else {  // a frame that counts the reference's walk, visit by visit: the stack walk over the LDS copy
      LdsS src{s_nodes, s_tris};
      found = traverse<ANY, COUNT, LdsS, StackT, GENERAL>(src, ray, stack + threadIdx.x, BLOCK, prim, h, n_nodes, n_tris, gen);
    }
    n_rays += 1;
    if (ANY) trace_write_any(lacc, ls, direct_add, as, occluded, os, pid, d4.w, found);
    else hits[pid * hs] = make_float4(hit_b2 ? h.b2 : (found ? h.t : kInf), __int_as_float(found ? prim : -1), h.b0, h.b1);
  };
  for (unsigned i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) trace_one(queue ? qv.get(i) : i);
  if (stats) {
    // one atomic per wave and counter
    for (int off = 32; off > 0; off >>= 1) { n_rays += __shfl_down(n_rays, off); if (COUNT) { n_nodes += __shfl_down(n_nodes, off); n_tris += __shfl_down(n_tris, off); } }
    if ((threadIdx.x & 63u) == 0u) {
      if (n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
      if (COUNT) { atomicAdd(&stats[st_nodes], (unsigned long long)n_nodes); atomicAdd(&stats[st_tris], (unsigned long long)n_tris); }
    }
  }
}

// ---- Measured on the LDS-resident kernel in round 4 and not kept (S1, one box, closest hit / shadow rays per frame; k_trace as it stands: 336 / 168 ms):
//  * tile-local ray ordering - the workgroup counting-sorts tiles of 2048 queue entries in LDS by direction octant (+ the 2 x 2 x 2 cell of the origin) so that
//    the 64 rays of a wave are neighbours in both: 371 -> 364 / 228 -> 210 ms against 371 / 228 for the same kernel unsorted - the 4 KB of LDS it needs cost two
//    resident workgroups per CU, which is worth more than the order (a wave's cost is its LONGEST ray: 17 nodes on average, about twice that for the slowest
//    of 64, whatever their order);
//  * persistent waves with refill as in k_trace_pair (a lane that finishes takes the next ray of the wave's share once 16 are idle; same per-ray steps):
//    413 / 275 ms. The walk itself gains lanes, but a ray's set-up - six IEEE divisions for 1 / d and the watertight test's shear, the queue lookup - then
//    runs for the 16-24 refilled lanes instead of all 64: ~200 instructions at a quarter of the lanes per ~1250 of walk. Refill thresholds 8 / 24 / 32 and
//    leaf thresholds 8 / 24 / 32 all lose;
//  * two rays per lane (both loaded and set up with all lanes active, the second taken out of registers when the first is done, batched over 8 / 16 / 32
//    ready lanes): 374 -> 632 - 668 / 171 -> 296 - 312 ms - 1.75x SLOWER per ray at 93 / 82 VGPRs. Together with the refill result this refutes the "tail"
//    reading of the 28-of-64 lanes: the lanes of a wave do not finish at very different times; they are idle INSIDE a round - the holders of a leaf while the
//    others step nodes, the walkers while the leaf phase tests triangles, and either side of every branch of the node step (hit / miss, push / pop);
//  * (kept) the scene planar in LDS (LdsSrc): 336 -> 329 ms; bank conflicts were real (round 3's counters) but not what the kernel waits for.
struct TraceOut { SPtr<float4> hits; size_t hs; bool hit_b2; unsigned* occluded; size_t os; float4* lacc; size_t ls; SPtr<const float4> direct_add; size_t as; };
RT_DEV TraceOut trace_out_of(const TraceIO& io) {
  TraceOut o; o.hits = io.hits; o.hs = io.hit_stride; o.hit_b2 = io.hit_b2 != 0; o.occluded = io.occluded; o.os = io.occ_stride;
  o.lacc = io.lacc; o.ls = io.lacc_stride; o.direct_add = io.direct_add; o.as = io.add_stride; return o;
}

// ---- K2/K4 for scenes that live in HBM. Ray lengths then vary by an order of magnitude (a camera ray that
// misses the root box ends after one node, one that grazes a silhouette visits a hundred), and with one ray per
// lane per grid-stride iteration a wave runs as long as its longest ray with most lanes idle (measured: 11 of 64
// lanes active per VALU instruction on the 1M-triangle scene). Here a wave is persistent and keeps its lanes
// fed: whenever RT_REFILL_MIN or more lanes have finished their ray, they take the next unclaimed rays of the
// wave's share of the queue (64-entry blocks, interleaved over the waves of the grid so that every wave samples
// the whole queue and the shares finish together; no atomics). The per-ray sequence of node visits and
// triangle tests is unchanged (closest-hit: descend to a leaf, then test its triangles; results bit-identical).
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16
#endif
// GENERAL: the scene holds primitives whose hit test is more than the watertight triangle test - alpha-masked triangles (an accepted hit is
// dropped when its mask texture evaluates to 0, mesh.rs:353-370 / 534-582). Such scenes trace through this kernel only; every other kernel
// keeps the bare triangle loop.
// TransformedPrimitive::intersect / intersect_p (rc/primitive.rs:90-101): the ray goes to object space as `Transform * Ray` does (origin as a point,
// direction as a vector, t_max kept - rc/ray.rs:83-93), the object - its tree, or its single primitive - is intersected there; t is the same parameter
// in both spaces. The nested walk keeps its stack in private memory: this is the general path, not the fast one.
// What an object's walk needs when it runs as child pairs (nested_pair_walk, below the top level's pair steps): the pair records (DScene::pairs holds the
// objects' records behind the top level's, child codes local to the object) and this lane's column of the deferred-tmin array above its pending entries.
// pairs == NULL: the one-node-per-step walk (frames that count visits; scenes whose objects have no records).
struct NestedCtx { const float4* pairs; float* tstack; size_t grid_lanes; };
template <bool ANY, class StackT>
RT_DEV bool nested_pair_walk(const float4* __restrict__ pairs, const float4* __restrict__ nodes, const float4* __restrict__ tri_p, Ray& r, StackT* stack, int stack_stride,
                             float* tstack, size_t grid_lanes, int& prim_out, TriHit& hit_out);
// The walk of an object that holds quadrics or masked triangles, OUT OF LINE: it is the rare path of the general kernels, and inlined into their leaf steps it cost the
// common paths registers (k_trace_top<.., NO_MASKS> 19 -> 112 spilled dwords). Scalars in, eight dwords out (the call ABI's register budget, MEASUREMENTS R2).
struct ObjWalkOut { int found, prim; float t, b0, b1, b2; unsigned n_nodes, n_tris; };
template <bool ANY, bool COUNT, class StackT, int G>
RT_DEVN ObjWalkOut object_walk_general(const DScene* self, unsigned node_base, unsigned prim_base, float ox, float oy, float oz, float dx, float dy, float dz, float t_max,
                                       StackT* stack, int stack_stride, bool shadow_masks) {
  const DScene& sc = *self;
  Ray r; r.o = mk3(ox, oy, oz); r.d = mk3(dx, dy, dz); r.t_max = t_max;
  const GlobalSrc src{sc.nodes + 2 * (size_t)node_base, sc.tri_p + 3 * (size_t)prim_base};
  const GeneralCtx gen{self, shadow_masks, (int)prim_base};
  ObjWalkOut o{}; TriHit h; h.t = kInf; h.b0 = h.b1 = h.b2 = 0.0f;
  o.found = traverse<ANY, COUNT, GlobalSrc, StackT, G>(src, r, stack, stack_stride, o.prim, h, o.n_nodes, o.n_tris, gen) ? 1 : 0;
  o.t = h.t; o.b0 = h.b0; o.b1 = h.b1; o.b2 = h.b2;
  return o;
}
// MASKS / QUADRICS (round 6): the object may hold masked triangles / quadrics (TransformedPrimitive wraps whatever the definition collected, primitive.rs:79-118): its
// primitives are tested in OBJECT space by the general tests - a quadric's own object_to_world is the CTM inside the definition, so the ray is transformed twice, once
// per Transform * Ray, as in the reference (not by one product matrix).
template <bool ANY, bool COUNT, class StackT, bool MASKS = false, bool QUADRICS = false>
RT_DEV bool instance_intersect(const DScene& sc, unsigned inst, f3 o, f3 d, float& t_max, int& prim_out, TriHit& hit_out, unsigned& n_nodes, unsigned& n_tris,
                               StackT* stack, int stack_stride, NestedCtx nc = NestedCtx{nullptr, nullptr, 0}, bool shadow_masks = false) {
  const DInstance& in = sc.instances[inst];
  Ray r; r.o = xf34_point(in.w2o, o); r.d = xf34_vector(in.w2o, d); r.t_max = t_max;
  bool found;
  constexpr int OBJ_GENERAL = MASKS ? RT_GEN_ALL : (QUADRICS ? RT_GEN_NO_MASKS : 0);
  if (in.n_nodes == 0u) {  // an object of one primitive is wrapped as it is (api.rs:1073-1082): no node test
    const float4 pa = sc.tri_p[3 * (size_t)in.prim_base], pb = sc.tri_p[3 * (size_t)in.prim_base + 1], pc = sc.tri_p[3 * (size_t)in.prim_base + 2];
    const f3 p0 = mk3(pa.x, pa.y, pa.z), p1 = mk3(pb.x, pb.y, pb.z), p2 = mk3(pc.x, pc.y, pc.z);
    const unsigned flags = __float_as_uint(pc.w);
    if (COUNT) n_tris += 1;
    TriHit h;
    if (OBJ_GENERAL != 0 && (flags & (RT_FLAG_SPHERE | RT_FLAG_GENERAL_TRI))) found = general_prim_test<MASKS, QUADRICS>(*sc.self, (int)in.prim_base, p0, p1, p2, flags, r, ray_pre(r), shadow_masks, h);
    else found = tri_test(p0, p1, p2, r, h);
    if (found) { prim_out = 0; hit_out = h; r.t_max = h.t; }
  } else if (OBJ_GENERAL != 0 && sc.obj_general) {  // (scenes whose objects hold such primitives carry no pair records for them: DScene::obj_pairs == 0)
    const ObjWalkOut w = object_walk_general<ANY, COUNT, StackT, OBJ_GENERAL>(sc.self, in.node_base, in.prim_base, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.t_max, stack, stack_stride, shadow_masks);
    found = w.found != 0;
    if (COUNT) { n_nodes += w.n_nodes; n_tris += w.n_tris; }
    if (found) { prim_out = w.prim; hit_out.t = w.t; hit_out.b0 = w.b0; hit_out.b1 = w.b1; hit_out.b2 = w.b2; if (!ANY) r.t_max = w.t; }
  } else {
    // the object's walk takes the entries of this lane's stack column above the caller's pending ones (rt_scene_create sizes the column for the
    // deepest top-level path plus the deepest object)
    if (!COUNT && nc.pairs != nullptr) {
      found = nested_pair_walk<ANY, StackT>(nc.pairs + 4 * (size_t)in.node_base, sc.nodes + 2 * (size_t)in.node_base, sc.tri_p + 3 * (size_t)in.prim_base, r, stack, stack_stride,
                                            nc.tstack, nc.grid_lanes, prim_out, hit_out);
    } else {
      const GlobalSrc src{sc.nodes + 2 * (size_t)in.node_base, sc.tri_p + 3 * (size_t)in.prim_base};
      found = traverse<ANY, COUNT, GlobalSrc, StackT>(src, r, stack, stack_stride, prim_out, hit_out, n_nodes, n_tris);
      if (found && !ANY) r.t_max = hit_out.t;
    }
  }
  if (found) t_max = r.t_max;
  return found;
}
// One primitive of a leaf of a GENERAL scene, for the persistent kernels: an object instance (the hit id then names (instance, the object's primitive)), a
// quadric, a masked triangle or a plain one. Returns whether the ray hits; for a closest-hit ray prim / hit / t_max are updated by the caller's rule.
// INSTANCES = false: the caller's scenes hold none (k_trace_top). OBJ_GENERAL: the instanced objects may hold quadrics / masked triangles (k_trace_big only: rt_scene_create
// sends every ray of such a scene there - the pair / four-wide kernels keep the plain object walk and their registers)
template <bool ANY, bool COUNT, class StackT, bool MASKS = true, bool QUADRICS = true, bool INSTANCES = true, bool OBJ_GENERAL = false>
RT_DEV bool general_leaf_prim(const DScene& sc, const float4* __restrict__ tri_p, int prim, const Ray& ray, const RayPre& rp, bool shadow_masks, StackT* nested_stack, int stack_stride,
                              TriHit& h, int& hit_prim, float& t_hit, unsigned& n_nodes, unsigned& n_tris, NestedCtx nc = NestedCtx{nullptr, nullptr, 0}) {
  const float4 a = tri_p[3 * prim], b = tri_p[3 * prim + 1], c = tri_p[3 * prim + 2];
  const f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(c.x, c.y, c.z);
  const unsigned flags = __float_as_uint(c.w);
  if (INSTANCES && (flags & RT_FLAG_INSTANCE)) {
    const unsigned k = __float_as_uint(c.x);
    int oprim = 0; float tm = ray.t_max;
    if (!instance_intersect<ANY, COUNT, StackT, MASKS && OBJ_GENERAL, QUADRICS && OBJ_GENERAL>(sc, k, ray.o, ray.d, tm, oprim, h, n_nodes, n_tris, nested_stack, stack_stride, nc, shadow_masks)) return false;
    hit_prim = (int)(sc.instances[k].id_base + (unsigned)oprim); t_hit = tm;
    return true;
  }
  hit_prim = prim;
  if ((MASKS || QUADRICS) && (flags & (RT_FLAG_SPHERE | RT_FLAG_GENERAL_TRI))) { if (!general_prim_test<MASKS, QUADRICS>(*sc.self, prim, p0, p1, p2, flags, ray, rp, shadow_masks, h)) return false; }
  else if (!tri_test_pre(p0, p1, p2, ray, rp, h)) return false;
  t_hit = h.t;
  return true;
}

template <bool ANY, bool COUNT, int BLOCK, int DEPTH, int GENERAL = 0>
__global__ void __launch_bounds__(BLOCK) k_trace_big(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                     unsigned count_static, unsigned long long* stats, int st_rays, int st_nodes, int st_tris) {
  const RT_SPTR_R(const float4) ray_o = io.ray_o; const RT_SPTR_R(const float4) ray_d = io.ray_d; const size_t rs = io.ray_stride;
  RT_SPTR_R(float4) hits = io.hits; const size_t hs = io.hit_stride; const bool hit_b2 = io.hit_b2 != 0;
  unsigned* __restrict__ occluded = io.occluded; const size_t os = io.occ_stride;
  float4* __restrict__ lacc = io.lacc; const size_t ls = io.lacc_stride; const RT_SPTR_R(const float4) direct_add = io.direct_add; const size_t as = io.add_stride;
  __shared__ int stack_mem[DEPTH * BLOCK];
  int* const stack = stack_mem + threadIdx.x;
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  const unsigned lane = __lane_id();
  const unsigned n_waves = gridDim.x * (BLOCK / 64), wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const GlobalSrc src{sc.nodes, sc.tri_p};
  unsigned n_nodes = 0, n_tris = 0, n_rays = 0;
  // wave-uniform cursor over the wave's share: virtual entry v -> queue entry ((v / 64) * n_waves + wave) * 64 + v % 64
  unsigned cursor = 0;
  bool exhausted = (unsigned long long)wave * 64ull >= count;
  // per-lane ray state
  bool active = false, found = false;
  unsigned pid = 0; float dw = 0.0f;
  Ray ray; ray.o = ray.d = mk3(0, 0, 0); ray.t_max = 0.0f;
  f3 inv_dir = mk3(0, 0, 0); int neg_x = 0, neg_y = 0, neg_z = 0; RayPre rp = ray_pre(ray);
  int sp = 0, cur = 0, prim = -1; TriHit hit; hit.t = kInf; hit.b0 = hit.b1 = hit.b2 = 0.0f;

  auto finish = [&]() {  // the lane's ray is complete: write its result (same epilogue as k_trace)
    if (ANY) trace_write_any(lacc, ls, direct_add, as, occluded, os, pid, dw, found);
    else hits[pid * hs] = make_float4(hit_b2 ? hit.b2 : (found ? hit.t : kInf), __int_as_float(found ? prim : -1), hit.b0, hit.b1);
    active = false;
  };

  for (;;) {
    const unsigned long long idle = __ballot(!active);
    if (!exhausted && (unsigned)__popcll(idle) >= (idle == ~0ull ? 1u : (unsigned)RT_REFILL_MIN)) {
      const unsigned v = cursor + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      const unsigned long long e = ((unsigned long long)(v >> 6) * n_waves + wave) * 64ull + (v & 63u);
      if (!active && e < count) {
        pid = queue ? qv.get((unsigned)e) : (unsigned)e;
        const float4 o4 = ray_o[pid * rs], d4 = ray_d[pid * rs];
        ray.o = mk3(o4.x, o4.y, o4.z); ray.d = mk3(d4.x, d4.y, d4.z); ray.t_max = o4.w; dw = d4.w;
        inv_dir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        neg_x = inv_dir.x < 0.0f; neg_y = inv_dir.y < 0.0f; neg_z = inv_dir.z < 0.0f;
        rp = ray_pre(ray);
        sp = 0; cur = 0; prim = -1; found = false; hit.t = kInf; hit.b0 = hit.b1 = hit.b2 = 0.0f;
        active = true; n_rays += 1;
      }
      cursor += (unsigned)__popcll(idle);
      exhausted = ((unsigned long long)(cursor >> 6) * n_waves + wave) * 64ull + (cursor & 63u) >= count;  // entries grow with the cursor
    }
    if (__ballot(active) == 0ull) { if (exhausted) break; else continue; }
    // ---- descend to the next leaf whose box the ray enters
    int leaf_off = 0, leaf_n = 0;
    while (active && leaf_n == 0) {
      float4 n0, n1;
      src.node(cur, n0, n1);
      if (COUNT) n_nodes += 1;
      if (slab_test(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z)) {
        const unsigned packed = __float_as_uint(n1.w);
        const int n_prims = (int)(packed & 0xffffu);
        const int offset = __float_as_int(n1.z);
        if (n_prims > 0) { leaf_off = offset; leaf_n = n_prims; break; }
        const int axis = (int)((packed >> 16) & 0xffu);
        const int neg = axis == 0 ? neg_x : (axis == 1 ? neg_y : neg_z);
        if (neg) { stack[(sp++) * BLOCK] = cur + 1; cur = offset; }
        else { stack[(sp++) * BLOCK] = offset; cur = cur + 1; }
      } else {
        if (sp == 0) { finish(); break; }
        cur = stack[(--sp) * BLOCK];
      }
    }
    // ---- test the leaf's triangles
    if (active) {
      for (int i = 0; i < leaf_n; ++i) {
        if (COUNT) n_tris += 1;
        TriHit h;
        if (GENERAL) {
          int hp = 0; float th = 0.0f; unsigned nn = 0, ntt = 0;
          const bool hitp = general_leaf_prim<ANY, COUNT, int, true, true, true, true>(sc, sc.tri_p, leaf_off + i, ray, rp, ANY && io.shadow_masks != 0, stack + sp * BLOCK, BLOCK, h, hp, th, nn, ntt);
          if (COUNT) { n_nodes += nn; n_tris += ntt; }
          if (!hitp) continue;
          found = true;
          if (ANY) break;
          ray.t_max = th; prim = hp; hit = h;
          continue;
        }
        f3 p0, p1, p2;
        src.tri(leaf_off + i, p0, p1, p2);
        if (tri_test_pre(p0, p1, p2, ray, rp, h)) {
          found = true;
          if (ANY) break;
          ray.t_max = h.t; prim = leaf_off + i; hit = h;  // `.or(result)`: later accepted hits replace
        }
      }
      if ((ANY && found) || sp == 0) finish();
      else cur = stack[(--sp) * BLOCK];
    }
  }
  if (stats) {
    for (int off = 32; off > 0; off >>= 1) { n_rays += __shfl_down(n_rays, off); if (COUNT) { n_nodes += __shfl_down(n_nodes, off); n_tris += __shfl_down(n_tris, off); } }
    if (lane == 0u) {
      if (n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
      if (COUNT) { atomicAdd(&stats[st_nodes], (unsigned long long)n_nodes); atomicAdd(&stats[st_tris], (unsigned long long)n_tris); }
    }
  }
}

// ---- child-pair traversal (HBM scenes, frames that do not count visits). The reference visits one node per
// step: fetch its box, test, then pop or descend (rc/bvh/mod.rs:381-425) - every step is a dependent round trip
// to memory, and on a scene that does not fit the caches the kernel is bound by exactly that latency. Here an
// interior node P is expanded in one step: a 64-byte record (DScene::pairs, built at rt_scene_create) holds the
// boxes of both children and what each child is; the near child's test is the reference's test at that moment,
// the far child's slab arithmetic is done now and its only ray-state-dependent clause, tmin < t_max, is
// re-evaluated with the then-current t_max when the entry is popped (t_max only shrinks, so a far child that
// fails now is never pushed). The decisions, the order of triangle tests and therefore the hit record are
// identical to the reference's; the number of dependent fetches per ray roughly halves.
//   child code: bit 31 clear: interior node, bits 0-28 node index, bits 29-30 split axis
//               bit 31 set  : leaf, bits 0-25 first triangle, bits 26-30 triangle count - 1
RT_DEV bool slab_geom(float4 n0, float4 n1, const Ray& ray, f3 inv_dir, int neg_x, int neg_y, int neg_z, float& tmin_out) {
  // Bounds3::intersect_p_fast (bounds.rs:127-157) without its `tmin < ray.t_max` clause; same operations, predicates combined (see slab_test)
  float bx0 = neg_x ? n0.w : n0.x, bx1 = neg_x ? n0.x : n0.w;
  float by0 = neg_y ? n1.x : n0.y, by1 = neg_y ? n0.y : n1.x;
  float bz0 = neg_z ? n1.y : n0.z, bz1 = neg_z ? n0.z : n1.y;
  float tmin = (bx0 - ray.o.x) * inv_dir.x;
  float tmax = (bx1 - ray.o.x) * inv_dir.x;
  float tymin = (by0 - ray.o.y) * inv_dir.y;
  float tymax = (by1 - ray.o.y) * inv_dir.y;
  const bool miss_xy = (tmin > tymax) | (tymin > tmax);
  tmin = tymin > tmin ? tymin : tmin;
  tmax = tymax < tmax ? tymax : tmax;
  float tzmin = (bz0 - ray.o.z) * inv_dir.z;
  float tzmax = (bz1 - ray.o.z) * inv_dir.z;
  const bool miss_z = (tmin > tzmax) | (tzmin > tmax);
  tmin = tzmin > tmin ? tzmin : tmin;
  tmax = tzmax < tmax ? tzmax : tmax;
  tmin_out = tmin;
  return !miss_xy & !miss_z & (tmax > 0.0f);
}
#define RT_PAIR_LEAF 0x80000000u
#define RT_PAIR_TOP 0x10000000u  // (k_trace_top: the child is one of the LDS-resident top records, named by its slot)

// An object's tree (TransformedPrimitive::intersect / intersect_p -> the object's BVH, rc/primitive.rs:90-101) walked as child pairs: the steps of
// pair_interior_step / pair_leaf_step / pair_pop for ONE ray, run to completion inside the top level's leaf step. Objects hold plain triangles. r.t_max shrinks
// with every accepted hit (closest hit); the far child's deferred tmin goes to this lane's column of the tmin array, above the top level's pending entries.
template <bool ANY, class StackT>
RT_DEV bool nested_pair_walk(const float4* __restrict__ pairs, const float4* __restrict__ nodes, const float4* __restrict__ tri_p, Ray& r, StackT* stack, int stack_stride,
                             float* tstack, size_t grid_lanes, int& prim_out, TriHit& hit_out) {
  const f3 inv_dir = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
  const int neg_x = inv_dir.x < 0.0f, neg_y = inv_dir.y < 0.0f, neg_z = inv_dir.z < 0.0f;
  const RayPre rp = ray_pre(r);
  unsigned cur;
  {
    const float4 r0 = nodes[0], r1 = nodes[1];  // the root is the one node tested on its own
    if (!slab_test(r0, r1, r, inv_dir, neg_x, neg_y, neg_z)) return false;
    const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
    cur = np > 0u ? (0x80000000u | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : (((packed >> 16) & 0xffu) << 29);
  }
  int sp = 0; bool found = false;
  for (;;) {
    bool pop = false;
    if (cur & 0x80000000u) {
      const int off = (int)(cur & 0x03ffffffu), n = (int)((cur >> 26) & 31u) + 1;
      for (int i = 0; i < n; ++i) {
        f3 p0, p1, p2; load_tri(tri_p, off + i, p0, p1, p2);
        TriHit h;
        if (tri_test_pre(p0, p1, p2, r, rp, h)) {
          found = true;
          if (ANY) return true;
          r.t_max = h.t; prim_out = off + i; hit_out = h;  // `.or(result)`: later accepted hits replace
        }
      }
      pop = true;
    } else {
      const unsigned P = cur & 0x1fffffffu, axis = (cur >> 29) & 3u;
      const float4 a0 = pairs[4 * (size_t)P], a1 = pairs[4 * (size_t)P + 1], b0 = pairs[4 * (size_t)P + 2], b1 = pairs[4 * (size_t)P + 3];
      const bool neg = (((neg_x ? 1u : 0u) | (neg_y ? 2u : 0u) | (neg_z ? 4u : 0u)) >> axis) & 1u;
      const float4 n0 = neg ? b0 : a0, n1 = neg ? b1 : a1, f0 = neg ? a0 : b0, f1 = neg ? a1 : b1;
      const unsigned code_a = __float_as_uint(a1.z), code_b = __float_as_uint(a1.w);
      const unsigned code_n = neg ? code_b : code_a, code_f = neg ? code_a : code_b;
      float tmin_n = 0.0f, tmin_f = 0.0f;
      const bool hit_n = slab_geom(n0, n1, r, inv_dir, neg_x, neg_y, neg_z, tmin_n) && tmin_n < r.t_max;
      const bool keep_f = slab_geom(f0, f1, r, inv_dir, neg_x, neg_y, neg_z, tmin_f) && tmin_f < r.t_max;
      if (hit_n) {
        if (keep_f) { stack[sp * stack_stride] = (StackT)code_f; if (!ANY) tstack[(size_t)sp * grid_lanes] = tmin_f; ++sp; }
        cur = code_n;
      } else if (keep_f) cur = code_f;
      else pop = true;
    }
    if (pop) {
      for (;;) {
        if (sp == 0) return found;
        --sp;
        const unsigned c = (unsigned)stack[sp * stack_stride];
        if (ANY || tstack[(size_t)sp * grid_lanes] < r.t_max) { cur = c; break; }
      }
    }
  }
}

// per-lane traversal state and the output arrays a finished ray is written to
// RT_TTOP (round 5): a closest-hit pop has to compare the entry's deferred tmin with the t_max of the moment. The tmins live in HBM ([depth][lane of the grid]:
// LDS holds the codes), so every pop - most of them dead: the far subtree behind a hit - waited one memory round trip before it could even ask for its node.
// Now the top entry's tmin is ALSO kept in a register: a pop compares the register and requests the new top's tmin at once, which the next pop finds arrived.
#ifndef RT_TTOP
#define RT_TTOP 1
#endif
struct PairLane {
  bool active, found; unsigned pid; float dw;
  // registers are the currency of this kernel (86 -> 80 VGPRs is one more wave per SIMD): the direction signs are read off inv_dir where
  // they are used (a compare either way) and of the watertight test's permutation only kz is kept (kx, ky follow from it)
  Ray ray; f3 inv_dir; float sx, sy, sz;
  // finite(): the ray's reciprocal direction has no infinite component (inv_dir_finite) - its node tests are minima / maxima (pair_slabs). Kept as bit 2 of
  // kz (set: NOT finite), no register of its own; set_inv goes before set_rp.
  RT_DEV bool finite() const { return kz < 4; }
  RT_DEV void set_inv(f3 v) { inv_dir = v; kz = inv_dir_finite(v) ? 0 : 4; }
  RT_DEV int neg_x() const { return inv_dir.x < 0.0f; }
  RT_DEV int neg_y() const { return inv_dir.y < 0.0f; }
  RT_DEV int neg_z() const { return inv_dir.z < 0.0f; }
  // the sign along a node's split axis as arithmetic on the three compares: written as a select between the components, the optimiser turns it into
  // an indexed load from a private copy of inv_dir - one scratch access per node visit
  RT_DEV bool neg_axis(unsigned axis) const { return ((((inv_dir.x < 0.0f) ? 1u : 0u) | ((inv_dir.y < 0.0f) ? 2u : 0u) | ((inv_dir.z < 0.0f) ? 4u : 0u)) >> axis) & 1u; }
  RT_DEV RayPre rp() const { RayPre r; r.kz = kz & 3; r.kx = r.kz + 1; if (r.kx == 3) r.kx = 0; r.ky = r.kx + 1; if (r.ky == 3) r.ky = 0; r.sx = sx; r.sy = sy; r.sz = sz; return r; }
  RT_DEV void set_rp(const RayPre& r) { kz = (kz & 4) | r.kz; sx = r.sx; sy = r.sy; sz = r.sz; }
  int sp, prim; unsigned cur; TriHit hit;
  float ttop;  // closest hit: the deferred tmin of the entry on TOP of the to-visit stack (RT_TTOP), so that a pop compares a register and the load of the next entry's tmin is in flight long before the next pop
  int kz;  // (apart from inv_dir: written together as neighbours, the two become one 16-byte store to a private copy of the lane's state - 24 bytes of scratch per lane)
};
// The two children of a pair record against the lane's ray: slab_geom && tmin < t_max for each. A finite ray (all but a few hundred of a frame) takes the
// min / max form of the node test (slab_interval_finite: 14 instructions per box instead of 30, no sign selects); the branch is skipped by waves without
// the other kind.
// RT_ANY_ORDER (occlusion rays only: intersect_p's answer - "some primitive passes its test and every box above it passes the node test" - does not depend on
// the order of the walk, and the kernels of HBM scenes wait for memory, so a walk that reaches an occluder after fewer fetches is a faster one): which of two
// children that are both hit is entered first. S4 per frame, shadow / environment MIS rays:
//   0 the reference's order (the child on the ray's side of the split)               478 / 799 ms
//   2 the box entered earlier                                                         431 / 784
//   5 a leaf before an interior node, otherwise the box entered LATER                 434 / 760
//   1 the box the ray crosses for the longer stretch of [0, t_max]                    391 / 729
//   3 a leaf before an interior node, otherwise as 1                                  379 / 722   (+ "the box that holds the origin" before that: 391 / 745)
//   6 a leaf before an interior node, otherwise the box the ray LEAVES later          367 / 701
//   8 the box the ray leaves later (min(tmax, t_max))                                 361 / 690   <- default: what ends a ray inside a scene lies far along it
// The four-wide kernel keeps the reference's order: its children sorted by stretch made S3's shadow rays slower (100.8 -> 112.2 ms), its leaves first too
// (108.1), and sorted by where the ray leaves them as well (114.8).
#ifndef RT_ANY_ORDER
#define RT_ANY_ORDER 8
#endif
struct PairTest { bool hit_n, keep_f, far_first; float tmin_n, tmin_f; };  // (by value: as reference parameters the three flags became a byte array in scratch, indexed by one of them)
RT_DEV PairTest pair_slabs(const bool finite, const f3 o, const float t_max, const f3 inv_dir, float4 n0, float4 n1, float4 f0, float4 f1, const bool want_order) {
  PairTest r; r.far_first = false; r.tmin_n = 0.0f; r.tmin_f = 0.0f;
  if (__builtin_expect(finite, 1)) {
    float tmax_n, tmax_f;
    slab_interval_finite(n0, n1, o, inv_dir, r.tmin_n, tmax_n);
    slab_interval_finite(f0, f1, o, inv_dir, r.tmin_f, tmax_f);
    r.hit_n = (r.tmin_n <= tmax_n) & (tmax_n > 0.0f) & (r.tmin_n < t_max);
    r.keep_f = (r.tmin_f <= tmax_f) & (tmax_f > 0.0f) & (r.tmin_f < t_max);
    if (want_order) {
      if (RT_ANY_ORDER == 1 || RT_ANY_ORDER == 3) r.far_first = (fminf(tmax_f, t_max) - fmaxf(r.tmin_f, 0.0f)) > (fminf(tmax_n, t_max) - fmaxf(r.tmin_n, 0.0f));
      else if (RT_ANY_ORDER == 2) r.far_first = r.tmin_f < r.tmin_n;
      else if (RT_ANY_ORDER == 5) r.far_first = r.tmin_f > r.tmin_n;                                                 // (leaf first, then) the box entered LATER
      else if (RT_ANY_ORDER == 6 || RT_ANY_ORDER == 8) r.far_first = fminf(tmax_f, t_max) > fminf(tmax_n, t_max);  // (6: leaf first, then) the box left later
    }
  } else {
    Ray ray; ray.o = o; ray.d = mk3(0, 0, 0); ray.t_max = t_max;  // (the node test reads the origin and t_max)
    const int neg_x = inv_dir.x < 0.0f, neg_y = inv_dir.y < 0.0f, neg_z = inv_dir.z < 0.0f;
    r.hit_n = slab_geom(n0, n1, ray, inv_dir, neg_x, neg_y, neg_z, r.tmin_n) && r.tmin_n < t_max;
    r.keep_f = slab_geom(f0, f1, ray, inv_dir, neg_x, neg_y, neg_z, r.tmin_f) && r.tmin_f < t_max;
  }
  return r;
}
RT_DEV bool lane_slab_test(const bool finite, const f3 o, const float t_max, const f3 inv_dir, float4 n0, float4 n1) {  // slab_test for the lane's ray (the root)
  if (__builtin_expect(finite, 1)) return slab_test_finite(n0, n1, o, t_max, inv_dir);
  Ray ray; ray.o = o; ray.d = mk3(0, 0, 0); ray.t_max = t_max;
  return slab_test(n0, n1, ray, inv_dir, inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f);
}
// (the leaf phase of these loops is held back until enough lanes wait at a leaf: leaf_phase_now, rtx_dev_scene.h)
template <bool ANY>
RT_DEV void pair_finish(PairLane& L, const TraceOut& o) {  // same epilogue as k_trace
  if (ANY) trace_write_any(o.lacc, o.ls, o.direct_add, o.as, o.occluded, o.os, L.pid, L.dw, L.found);
  else o.hits[L.pid * o.hs] = make_float4(o.hit_b2 ? L.hit.b2 : (L.found ? L.hit.t : kInf), __int_as_float(L.found ? L.prim : -1), L.hit.b0, L.hit.b1);
  L.active = false;
}
// next pending entry that still passes tmin < t_max, or the ray is complete
template <bool ANY, int BLOCK>
RT_DEV void pair_pop(PairLane& L, const TraceOut& o, const unsigned* stack, const float* tstack, size_t grid_lanes) {
  for (;;) {
    if (L.sp == 0) { pair_finish<ANY>(L, o); return; }
    --L.sp;
    const unsigned c = stack[L.sp * BLOCK];
    if (ANY) { L.cur = c; return; }  // t_max never changes: the test at push time stands
    if (RT_TTOP) {
      const float t = L.ttop;
      if (L.sp > 0) L.ttop = tstack[(size_t)(L.sp - 1) * grid_lanes];
      if (t < L.ray.t_max) { L.cur = c; return; }
    } else if (tstack[(size_t)L.sp * grid_lanes] < L.ray.t_max) { L.cur = c; return; }
  }
}
template <bool ANY, int BLOCK>
RT_DEV void pair_interior_step(PairLane& L, const TraceOut& o, const float4* __restrict__ pairs, unsigned* stack, float* tstack, size_t grid_lanes) {
  const unsigned P = L.cur & 0x1fffffffu, axis = (L.cur >> 29) & 3u;
  const float4 a0 = pairs[4 * (size_t)P], a1 = pairs[4 * (size_t)P + 1], b0 = pairs[4 * (size_t)P + 2], b1 = pairs[4 * (size_t)P + 3];
  const bool neg = L.neg_axis(axis);
  // reference: negative direction along the split axis => second child first (bvh/mod.rs:411-417)
  const float4 n0 = neg ? b0 : a0, n1 = neg ? b1 : a1, f0 = neg ? a0 : b0, f1 = neg ? a1 : b1;
  const unsigned code_a = __float_as_uint(a1.z), code_b = __float_as_uint(a1.w);
  const unsigned code_n = neg ? code_b : code_a, code_f = neg ? code_a : code_b;
  const PairTest r = pair_slabs(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, n0, n1, f0, f1, ANY && RT_ANY_ORDER != 0);
  bool ff = ANY && r.far_first;
  if (ANY && RT_ANY_ORDER >= 3 && RT_ANY_ORDER != 8 && ((code_n ^ code_f) & RT_PAIR_LEAF) != 0u) ff = (code_f & RT_PAIR_LEAF) != 0u;  // exactly one child is a leaf: its primitives first
  const unsigned first = ff ? code_f : code_n, second = ff ? code_n : code_f;  // (closest hit: always near, far)
  if (r.hit_n & r.keep_f) { stack[L.sp * BLOCK] = second; if (!ANY) { tstack[(size_t)L.sp * grid_lanes] = r.tmin_f; L.ttop = r.tmin_f; } ++L.sp; L.cur = first; }
  else if (r.hit_n) L.cur = code_n;
  else if (r.keep_f) L.cur = code_f;
  else pair_pop<ANY, BLOCK>(L, o, stack, tstack, grid_lanes);
}
// GENERAL scenes: a leaf that holds anything but plain triangles carries RT_PAIR_GENERAL in its code (first primitive then in bits 0-24) and walks its
// primitives through general_leaf_prim; every other leaf of such a scene, and every leaf of a plain scene, runs the bare triangle loop.
#define RT_PAIR_GENERAL 0x02000000u
template <bool ANY, int GENERAL, bool INSTANCES = true>
RT_DEV bool pair_leaf_prims(PairLane& L, const DScene& sc, const float4* __restrict__ tri_p, bool shadow_masks, unsigned* nested_stack, int stack_stride, NestedCtx nc = NestedCtx{nullptr, nullptr, 0}) {
  const int off = (int)(L.cur & (GENERAL ? 0x01ffffffu : 0x03ffffffu)), n = (int)((L.cur >> 26) & 31u) + 1;
  if (GENERAL && (L.cur & RT_PAIR_GENERAL)) {
    for (int i = 0; i < n; ++i) {
      TriHit h; int hp = 0; float th = 0.0f; unsigned nn = 0, ntt = 0;
      if (!general_leaf_prim<ANY, false, unsigned, GENERAL == RT_GEN_ALL, GENERAL != RT_GEN_INSTANCES_ONLY, INSTANCES>(sc, tri_p, off + i, L.ray, L.rp(), shadow_masks, nested_stack, stack_stride, h, hp, th, nn, ntt, nc)) continue;
      L.found = true;
      if (ANY) break;
      L.ray.t_max = th; L.prim = hp; L.hit = h;
    }
    return L.found;
  }
  for (int i = 0; i < n; ++i) {
    f3 p0, p1, p2;
    load_tri(tri_p, off + i, p0, p1, p2);
    TriHit h;
    if (tri_test_pre(p0, p1, p2, L.ray, L.rp(), h)) {
      L.found = true;
      if (ANY) break;
      L.ray.t_max = h.t; L.prim = off + i; L.hit = h;  // `.or(result)`: later accepted hits replace
    }
  }
  return L.found;
}
template <bool ANY, int BLOCK, int GENERAL = 0>
RT_DEV void pair_leaf_step(PairLane& L, const TraceOut& o, const DScene& sc, const float4* __restrict__ tri_p, unsigned* stack, const float* tstack, size_t grid_lanes, bool shadow_masks) {
  (void)pair_leaf_prims<ANY, GENERAL>(L, sc, tri_p, shadow_masks, stack + L.sp * BLOCK, BLOCK,
                                      NestedCtx{sc.obj_pairs ? sc.pairs : nullptr, const_cast<float*>(tstack) + (size_t)L.sp * grid_lanes, grid_lanes});
  if (ANY && L.found) pair_finish<ANY>(L, o); else pair_pop<ANY, BLOCK>(L, o, stack, tstack, grid_lanes);
}

// (closest hit with quadrics AND the nested walk: 154 VGPRs at three waves; bound to four it would spill 24 dwords)
template <bool ANY, int BLOCK, int DEPTH, int GENERAL = 0>
__global__ void __launch_bounds__(BLOCK, (GENERAL == RT_GEN_NO_MASKS && !ANY) ? 3 : RT_GEN_MIN_WAVES(GENERAL)) k_trace_pair(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                      unsigned count_static, unsigned long long* stats, int st_rays, float* __restrict__ tmin_stack_mem, unsigned refill_min) {
  const RT_SPTR_R(const float4) ray_o = io.ray_o; const RT_SPTR_R(const float4) ray_d = io.ray_d; const size_t rs = io.ray_stride;
  __shared__ unsigned stack_mem[DEPTH * BLOCK];
  unsigned* const stack = stack_mem + threadIdx.x;
  // the deferred tmin of each stack entry lives in HBM, [depth][lane of the grid]: a push is a fire-and-forget
  // store and a pop's load is the only extra latency
  const size_t grid_lanes = (size_t)gridDim.x * BLOCK;
  float* const tstack = tmin_stack_mem + (size_t)blockIdx.x * BLOCK + threadIdx.x;
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  const unsigned lane = __lane_id();
  const unsigned n_waves = gridDim.x * (BLOCK / 64), wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const float4* __restrict__ pairs = sc.pairs; const float4* __restrict__ tri_p = sc.tri_p; const float4* __restrict__ nodes = sc.nodes;
  const TraceOut out = trace_out_of(io);
  const unsigned leaf_min = refill_min >> 8; refill_min &= 0xffu;  // the launch's two knobs travel in one word
  unsigned n_rays = 0;
  unsigned cursor = 0;
  bool exhausted = (unsigned long long)wave * 64ull >= count;
  PairLane L;
  L.active = false; L.found = false; L.pid = 0; L.dw = 0.0f;
  L.ray.o = L.ray.d = mk3(0, 0, 0); L.ray.t_max = 0.0f; L.set_inv(mk3(0, 0, 0)); L.set_rp(ray_pre(L.ray));
  L.sp = 0; L.prim = -1; L.cur = 0; L.ttop = 0.0f; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;

  for (;;) {
    const unsigned long long idle = __ballot(!L.active);
    if (!exhausted && (unsigned)__popcll(idle) >= (idle == ~0ull ? 1u : refill_min)) {
      const unsigned v = cursor + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      const unsigned long long e = ((unsigned long long)(v >> 6) * n_waves + wave) * 64ull + (v & 63u);
      if (!L.active && e < count) {
        L.pid = queue ? qv.get((unsigned)e) : (unsigned)e;
        const float4 o4 = ray_o[L.pid * rs], d4 = ray_d[L.pid * rs];
        L.ray.o = mk3(o4.x, o4.y, o4.z); L.ray.d = mk3(d4.x, d4.y, d4.z); L.ray.t_max = o4.w; L.dw = d4.w;
        L.set_inv(mk3(1.0f / L.ray.d.x, 1.0f / L.ray.d.y, 1.0f / L.ray.d.z));
        L.set_rp(ray_pre(L.ray));
        L.sp = 0; L.prim = -1; L.found = false; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
        L.active = true; n_rays += 1;
        // the root is the one node tested on its own
        const float4 r0 = nodes[0], r1 = nodes[1];
        if (lane_slab_test(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, r0, r1)) {
          const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
          L.cur = np > 0u ? (RT_PAIR_LEAF | (GENERAL ? RT_PAIR_GENERAL : 0u) | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : (((packed >> 16) & 0xffu) << 29);
        } else pair_finish<ANY>(L, out);
      }
      cursor += (unsigned)__popcll(idle);
      exhausted = ((unsigned long long)(cursor >> 6) * n_waves + wave) * 64ull + (cursor & 63u) >= count;
    }
    if (__ballot(L.active) == 0ull) { if (exhausted) break; else continue; }
    const bool shadow_masks = ANY && io.shadow_masks != 0;
    {
      const bool at_leaf = L.active && (L.cur & RT_PAIR_LEAF) != 0u;
      const bool leaves_now = leaf_phase_now(L.active, at_leaf, leaf_min);
      if (L.active) {
        if (at_leaf) { if (leaves_now) pair_leaf_step<ANY, BLOCK, GENERAL>(L, out, sc, tri_p, stack, tstack, grid_lanes, shadow_masks); }
        else pair_interior_step<ANY, BLOCK>(L, out, pairs, stack, tstack, grid_lanes);
      }
    }
  }
  if (stats) {
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
  }
}

// ---- two-level traversal as ONE loop (round 4). In k_trace_pair<.., RT_GEN_INSTANCES_ONLY> an object instance is a primitive of a top-level
// leaf and its whole walk (nested_pair_walk) runs inside that lane's leaf step: the wave waits until every lane holds a leaf, then until the LONGEST of up to
// 4 x 64 nested walks is over - 10 000 placements of a 1280-triangle object traced at 600 Msamples/s against 729 for the same triangles written out. Here a
// lane that reaches an instance ENTERS it: the ray goes to object space (Transform * Ray, rc/ray.rs:83-93), the world ray waits in LDS, the rest of the
// top-level leaf waits on the lane's stack as one more entry, and the lane carries on in the main loop - its interior steps are the same instructions as a
// top-level lane's (other record base), so lanes on different levels share the wave's steps. When the object's entries are used up the lane restores its world
// ray and pops on. Per ray the sequence of box tests, triangle tests and t_max updates is TransformedPrimitive::intersect's (primitive.rs:90-101) inside
// BVH::intersect's: hit records and occlusion results bit-equal to the nested form (tests/test_gpu_instances.py).
// Scenes: object instances over plain triangles, objects with pair records (DScene::obj_pairs) - the other general scenes keep k_trace_pair / k_trace_quad.
#define RT_INST_NONE 0xffffffffu
template <bool ANY, int BLOCK, int DEPTH>
__global__ void __launch_bounds__(BLOCK, 4) k_trace_inst(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                         unsigned count_static, unsigned long long* stats, int st_rays, float* __restrict__ tmin_stack_mem, unsigned refill_min) {
  const RT_SPTR_R(const float4) ray_o = io.ray_o; const RT_SPTR_R(const float4) ray_d = io.ray_d; const size_t rs = io.ray_stride;
  __shared__ unsigned stack_mem[DEPTH * BLOCK];
  __shared__ float s_world[13 * BLOCK];  // the world-space ray of a lane that is inside an instance: o, d, 1 / d, the watertight test's shear
  unsigned* const stack = stack_mem + threadIdx.x;
  float* const wsave = s_world + threadIdx.x;
  const size_t grid_lanes = (size_t)gridDim.x * BLOCK;
  float* const tstack = tmin_stack_mem + (size_t)blockIdx.x * BLOCK + threadIdx.x;
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  const unsigned lane = __lane_id();
  const unsigned n_waves = gridDim.x * (BLOCK / 64), wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const float4* __restrict__ pairs = sc.pairs; const float4* __restrict__ tri_p = sc.tri_p; const float4* __restrict__ nodes = sc.nodes;
  const TraceOut out = trace_out_of(io);
  const unsigned leaf_min = refill_min >> 8; refill_min &= 0xffu;
  unsigned n_rays = 0, cursor = 0;
  bool exhausted = (unsigned long long)wave * 64ull >= count;
  PairLane L;
  L.active = false; L.found = false; L.pid = 0; L.dw = 0.0f;
  L.ray.o = L.ray.d = mk3(0, 0, 0); L.ray.t_max = 0.0f; L.set_inv(mk3(0, 0, 0)); L.set_rp(ray_pre(L.ray));
  L.sp = 0; L.prim = -1; L.cur = 0; L.ttop = 0.0f; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
  unsigned inst = RT_INST_NONE, node_base = 0u, prim_base = 0u, id_base = 0u; int sp_base = -1;  // the instance the lane is inside, its records, the stack height it was entered at

  // next pending entry that still passes tmin < t_max; an object whose entries are used up is left first; no entry left: the ray is complete
  auto pop = [&]() {
    for (;;) {
      if (inst != RT_INST_NONE && L.sp == sp_base) {
        L.ray.o = mk3(wsave[0], wsave[BLOCK], wsave[2 * BLOCK]); L.ray.d = mk3(wsave[3 * BLOCK], wsave[4 * BLOCK], wsave[5 * BLOCK]);
        L.set_inv(mk3(wsave[6 * BLOCK], wsave[7 * BLOCK], wsave[8 * BLOCK]));
        L.kz = __float_as_int(wsave[9 * BLOCK]); L.sx = wsave[10 * BLOCK]; L.sy = wsave[11 * BLOCK]; L.sz = wsave[12 * BLOCK];
        inst = RT_INST_NONE; node_base = prim_base = id_base = 0u; sp_base = -1;
      }
      if (L.sp == 0) { pair_finish<ANY>(L, out); return; }
      --L.sp;
      const unsigned c = stack[L.sp * BLOCK];
      if (ANY) { L.cur = c; return; }
      if (RT_TTOP) {
        const float t = L.ttop;
        if (L.sp > 0) L.ttop = tstack[(size_t)(L.sp - 1) * grid_lanes];
        if (t < L.ray.t_max) { L.cur = c; return; }
      } else
      if (tstack[(size_t)L.sp * grid_lanes] < L.ray.t_max) { L.cur = c; return; }
    }
  };
  for (;;) {
    const unsigned long long idle = __ballot(!L.active);
    if (!exhausted && (unsigned)__popcll(idle) >= (idle == ~0ull ? 1u : refill_min)) {
      const unsigned v = cursor + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      const unsigned long long e = ((unsigned long long)(v >> 6) * n_waves + wave) * 64ull + (v & 63u);
      if (!L.active && e < count) {
        L.pid = queue ? qv.get((unsigned)e) : (unsigned)e;
        const float4 o4 = ray_o[L.pid * rs], d4 = ray_d[L.pid * rs];
        L.ray.o = mk3(o4.x, o4.y, o4.z); L.ray.d = mk3(d4.x, d4.y, d4.z); L.ray.t_max = o4.w; L.dw = d4.w;
        L.set_inv(mk3(1.0f / L.ray.d.x, 1.0f / L.ray.d.y, 1.0f / L.ray.d.z));
        L.set_rp(ray_pre(L.ray));
        L.sp = 0; L.prim = -1; L.found = false; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
        inst = RT_INST_NONE; node_base = prim_base = id_base = 0u; sp_base = -1;
        L.active = true; n_rays += 1;
        const float4 r0 = nodes[0], r1 = nodes[1];  // the root is the one node tested on its own
        if (lane_slab_test(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, r0, r1)) {
          const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
          L.cur = np > 0u ? (RT_PAIR_LEAF | RT_PAIR_GENERAL | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : (((packed >> 16) & 0xffu) << 29);
        } else pair_finish<ANY>(L, out);
      }
      cursor += (unsigned)__popcll(idle);
      exhausted = ((unsigned long long)(cursor >> 6) * n_waves + wave) * 64ull + (cursor & 63u) >= count;
    }
    if (__ballot(L.active) == 0ull) { if (exhausted) break; else continue; }
    const bool at_leaf = L.active && (L.cur & RT_PAIR_LEAF) != 0u;
    const bool leaves_now = leaf_phase_now(L.active, at_leaf, leaf_min);
    if (!L.active) continue;
    if (!at_leaf) {  // one child pair, at either level (pair_interior_step; an object's records sit behind the top level's, codes local to the object)
      const unsigned P = L.cur & 0x1fffffffu, axis = (L.cur >> 29) & 3u;
      const float4* __restrict__ rec = pairs + 4 * ((size_t)node_base + P);
      const float4 a0 = rec[0], a1 = rec[1], b0 = rec[2], b1 = rec[3];
      const bool neg = L.neg_axis(axis);
      const float4 n0 = neg ? b0 : a0, n1 = neg ? b1 : a1, f0 = neg ? a0 : b0, f1 = neg ? a1 : b1;
      const unsigned code_a = __float_as_uint(a1.z), code_b = __float_as_uint(a1.w);
      const unsigned code_n = neg ? code_b : code_a, code_f = neg ? code_a : code_b;
      const PairTest pt = pair_slabs(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, n0, n1, f0, f1, false);
      const bool hit_n = pt.hit_n, keep_f = pt.keep_f; const float tmin_f = pt.tmin_f;
      if (hit_n) {
        if (keep_f) { stack[L.sp * BLOCK] = code_f; if (!ANY) { tstack[(size_t)L.sp * grid_lanes] = tmin_f; L.ttop = tmin_f; } ++L.sp; }
        L.cur = code_n;
      } else if (keep_f) L.cur = code_f;
      else pop();
      continue;
    }
    if (!leaves_now) continue;
    if (inst != RT_INST_NONE) {  // a leaf of the object: plain triangles, in object space
      const int off = (int)(L.cur & 0x03ffffffu), n = (int)((L.cur >> 26) & 31u) + 1;
      const RayPre rp = L.rp();
      for (int i = 0; i < n; ++i) {
        f3 p0, p1, p2; load_tri(tri_p + 3 * (size_t)prim_base, off + i, p0, p1, p2);
        TriHit h;
        if (tri_test_pre(p0, p1, p2, L.ray, rp, h)) {
          L.found = true;
          if (ANY) break;
          L.ray.t_max = h.t; L.prim = (int)(id_base + (unsigned)(off + i)); L.hit = h;  // `.or(result)`: later accepted hits replace
        }
      }
      if (ANY && L.found) pair_finish<ANY>(L, out); else pop();
      continue;
    }
    {  // a leaf of the top level: triangles and instances, in order; entering an instance leaves the rest of the leaf on the stack
      const int off = (int)(L.cur & 0x01ffffffu), n = (int)((L.cur >> 26) & 31u) + 1;
      bool entered = false;
      for (int i = 0; i < n && !entered; ++i) {
        const float4 a = tri_p[3 * (size_t)(off + i)], b = tri_p[3 * (size_t)(off + i) + 1], c = tri_p[3 * (size_t)(off + i) + 2];
        if (!(__float_as_uint(c.w) & RT_FLAG_INSTANCE)) {
          TriHit h;
          if (tri_test_pre(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), L.ray, L.rp(), h)) {
            L.found = true;
            if (ANY) break;
            L.ray.t_max = h.t; L.prim = off + i; L.hit = h;
          }
          continue;
        }
        const unsigned k = __float_as_uint(c.x);
        const DInstance& in = sc.instances[k];
        Ray r; r.o = xf34_point(in.w2o, L.ray.o); r.d = xf34_vector(in.w2o, L.ray.d); r.t_max = L.ray.t_max;
        if (in.n_nodes == 0u) {  // an object of one primitive is wrapped as it is (api.rs:1073-1082): no node test, no walk
          f3 p0, p1, p2; load_tri(tri_p, (int)in.prim_base, p0, p1, p2);
          TriHit h;
          if (tri_test(p0, p1, p2, r, h)) {
            L.found = true;
            if (ANY) break;
            L.ray.t_max = h.t; L.prim = (int)in.id_base; L.hit = h;
          }
          continue;
        }
        const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        const float4 r0 = nodes[2 * (size_t)in.node_base], r1 = nodes[2 * (size_t)in.node_base + 1];  // the object's root: the one node tested on its own
        if (!slab_test(r0, r1, r, inv, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f)) continue;
        if (i + 1 < n) {  // the rest of this leaf: an entry that no t_max can discard (the reference's loop over the leaf's primitives goes on whatever was hit)
          stack[L.sp * BLOCK] = RT_PAIR_LEAF | RT_PAIR_GENERAL | (unsigned)(off + i + 1) | ((unsigned)(n - i - 2) << 26);
          if (!ANY) { tstack[(size_t)L.sp * grid_lanes] = -kInf; L.ttop = -kInf; }
          ++L.sp;
        }
        wsave[0] = L.ray.o.x; wsave[BLOCK] = L.ray.o.y; wsave[2 * BLOCK] = L.ray.o.z; wsave[3 * BLOCK] = L.ray.d.x; wsave[4 * BLOCK] = L.ray.d.y; wsave[5 * BLOCK] = L.ray.d.z;
        wsave[6 * BLOCK] = L.inv_dir.x; wsave[7 * BLOCK] = L.inv_dir.y; wsave[8 * BLOCK] = L.inv_dir.z;
        wsave[9 * BLOCK] = __int_as_float(L.kz); wsave[10 * BLOCK] = L.sx; wsave[11 * BLOCK] = L.sy; wsave[12 * BLOCK] = L.sz;
        L.ray.o = r.o; L.ray.d = r.d; L.set_inv(inv); L.set_rp(ray_pre(r));
        inst = k; node_base = in.node_base; prim_base = in.prim_base; id_base = in.id_base; sp_base = L.sp;
        const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
        L.cur = np > 0u ? (RT_PAIR_LEAF | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : (((packed >> 16) & 0xffu) << 29);
        entered = true;
      }
      if (entered) continue;
      if (ANY && L.found) pair_finish<ANY>(L, out); else pop();
    }
  }
  if (stats) {
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
  }
}

// ---- child-pair traversal with the top of the tree in LDS. k_trace_pair waits on memory two thirds of its wave cycles: every step is a dependent
// fetch of a 64-byte record from L2 (or beyond), and every ray starts at the root. Here the pair records of the first levels of the tree (up to
// RT_TOP_MAX interior nodes, breadth first; rt_scene_create) are copied into LDS once per workgroup and the steps that touch them never leave the
// CU; deeper nodes are fetched as before. To pay for the 16 KB the to-visit stack keeps only its first RT_TOP_LDS_DEPTH entries in LDS and spills
// deeper ones to an HBM array (rare: the stack is that deep only below the LDS levels of a path that kept its far children), and a workgroup is 512
// lanes so that three of them (48 KB each) fill a CU at six waves per SIMD. Records, steps and decisions are those of k_trace_pair: same hit records.
//   interior code with RT_PAIR_TOP set: bits 0-7 = LDS slot of the node's record (bits 29-30 split axis as before)
#ifndef RT_TOP_MAX
#define RT_TOP_MAX 256
#endif
#ifndef RT_TOP_LDS_DEPTH
#define RT_TOP_LDS_DEPTH 16
#endif

template <int BLOCK>
struct SplitStack {  // lds: this lane's column of [RT_TOP_LDS_DEPTH][BLOCK]; hbm: this lane's column of [deeper][grid lanes]
  unsigned* lds; unsigned* hbm; size_t grid_lanes;
  RT_DEV void put(int sp, unsigned v) const { if (sp < RT_TOP_LDS_DEPTH) lds[sp * BLOCK] = v; else hbm[(size_t)(sp - RT_TOP_LDS_DEPTH) * grid_lanes] = v; }
  RT_DEV unsigned get(int sp) const { return sp < RT_TOP_LDS_DEPTH ? lds[sp * BLOCK] : hbm[(size_t)(sp - RT_TOP_LDS_DEPTH) * grid_lanes]; }
};
template <bool ANY, int BLOCK>
RT_DEV void top_pop(PairLane& L, const TraceOut& o, const SplitStack<BLOCK>& stk, const float* tstack, size_t grid_lanes) {
  for (;;) {
    if (L.sp == 0) { pair_finish<ANY>(L, o); return; }
    --L.sp;
    const unsigned c = stk.get(L.sp);
    if (ANY) { L.cur = c; return; }
    if (RT_TTOP) {
      const float t = L.ttop;
      if (L.sp > 0) L.ttop = tstack[(size_t)(L.sp - 1) * grid_lanes];
      if (t < L.ray.t_max) { L.cur = c; return; }
    } else
    if (tstack[(size_t)L.sp * grid_lanes] < L.ray.t_max) { L.cur = c; return; }
  }
}
template <bool ANY, int BLOCK>
RT_DEV void top_interior_step(PairLane& L, const TraceOut& o, const float4* __restrict__ pairs, const float4* s_top, const SplitStack<BLOCK>& stk, float* tstack, size_t grid_lanes) {
  const unsigned axis = (L.cur >> 29) & 3u;
  float4 a0, a1, b0, b1;
  if (L.cur & RT_PAIR_TOP) { const unsigned k = L.cur & 0xffu; a0 = s_top[4 * k]; a1 = s_top[4 * k + 1]; b0 = s_top[4 * k + 2]; b1 = s_top[4 * k + 3]; }
  else { const size_t P = L.cur & 0x0fffffffu; a0 = pairs[4 * P]; a1 = pairs[4 * P + 1]; b0 = pairs[4 * P + 2]; b1 = pairs[4 * P + 3]; }
  const bool neg = L.neg_axis(axis);
  const float4 n0 = neg ? b0 : a0, n1 = neg ? b1 : a1, f0 = neg ? a0 : b0, f1 = neg ? a1 : b1;
  const unsigned code_a = __float_as_uint(a1.z), code_b = __float_as_uint(a1.w);
  const unsigned code_n = neg ? code_b : code_a, code_f = neg ? code_a : code_b;
  const PairTest r = pair_slabs(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, n0, n1, f0, f1, ANY && RT_ANY_ORDER != 0);
  bool ff = ANY && r.far_first;
  if (ANY && RT_ANY_ORDER >= 3 && RT_ANY_ORDER != 8 && ((code_n ^ code_f) & RT_PAIR_LEAF) != 0u) ff = (code_f & RT_PAIR_LEAF) != 0u;
  const unsigned first = ff ? code_f : code_n, second = ff ? code_n : code_f;
  if (r.hit_n & r.keep_f) { stk.put(L.sp, second); if (!ANY) { tstack[(size_t)L.sp * grid_lanes] = r.tmin_f; L.ttop = r.tmin_f; } ++L.sp; L.cur = first; }
  else if (r.hit_n) L.cur = code_n;
  else if (r.keep_f) L.cur = code_f;
  else top_pop<ANY, BLOCK>(L, o, stk, tstack, grid_lanes);
}
template <bool ANY, int BLOCK, int GENERAL = 0>  // GENERAL here: quadrics and masked triangles (an instanced scene needs a contiguous stack column: k_trace_pair)
RT_DEV void top_leaf_step(PairLane& L, const TraceOut& o, const DScene& sc, const float4* __restrict__ tri_p, const SplitStack<BLOCK>& stk, const float* tstack, size_t grid_lanes, bool shadow_masks) {
  (void)pair_leaf_prims<ANY, GENERAL, false>(L, sc, tri_p, shadow_masks, nullptr, 0);  // (an instanced scene needs a contiguous stack column: k_trace_pair)
  if (ANY && L.found) pair_finish<ANY>(L, o); else top_pop<ANY, BLOCK>(L, o, stk, tstack, grid_lanes);
}
template <bool ANY, int BLOCK, int GENERAL = 0>
__global__ void __launch_bounds__(BLOCK, GENERAL ? 4 : 6) k_trace_top(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                     unsigned count_static, unsigned long long* stats, int st_rays, float* __restrict__ tmin_stack_mem, unsigned* __restrict__ deep_stack_mem,
                                                     unsigned refill_min) {
  __shared__ unsigned stack_mem[RT_TOP_LDS_DEPTH * BLOCK];
  __shared__ float4 s_top[4 * RT_TOP_MAX];
  const RT_SPTR_R(const float4) ray_o = io.ray_o; const RT_SPTR_R(const float4) ray_d = io.ray_d; const size_t rs = io.ray_stride;
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  if ((unsigned long long)blockIdx.x * BLOCK >= count) return;  // (uniform per workgroup: nothing to stage the top of the tree for)
  for (unsigned i = threadIdx.x; i < 4u * sc.n_top; i += BLOCK) s_top[i] = sc.top_pairs[i];
  __syncthreads();
  const size_t grid_lanes = (size_t)gridDim.x * BLOCK;
  const size_t my_lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  float* const tstack = tmin_stack_mem + my_lane;
  const SplitStack<BLOCK> stk{stack_mem + threadIdx.x, deep_stack_mem + my_lane, grid_lanes};
  const unsigned lane = __lane_id();
  const unsigned n_waves = gridDim.x * (BLOCK / 64), wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const float4* __restrict__ pairs = sc.pairs; const float4* __restrict__ tri_p = sc.tri_p; const float4* __restrict__ nodes = sc.nodes;
  const TraceOut out = trace_out_of(io);
  const unsigned leaf_min = refill_min >> 8; refill_min &= 0xffu;  // the launch's two knobs travel in one word
  unsigned n_rays = 0, cursor = 0;
  bool exhausted = (unsigned long long)wave * 64ull >= count;
  PairLane L;
  L.active = false; L.found = false; L.pid = 0; L.dw = 0.0f;
  L.ray.o = L.ray.d = mk3(0, 0, 0); L.ray.t_max = 0.0f; L.set_inv(mk3(0, 0, 0)); L.set_rp(ray_pre(L.ray));
  L.sp = 0; L.prim = -1; L.cur = 0; L.ttop = 0.0f; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
  for (;;) {
    const unsigned long long idle = __ballot(!L.active);
    if (!exhausted && (unsigned)__popcll(idle) >= (idle == ~0ull ? 1u : refill_min)) {  // the refill scheme of k_trace_pair
      const unsigned v = cursor + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      const unsigned long long e = ((unsigned long long)(v >> 6) * n_waves + wave) * 64ull + (v & 63u);
      if (!L.active && e < count) {
        L.pid = queue ? qv.get((unsigned)e) : (unsigned)e;
        const float4 o4 = ray_o[L.pid * rs], d4 = ray_d[L.pid * rs];
        L.ray.o = mk3(o4.x, o4.y, o4.z); L.ray.d = mk3(d4.x, d4.y, d4.z); L.ray.t_max = o4.w; L.dw = d4.w;
        L.set_inv(mk3(1.0f / L.ray.d.x, 1.0f / L.ray.d.y, 1.0f / L.ray.d.z));
        L.set_rp(ray_pre(L.ray));
        L.sp = 0; L.prim = -1; L.found = false; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
        L.active = true; n_rays += 1;
        const float4 r0 = nodes[0], r1 = nodes[1];  // the root is the one node tested on its own
        if (lane_slab_test(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, r0, r1)) {
          const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
          L.cur = np > 0u ? (RT_PAIR_LEAF | (GENERAL ? RT_PAIR_GENERAL : 0u) | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : ((((packed >> 16) & 0xffu) << 29) | (sc.n_top ? RT_PAIR_TOP : 0u));  // root = slot 0
        } else pair_finish<ANY>(L, out);
      }
      cursor += (unsigned)__popcll(idle);
      exhausted = ((unsigned long long)(cursor >> 6) * n_waves + wave) * 64ull + (cursor & 63u) >= count;
    }
    if (__ballot(L.active) == 0ull) { if (exhausted) break; else continue; }
    {
      const bool at_leaf = L.active && (L.cur & RT_PAIR_LEAF) != 0u;
      const bool leaves_now = leaf_phase_now(L.active, at_leaf, leaf_min);
      if (L.active) {
        if (at_leaf) { if (leaves_now) top_leaf_step<ANY, BLOCK, GENERAL>(L, out, sc, tri_p, stk, tstack, grid_lanes, ANY && io.shadow_masks != 0); }
        else top_interior_step<ANY, BLOCK>(L, out, pairs, s_top, stk, tstack, grid_lanes);
      }
    }
  }
  if (stats) {
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
  }
}

// ---- four-wide any-hit traversal (HBM scenes). The tree is the reference's binary tree; a quad record holds, for an interior node,
// the boxes of its grandchildren (a child that is a leaf stands for itself), so one 128-byte fetch decides two binary levels and the
// chain of dependent node fetches per ray halves. Entries are entered in the order the signs of the ray direction along the three
// split axes involved give, i.e. in the binary loop's order, and a subtree is skipped under the same condition (a box contains its
// children's), so the walk visits the same leaves. Used for shadow rays only: measured on closest-hit rays it loses, because an ordered
// closest-hit walk usually discards the far subtree with one dead pop where this step has already pushed both of its halves, and
// each dead pop waits for its deferred tmin (S4 closest hit 150 -> 155 ms; any hit 66 -> 57 ms, S3 any hit -15 %).
RT_DEV bool slab_geom6(const float* __restrict__ b, const Ray& ray, f3 inv_dir, int neg_x, int neg_y, int neg_z, float& tmin_out) {  // slab_geom on (min.xyz, max.xyz)
  const float bx0 = neg_x ? b[3] : b[0], bx1 = neg_x ? b[0] : b[3];
  const float by0 = neg_y ? b[4] : b[1], by1 = neg_y ? b[1] : b[4];
  const float bz0 = neg_z ? b[5] : b[2], bz1 = neg_z ? b[2] : b[5];
  float tmin = (bx0 - ray.o.x) * inv_dir.x;
  float tmax = (bx1 - ray.o.x) * inv_dir.x;
  const float tymin = (by0 - ray.o.y) * inv_dir.y;
  const float tymax = (by1 - ray.o.y) * inv_dir.y;
  const bool miss_xy = (tmin > tymax) | (tymin > tmax);
  tmin = tymin > tmin ? tymin : tmin;
  tmax = tymax < tmax ? tymax : tmax;
  const float tzmin = (bz0 - ray.o.z) * inv_dir.z;
  const float tzmax = (bz1 - ray.o.z) * inv_dir.z;
  const bool miss_z = (tmin > tzmax) | (tzmin > tmax);
  tmin = tzmin > tmin ? tzmin : tmin;
  tmax = tzmax < tmax ? tzmax : tmax;
  tmin_out = tmin;
  return !miss_xy & !miss_z & (tmax > 0.0f);
}
template <bool ANY, int BLOCK>
RT_DEV void quad_interior_step(PairLane& L, const TraceOut& o, const float4* __restrict__ quads, unsigned* stack, float* tstack, size_t grid_lanes) {
  const unsigned P = L.cur & 0x1fffffffu, axis_p = (L.cur >> 29) & 3u;
  const float4* __restrict__ rec = quads + 8 * (size_t)P;
  float bx[24];
  const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
  {
    bx[0] = q0.x; bx[1] = q0.y; bx[2] = q0.z; bx[3] = q0.w; bx[4] = q1.x; bx[5] = q1.y; bx[6] = q1.z; bx[7] = q1.w;
    bx[8] = q2.x; bx[9] = q2.y; bx[10] = q2.z; bx[11] = q2.w; bx[12] = q3.x; bx[13] = q3.y; bx[14] = q3.z; bx[15] = q3.w;
    bx[16] = q4.x; bx[17] = q4.y; bx[18] = q4.z; bx[19] = q4.w; bx[20] = q5.x; bx[21] = q5.y; bx[22] = q5.z; bx[23] = q5.w;
  }
  const float4 qc = rec[6];
  const unsigned code[4] = {__float_as_uint(qc.x), __float_as_uint(qc.y), __float_as_uint(qc.z), __float_as_uint(qc.w)};
  const unsigned axes = __float_as_uint(rec[7].x);
  bool hit[4]; float tmin[4];
  if (__builtin_expect(L.finite(), 1)) {  // the min / max node test of a finite ray (slab_interval_finite)
    const f3 qo = mk3(L.ray.o.x, L.ray.o.y, L.ray.o.z), qinv = mk3(L.inv_dir.x, L.inv_dir.y, L.inv_dir.z); const float qt = L.ray.t_max;
    float tmax[4];
    slab_interval_finite_scalar(q0, q1, qo, qinv, tmin[0], tmax[0]);
    slab_interval_finite_scalar(make_float4(q1.z, q1.w, q2.x, q2.y), make_float4(q2.z, q2.w, 0.0f, 0.0f), qo, qinv, tmin[1], tmax[1]);
    slab_interval_finite_scalar(q3, q4, qo, qinv, tmin[2], tmax[2]);
    slab_interval_finite_scalar(make_float4(q4.z, q4.w, q5.x, q5.y), make_float4(q5.z, q5.w, 0.0f, 0.0f), qo, qinv, tmin[3], tmax[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) hit[k] = (code[k] != 0xffffffffu) & (tmin[k] <= tmax[k]) & (tmax[k] > 0.0f) & (tmin[k] < qt);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tmin[k] = 0.0f;
      hit[k] = (code[k] != 0xffffffffu) & slab_geom6(bx + 6 * k, L.ray, L.inv_dir, L.neg_x(), L.neg_y(), L.neg_z(), tmin[k]);
      hit[k] = hit[k] & (tmin[k] < L.ray.t_max);
    }
  }
  auto neg_of = [&](unsigned ax) { return L.neg_axis(ax); };
  // negative direction along a split axis => that node's second child first (bvh/mod.rs:411-417), at all three nodes involved
  const bool sp = neg_of(axis_p), sa = neg_of(axes & 3u) & (code[1] != 0xffffffffu), sb = neg_of((axes >> 2) & 3u) & (code[3] != 0xffffffffu);
  const int a0 = sa ? 1 : 0, a1 = sa ? 0 : 1, b0 = sb ? 3 : 2, b1 = sb ? 2 : 3;
  const int seq[4] = {sp ? b0 : a0, sp ? b1 : a1, sp ? a0 : b0, sp ? a1 : b1};
  bool h[4]; float t[4]; unsigned c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = seq[j];
    h[j] = k == 0 ? hit[0] : (k == 1 ? hit[1] : (k == 2 ? hit[2] : hit[3]));
    t[j] = k == 0 ? tmin[0] : (k == 1 ? tmin[1] : (k == 2 ? tmin[2] : tmin[3]));
    c[j] = k == 0 ? code[0] : (k == 1 ? code[1] : (k == 2 ? code[2] : code[3]));
  }
  // enter the first entry that is hit, leave the later ones pending (last one deepest in the stack)
  bool entered = false; unsigned next = 0;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    bool earlier = false;
#pragma unroll
    for (int e = 0; e < j; ++e) earlier |= h[e];
    if (h[j]) {
      if (earlier) { stack[L.sp * BLOCK] = c[j]; if (!ANY) { tstack[(size_t)L.sp * grid_lanes] = t[j]; L.ttop = t[j]; } ++L.sp; }
      else { entered = true; next = c[j]; }
    }
  }
  if (entered) L.cur = next; else pair_pop<ANY, BLOCK>(L, o, stack, tstack, grid_lanes);
}


template <int BLOCK, int DEPTH, int GENERAL = 0>  // occlusion rays only (closest hit four-wide was built in round 5, exact and slower: MEASUREMENTS R5)
__global__ void __launch_bounds__(BLOCK, RT_GEN_MIN_WAVES(GENERAL)) k_trace_quad(DScene sc, TraceIO io, const unsigned* __restrict__ queue, const unsigned* __restrict__ shard_counts, unsigned shard_cap,
                                                      unsigned count_static, unsigned long long* stats, int st_rays, float* __restrict__ tmin_stack_mem, unsigned refill_min) {
  constexpr bool ANY = true;
  const RT_SPTR_R(const float4) ray_o = io.ray_o; const RT_SPTR_R(const float4) ray_d = io.ray_d; const size_t rs = io.ray_stride;
  __shared__ unsigned stack_mem[DEPTH * BLOCK];
  unsigned* const stack = stack_mem + threadIdx.x;
  const size_t grid_lanes = (size_t)gridDim.x * BLOCK;
  float* const tstack = tmin_stack_mem + (size_t)blockIdx.x * BLOCK + threadIdx.x;
  QView qv; if (queue) qv.init(io.queue_is_slots ? nullptr : queue, shard_counts, shard_cap);
  const unsigned count = queue ? qv.total() : count_static;
  const unsigned lane = __lane_id();
  const unsigned n_waves = gridDim.x * (BLOCK / 64), wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const float4* __restrict__ quads = sc.quads; const float4* __restrict__ tri_p = sc.tri_p; const float4* __restrict__ nodes = sc.nodes;
  const TraceOut out = trace_out_of(io);
  const unsigned leaf_min = refill_min >> 8; refill_min &= 0xffu;  // the launch's two knobs travel in one word
  unsigned n_rays = 0;
  unsigned cursor = 0;
  bool exhausted = (unsigned long long)wave * 64ull >= count;
  PairLane L;
  L.active = false; L.found = false; L.pid = 0; L.dw = 0.0f;
  L.ray.o = L.ray.d = mk3(0, 0, 0); L.ray.t_max = 0.0f; L.set_inv(mk3(0, 0, 0)); L.set_rp(ray_pre(L.ray));
  L.sp = 0; L.prim = -1; L.cur = 0; L.ttop = 0.0f; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
  for (;;) {
    const unsigned long long idle = __ballot(!L.active);
    if (!exhausted && (unsigned)__popcll(idle) >= (idle == ~0ull ? 1u : refill_min)) {  // same refill scheme as k_trace_pair
      const unsigned v = cursor + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      const unsigned long long e = ((unsigned long long)(v >> 6) * n_waves + wave) * 64ull + (v & 63u);
      if (!L.active && e < count) {
        L.pid = queue ? qv.get((unsigned)e) : (unsigned)e;
        const float4 o4 = ray_o[L.pid * rs], d4 = ray_d[L.pid * rs];
        L.ray.o = mk3(o4.x, o4.y, o4.z); L.ray.d = mk3(d4.x, d4.y, d4.z); L.ray.t_max = o4.w; L.dw = d4.w;
        L.set_inv(mk3(1.0f / L.ray.d.x, 1.0f / L.ray.d.y, 1.0f / L.ray.d.z));
        L.set_rp(ray_pre(L.ray));
        L.sp = 0; L.prim = -1; L.found = false; L.hit.t = kInf; L.hit.b0 = L.hit.b1 = L.hit.b2 = 0.0f;
        L.active = true; n_rays += 1;
        const float4 r0 = nodes[0], r1 = nodes[1];  // the root is the one node tested on its own
        if (lane_slab_test(L.finite(), L.ray.o, L.ray.t_max, L.inv_dir, r0, r1)) {
          const unsigned packed = __float_as_uint(r1.w), np = packed & 0xffffu;
          L.cur = np > 0u ? (RT_PAIR_LEAF | (GENERAL ? RT_PAIR_GENERAL : 0u) | (unsigned)__float_as_int(r1.z) | ((np - 1u) << 26)) : (((packed >> 16) & 0xffu) << 29);
        } else pair_finish<ANY>(L, out);
      }
      cursor += (unsigned)__popcll(idle);
      exhausted = ((unsigned long long)(cursor >> 6) * n_waves + wave) * 64ull + (cursor & 63u) >= count;
    }
    if (__ballot(L.active) == 0ull) { if (exhausted) break; else continue; }
    {
      const bool at_leaf = L.active && (L.cur & RT_PAIR_LEAF) != 0u;
      const bool leaves_now = leaf_phase_now(L.active, at_leaf, leaf_min);
      if (L.active) {
        if (at_leaf) { if (leaves_now) pair_leaf_step<ANY, BLOCK, GENERAL>(L, out, sc, tri_p, stack, tstack, grid_lanes, ANY && io.shadow_masks != 0); }
        else quad_interior_step<ANY, BLOCK>(L, out, quads, stack, tstack, grid_lanes);
      }
    }
  }
  if (stats) {
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[st_rays], (unsigned long long)n_rays);
  }
}

// ================================================================================ K2b material binning
// The generic shade kernel evaluates tagged lobes ("for each lobe: switch (kind)"); a wave whose 64 vertices carry
// different materials executes the union of their code paths (measured on S4: 20 of 64 lanes active per VALU
// instruction). Between closest hit and shade the vertex queue is therefore counting-sorted by the code class of the hit
// triangle's material (materials of one kind whose parameters are textures of the same kinds run the same code and share a
// class; misses last), so that most waves run one code path. Order within a bin is arbitrary: paths are
// independent and the film sums each pixel's samples in sample order. hist/cursor: RT_BIN_MAX + 1 zeroed words each.
#define RT_BIN_MAX 256
#define RT_BIN_LDS_PRIMS 8192
// hit id -> index of the primitive in the scene's arrays (an id past the top level's primitives names (instance, the object's primitive))
RT_DEV int hit_primitive(const DInstance* __restrict__ instances, unsigned n_instances, unsigned n_top_prims, int hit_id) {
  if (n_instances == 0u || (unsigned)hit_id < n_top_prims) return hit_id;
  unsigned lo = 0, hi = n_instances;
  while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (instances[mid].id_base <= (unsigned)hit_id) lo = mid; else hi = mid; }
  return (int)(instances[lo].prim_base + ((unsigned)hit_id - instances[lo].id_base));
}
RT_DEV unsigned bin_of(const DScene& sc, const unsigned short* __restrict__ prim_class /* DScene::prim_class, or the workgroup's LDS copy */, const float4* __restrict__ hit, unsigned slot, unsigned n_bins) {
  int prim = __float_as_int(hit[slot].y);
  if (prim < 0) return n_bins - 1u;
  prim = hit_primitive(sc.instances, sc.n_instances, sc.n_top_prims, prim);
  const unsigned pc = prim_class[prim], m = pc & 0x7fffu;
  // route_quadric_hits: the bin before the miss bin collects the vertices on analytic quadrics, whatever their material - it lies in the generic range, so the
  // register-resident front-ends of such a scene (QLIGHTS forms) see triangles only and need no Sphere::intersect to rebuild an interaction
  if (sc.route_quadric_hits) {
    if (pc & 0x8000u) return n_bins - 2u;
    return m < n_bins - 2u ? m : n_bins - 3u;
  }
  return m < n_bins - 1u ? m : n_bins - 2u;
}
// bin_at[i]: the bin of queue entry i, kept for k_bin_scatter - finding it takes two dependent gathers (hit record -> triangle -> material) that need not be repeated
static __global__ void __launch_bounds__(256) k_bin_count(DScene sc, PassState ps, unsigned n_bins, unsigned* __restrict__ hist, unsigned short* __restrict__ bin_at) {
  __shared__ unsigned lh[RT_BIN_MAX + 1];
  __shared__ unsigned short s_pc[RT_BIN_LDS_PRIMS];  // the primitives' classes of a scene of few primitives (S3: 1294) in LDS
  for (unsigned i = threadIdx.x; i <= RT_BIN_MAX; i += 256u) lh[i] = 0u;
  const bool pc_lds = sc.n_tris <= RT_BIN_LDS_PRIMS;
  if (pc_lds) for (unsigned i = threadIdx.x; i < sc.n_tris; i += 256u) s_pc[i] = sc.prim_class[i];
  const unsigned short* const pcls = pc_lds ? (const unsigned short*)s_pc : sc.prim_class;
  __syncthreads();
  QView qv; if (ps.cnt_in) qv.init(ps.q_in, ps.cnt_in, ps.shard_cap);
  const unsigned count = ps.cnt_in ? qv.total() : ps.cap;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
    const unsigned slot = ps.cnt_in ? qv.get(i) : i;
    const unsigned b = bin_of(sc, pcls, sraw(ps.hit), slot, n_bins);
    bin_at[i] = (unsigned short)b;
    atomicAdd(&lh[b], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n_bins; i += 256u) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
// The bounce's paths grouped by bin; sorted_cnt[RT_QSHARDS]: {total, 0, ...} so that the result reads as a one-shard queue. `sorted` receives the slots and the shade
// kernels gather their records through it (a bin keeps the queue's order in rounds of 256 entries: runs of neighbouring slots).
static __global__ void __launch_bounds__(256) k_bin_scatter(DScene sc, PassState ps, unsigned n_bins, const unsigned* __restrict__ hist, unsigned* __restrict__ cursor,
                                                     unsigned* __restrict__ sorted, unsigned* __restrict__ sorted_cnt, unsigned split_bin, unsigned split_bin2,
                                                     unsigned split_bin3, unsigned* __restrict__ ranges, const unsigned short* __restrict__ bin_at) {
  __shared__ unsigned base[RT_BIN_MAX + 1], lcount[RT_BIN_MAX + 1], lbase[RT_BIN_MAX + 1];
  if (threadIdx.x == 0) {  // exclusive prefix of the histogram (<= 257 entries)
    unsigned run = 0;
    for (unsigned b = 0; b < n_bins; ++b) { base[b] = run; run += hist[b]; }
    if (blockIdx.x == 0) {
      sorted_cnt[0] = run; for (int k = 1; k < RT_QSHARDS; ++k) sorted_cnt[k * RT_CNT_STRIDE] = 0u;
      // bins [0, split_bin): classes of the Lambert front-end; [split_bin, split_bin2): of the two-lobe front-end; [split_bin2, split_bin3): of its wide form;
      // the rest: generic
      const unsigned split = split_bin < n_bins ? base[split_bin] : run, split2 = split_bin2 < n_bins ? base[split_bin2] : run;
      const unsigned split3 = split_bin3 < n_bins ? base[split_bin3] : run;
      const unsigned miss = base[n_bins - 1u];  // the last bin: rays that left the scene
      ranges[0] = 0u; ranges[1] = split; ranges[2] = split; ranges[3] = split2; ranges[4] = split3; ranges[5] = miss; ranges[6] = miss; ranges[7] = run;
      ranges[8] = split2; ranges[9] = split3;
    }
  }
  for (unsigned i = threadIdx.x; i <= RT_BIN_MAX; i += 256u) lcount[i] = 0u;
  __syncthreads();
  QView qv; if (ps.cnt_in) qv.init(ps.q_in, ps.cnt_in, ps.shard_cap);
  const unsigned count = ps.cnt_in ? qv.total() : ps.cap;
  constexpr unsigned E = 8;  // entries per lane per round: the barriers and the global atomics are per 2048 entries
  for (unsigned start = blockIdx.x * (256u * E); start < count; start += gridDim.x * (256u * E)) {
    unsigned pid[E], bin[E], rank[E]; bool live[E];
#pragma unroll
    for (unsigned k = 0; k < E; ++k) {  // rounds of 256 consecutive entries: a bin keeps the queue's order up to that granularity
      const unsigned i = start + k * 256u + threadIdx.x;
      live[k] = i < count; pid[k] = 0; bin[k] = 0; rank[k] = 0;
      if (live[k]) { pid[k] = ps.cnt_in ? qv.get(i) : i; bin[k] = bin_at[i]; }
    }
#pragma unroll
    for (unsigned k = 0; k < E; ++k) if (live[k]) rank[k] = atomicAdd(&lcount[bin[k]], 1u);
    __syncthreads();
    for (unsigned b = threadIdx.x; b < n_bins; b += 256u) if (lcount[b]) lbase[b] = atomicAdd(&cursor[b * RT_CNT_STRIDE], lcount[b]);  // one global atomic per bin per 1024 entries
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < E; ++k) if (live[k]) {
      const unsigned p = base[bin[k]] + lbase[bin[k]] + rank[k], from = pid[k];
      sorted[p] = from;
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < n_bins; b += 256u) lcount[b] = 0u;
    __syncthreads();
  }
}

// Can a ray from o along d (any t >= 0) pass through the box [lo, hi] widened by 1e-3 of its largest extent? Conservative on purpose (the widening is orders
// above what Sphere::intersect's error bounds can add to the surface; a zero direction component tests the origin against the slab): `false` is a proof.
RT_DEV bool ray_may_reach_box(f3 lo, f3 hi, f3 o, f3 d) {
  const float m = 1e-3f * fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z) + 1e-6f * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
  float t0 = 0.0f, t1 = kInf;
#define RT_SLAB(a) do { const float l_ = lo.a - m - o.a, h_ = hi.a + m - o.a; \
    if (d.a != 0.0f) { const float i_ = 1.0f / d.a; const float n_ = l_ * i_, f_ = h_ * i_; t0 = fmaxf(t0, fminf(n_, f_)); t1 = fminf(t1, fmaxf(n_, f_)); } \
    else if (l_ > 0.0f || h_ < 0.0f) return false; } while (0)
  RT_SLAB(x); RT_SLAB(y); RT_SLAB(z);
#undef RT_SLAB
  return t0 <= t1;
}

// what the shade kernels' LDS holds of a scene's small tables at most (k_shade's LDSREC forms; rt_scene_create decides per scene)
#define RT_LDS_LIGHTS 8
#define RT_LDS_MATERIALS 16
#define RT_LDS_TEXTURES 64
#define RT_LDS_IMAGES 8
// ================================================================================ K3 shade: rtx_shade_kernels.h (the k_shade front-ends and kernel, compiled by rtx_shade.hip); the miss bin's kernel:
// The miss bin of a binned queue: a path whose ray left the scene. PathIntegrator::li adds the environment's radiance only for camera rays
// and after specular bounces (path.rs:127-136, otherwise the light samples already account for it) and terminates the path; throughput,
// RNG and sampler counters stay as they are. One load for most entries - instead of the generic shade kernel's whole prologue.
static __global__ void __launch_bounds__(256) k_shade_miss(DScene sc, PassState ps) {
  const unsigned first = ps.range[0], count = ps.range[1];
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = first + blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const unsigned slot = ps.q_in ? ps.q_in[i] : i;  // (a binned queue: the sorted list of slots, or the moved records themselves)
    uint4 s4 = make_uint4(pack_state(0, false, 1, 2), slot, 0u, 0u);  // bounce 0 of a pass whose samples are all traced: PassState::fresh
    if (!RT_FRESH_ST(ps)) s4 = ps.in.st[slot];
    const unsigned st = s4.x, pid = s4.y;
    const int bounces = (int)(st & 0xffu); const bool specular_bounce = (st >> 8) & 1u;
    if (!(bounces == 0 || specular_bounce) || sc.n_infinite == 0) continue;
    float4 b4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (!RT_FRESH_BETA(ps)) b4 = ps.in.beta[slot];
    const float4 d4 = ps.in.d[slot], l4 = ps.lacc[pid];
    const f3 ray_d = mk3(d4.x, d4.y, d4.z);
    const rgb3 beta = mkc(b4.x, b4.y, b4.z);
    rgb3 L = mkc(l4.x, l4.y, l4.z);
    for (int k = 0; k < sc.n_infinite; ++k) L = L + beta * infinite_le(sc, sc.lights[k == 0 ? sc.infinite_ids[0] : (k == 1 ? sc.infinite_ids[1] : (k == 2 ? sc.infinite_ids[2] : sc.infinite_ids[3]))], ray_d);  // constant indices: the kernel argument stays in SGPRs
    ps.lacc[pid] = make_float4(L.r, L.g, L.b, l4.w);
  }
}

// ================================================================================ K5 resolve
// estimate_direct for the vertices whose BSDF-sampled MIS ray was traced: ld = [unoccluded] Ld1 +
// [the MIS ray reached the sampled light] f*Le*w/pdf; L += beta_at_vertex * (ld / light_pick_pdf).
// Everything about the vertex sits in its 128-byte MisRec, so the entries are gathered through the two MIS queues (closest-hit rays, then
// occlusion-only rays) at one line per vertex whatever order the shade kernel filled them in.
template <bool GENERAL>  // GENERAL: the emitter a MIS ray reached may be an analytic sphere
__global__ void __launch_bounds__(256) k_resolve(DScene sc, PassState ps) {
  QView qv; qv.init(ps.q_mis, ps.cnt_out + 2 * RT_QSHARDS * RT_CNT_STRIDE, ps.shard_cap);
  QView qa; qa.init(ps.q_misany, ps.cnt_out + 3 * RT_QSHARDS * RT_CNT_STRIDE, ps.shard_cap);
  const unsigned n_closest = qv.total(), count = n_closest + qa.total();
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const unsigned rec = i < n_closest ? qv.get(i) : qa.get(i - n_closest);  // the vertex's record index (its position in the shade launch's queue)
    const bool shadow_blocked = ps.occ_sh[rec] != 0;
    const bool mis_blocked = i >= n_closest && ps.occ_mi[rec] != 0;  // (an occlusion-only MIS ray)
    if (shadow_blocked && mis_blocked) continue;  // ld = 0: L + beta * (0 / pick_pdf) = L
    const unsigned pend = ps.mi.flags[rec];
    float4 a = ps.mi.a[rec], c = ps.mi.c[rec];
    rgb3 ld = mkc(0, 0, 0);
    if ((pend & RT_PEND_SHADOW) && !shadow_blocked) ld = ld + mkc(a.x, a.y, a.z);
    {
      float4 b = ps.mi.b[rec], d4 = ps.mi.d[rec]; const float4 h4 = (pend & RT_PEND_MIS_ANY) ? make_float4(0, 0, 0, 0) : ps.mi.hit[rec];
      const int light_num = (int)((pend >> 2) & 0x0fffffffu);
      const DLight& light = sc.lights[light_num];
      f3 wi = mk3(d4.x, d4.y, d4.z);
      rgb3 li = mkc(0, 0, 0);
      if (pend & RT_PEND_MIS_ANY) {  // infinite light, occlusion only (integrator/mod.rs:291-309: a hit surface is never this light)
        if (!mis_blocked) li = infinite_le(sc, light, wi);
      } else {
        const int prim = __float_as_int(h4.y);
        if (GENERAL && sc.n_instances != 0u && prim >= 0 && (unsigned)prim >= sc.n_top_prims) {  // a hit inside an object instance: an emitter there is in no light list (api.rs:954-964), so it is never the sampled light
        } else if (prim >= 0) {  // integrator/mod.rs:293-307: emitted radiance only if the hit emitter IS the sampled light
          if (GENERAL && tri_light(sc.tri_p, prim) == light_num && (tri_flags(sc.tri_p, prim) & RT_FLAG_SPHERE)) {
            float4 o4 = ps.mi.o[rec];
            SurfaceInteraction lsi; (void)sphere_fill_interaction(sc.spheres[prim_sphere_index(sc.tri_p, prim)], mk3(o4.x, o4.y, o4.z), wi, lsi);
            li = area_light_l(light, lsi.hit.n, -wi);
          } else if (tri_light(sc.tri_p, prim) == light_num) {
            float4 o4 = ps.mi.o[rec];
            f3 p0, p1, p2; load_tri(sc.tri_p, prim, p0, p1, p2);
            Ray r; r.o = mk3(o4.x, o4.y, o4.z); r.d = wi; r.t_max = kInf;
            TriHit th; (void)tri_test_call(p0, p1, p2, r, th);
            f3 p, n; tri_hit_point_normal(*sc.self, prim, th, p, n);
            li = area_light_l(light, n, -wi);
          }
        } else if (light.kind == 3) li = infinite_le(sc, light, wi);  // light.le(ray)
      }
      if (!is_black(li)) ld = ld + vdiv(mkc(b.x, b.y, b.z) * li * b.w, c.w);
    }
    rgb3 add = mkc(c.x, c.y, c.z) * vdiv(ld, a.w);
    if (add.r == 0.0f && add.g == 0.0f && add.b == 0.0f) continue;  // both rays blocked (most vertices of an interior): L + 0 = L, the scattered read-modify-write is skipped (a NaN is not 0)
    const float4 md4 = ps.mi.d[rec]; const unsigned pid = __float_as_uint(md4.w);  // the path the vertex belongs to
    float4 l4 = ps.lacc[pid];
    ps.lacc[pid] = make_float4(l4.x + add.r, l4.y + add.g, l4.z + add.b, l4.w);
  }
}

// ================================================================================ K6 film
// FilmTile::add_sample (film.rs:298-361) + merge (:177-194). One lane per batch pixel walks the pass's
// samples in order; the pixel's own running sum (registers within a pass, own_acc between passes) is
// therefore accumulated in exactly the reference's per-tile order and flushed once after the last pass;
// splats onto other pixels (filter radius > 0.5, or a sample exactly on a pixel edge) go through float
// atomics. film_acc: float4 (R, G, B sums, weight sum) per cropped pixel.
static __global__ void __launch_bounds__(256) k_film_accumulate(FrameParams fp, PassState ps, const float* __restrict__ filter_table, float4* film_acc) {
  const unsigned stride = gridDim.x * blockDim.x;
  const int cw = fp.crop_x1 - fp.crop_x0;
  const float inv_rx = 1.0f / fp.radius_x, inv_ry = 1.0f / fp.radius_y;
  const bool first_pass = ps.s0 == 0u, last_pass = ps.s0 + ps.n_samples >= ps.spp;
  for (unsigned pix = blockIdx.x * blockDim.x + threadIdx.x; pix < ps.n_pixels; pix += stride) {
    int x, y; unsigned long long pixel_index;
    owned_pixel(fp, fp.chunk_first + pix, x, y, pixel_index);
    if (y >= fp.sb_y1) continue;
    rgb3 own = mkc(0, 0, 0); float own_w = 0.0f;
    if (!first_pass) { float4 a = ps.own_acc[pix]; own = mkc(a.x, a.y, a.z); own_w = a.w; }
    unsigned scrubbed = 0;
    // one sample into the pixel's own sum and its neighbours' (FilmTile::add_sample), in sample order
    auto add_sample = [&](const float4 l4, const float2 pf) {
      if (__float_as_uint(l4.w) & RT_STATE_OUT_OF_BOUNDS) return;
      rgb3 c = mkc(l4.x, l4.y, l4.z);
      bool bad = false;  // renderer.rs:115-126
      if (has_nan(c)) { c = mkc(0, 0, 0); bad = true; }
      if (lum_y(c) < -1e-5f) { c = mkc(0, 0, 0); bad = true; }
      if (isinf(lum_y(c))) { c = mkc(0, 0, 0); bad = true; }
      scrubbed += bad;
      rgb3 Lc = lum_y(c) > fp.max_sample_luminance ? c * fp.max_sample_luminance / lum_y(c) : c;
      float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
      float p0x = ceilf(dx - fp.radius_x), p0y = ceilf(dy - fp.radius_y);
      float p1x = floorf(dx + fp.radius_x + 1.0f), p1y = floorf(dy + fp.radius_y + 1.0f);
      int x0 = f2i_sat(max_po(p0x, (float)fp.crop_x0)), y0 = f2i_sat(max_po(p0y, (float)fp.crop_y0));
      int x1 = f2i_sat(min_po(p1x, (float)fp.crop_x1)), y1 = f2i_sat(min_po(p1y, (float)fp.crop_y1));
      for (int yy = y0; yy < y1; ++yy) {
        float fy = fabsf(((float)yy - dy) * inv_ry * 16.0f);
        int iy = (int)f2u_sat(fminf(floorf(fy), 15.0f));
        for (int xx = x0; xx < x1; ++xx) {
          float fx = fabsf(((float)xx - dx) * inv_rx * 16.0f);
          int ix = (int)f2u_sat(fminf(floorf(fx), 15.0f));
          float fw = filter_table[iy * 16 + ix];
          if (xx == x && yy == y) { own = own + Lc * fw; own_w += fw; }
          else {
            float* dst = (float*)&film_acc[(size_t)(yy - fp.crop_y0) * cw + (xx - fp.crop_x0)];
            rgb3 v = Lc * fw;
            atomicAdd(dst + 0, v.r); atomicAdd(dst + 1, v.g); atomicAdd(dst + 2, v.b); atomicAdd(dst + 3, fw);
          }
        }
      }
        };
    // Round 6: the records of EIGHT samples are requested together, then added in order. A pixel's samples are one lane's sequential loop (the reference's order of a
    // pixel's sum); with one load per iteration a launch over few pixels - a shard's batch of 2^15 on an 8-GPU frame - waited a memory round trip per sample (1024 of them:
    // 3.8 ms of film time per S1 shard against 1.3 for its share of the whole frame's). Same loads, same adds, same order.
    constexpr unsigned KF = 8;
    for (unsigned s0 = 0; s0 < ps.n_samples; s0 += KF) {
      float4 lb[KF]; float2 pb[KF];
#pragma unroll
      for (unsigned k = 0; k < KF; ++k) {
        const unsigned sl = s0 + k < ps.n_samples ? s0 + k : ps.n_samples - 1u;
        const unsigned pid = sl * ps.n_pixels + pix;
        lb[k] = ps.lacc[pid]; pb[k] = ps.pfilm[pid];
      }
#pragma unroll
      for (unsigned k = 0; k < KF; ++k) if (s0 + k < ps.n_samples) add_sample(lb[k], pb[k]);
    }
    if (!last_pass) ps.own_acc[pix] = make_float4(own.r, own.g, own.b, own_w);
    else if (x >= fp.crop_x0 && x < fp.crop_x1 && y >= fp.crop_y0 && y < fp.crop_y1) {
      float* dst = (float*)&film_acc[(size_t)(y - fp.crop_y0) * cw + (x - fp.crop_x0)];
      atomicAdd(dst + 0, own.r); atomicAdd(dst + 1, own.g); atomicAdd(dst + 2, own.b); atomicAdd(dst + 3, own_w);
    }
    if (scrubbed) atomicAdd(&ps.stats[ST_SCRUBBED], (unsigned long long)scrubbed);
  }
}
// merge_film_tile's RGB -> XYZ (spectrum.rs:98-106); output (X, Y, Z, filter_weight_sum)
static __global__ void k_film_finalize(const float4* film_acc, float4* film_xyzw, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 a = film_acc[i];
  float X = 0.412453f * a.x + 0.357580f * a.y + 0.180423f * a.z;
  float Y = 0.212671f * a.x + 0.715160f * a.y + 0.072169f * a.z;
  float Z = 0.019334f * a.x + 0.119193f * a.y + 0.950227f * a.z;
  film_xyzw[i] = make_float4(X, Y, Z, a.w);
}

// ================================================================================ per-triangle / per-light constants (rt_scene_create)
// What Triangle::intersect computes of the triangle alone (dpdu, dpdv, the geometric normal and - without per-vertex normals or tangents - the whole shading
// frame and the first axis of Bsdf::new's frame) is evaluated once here, by the functions the per-vertex path uses (tri_geo, tri_frame: IEEE + - * / sqrt
// without contraction give the same bits wherever they run), and kept in one 128-byte record per triangle together with the traversal record.
static __global__ void __launch_bounds__(256) k_tri_records(DScene sc, float4* __restrict__ out) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= sc.n_tris) return;
  const float4 a = sc.tri_p[3 * (size_t)t], b = sc.tri_p[3 * (size_t)t + 1], c = sc.tri_p[3 * (size_t)t + 2];
  float4* r = out + 8 * (size_t)t;
  r[0] = a; r[1] = b; r[2] = c;
  const unsigned flags = __float_as_uint(c.w);
  const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (flags & (RT_FLAG_SPHERE | RT_FLAG_INSTANCE)) { r[3] = z; r[4] = z; r[5] = z; r[6] = z; r[7] = z; return; }
  const f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(c.x, c.y, c.z);
  f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
  if (flags & 4u) { const float* u = sc.tri_uv + 6 * (size_t)t; uv0 = mk2(u[0], u[1]); uv1 = mk2(u[2], u[3]); uv2 = mk2(u[4], u[5]); }
  const TriGeo g = tri_geo(p0, p1, p2, uv0, uv1, uv2);
  if (RT_REC_CONST_FRAME(flags)) {
    const TriFrame f = tri_frame(g, flags, g.n, g.n);
    const f3 ssb = normalize(f.ss);
    r[3] = make_float4(f.n.x, f.n.y, f.n.z, ssb.x); r[4] = make_float4(f.ss.x, f.ss.y, f.ss.z, ssb.y); r[5] = make_float4(f.ts.x, f.ts.y, f.ts.z, ssb.z);
  } else { r[3] = make_float4(g.n.x, g.n.y, g.n.z, 0.0f); r[4] = z; r[5] = z; }
  r[6] = make_float4(g.dpdu.x, g.dpdu.y, g.dpdu.z, 0.0f); r[7] = make_float4(g.dpdv.x, g.dpdv.y, g.dpdv.z, 0.0f);
}
// DScene::ld_rows8: the tables of a scene with <= 3 lights repacked, one 32-byte record per built voxel
static __global__ void __launch_bounds__(256) k_lightdist_rows8(const float* __restrict__ func, const float* __restrict__ cdf, const float* __restrict__ fint, int n_lights, unsigned rows, float4* __restrict__ out) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* f = func + (size_t)r * n_lights; const float* c = cdf + (size_t)r * (n_lights + 1);
  out[2 * (size_t)r] = make_float4(fint[r], f[0], n_lights > 1 ? f[1] : 0.0f, n_lights > 2 ? f[2] : 0.0f);
  out[2 * (size_t)r + 1] = make_float4(c[0], c[1], n_lights > 1 ? c[2] : 0.0f, n_lights > 2 ? c[3] : 0.0f);
}
static __global__ void __launch_bounds__(256) k_lightdist_dense8(const float4* __restrict__ rows8, const int* __restrict__ slot_of, unsigned long long n_voxels, float4* __restrict__ out) {
  const unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_voxels) return;
  const int slot = slot_of[v];
  out[2 * v] = slot >= 0 ? rows8[2 * (size_t)slot] : make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
  out[2 * v + 1] = slot >= 0 ? rows8[2 * (size_t)slot + 1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
static __global__ void __launch_bounds__(256) k_light_consts(DScene sc, DLight* __restrict__ lights, int n_all /* sampled lights + unlisted emitters */) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= n_all) return;
  DLight& l = lights[j];
  l.nrm[0] = l.nrm[1] = l.nrm[2] = 0.0f; l.inv_area = 0.0f;
  if (l.kind != 0) return;
  l.inv_area = 1.0f / l.area;
  if (tri_flags(sc.tri_p, l.prim) & RT_FLAG_SPHERE) return;
  f3 p0, p1, p2; load_tri(sc.tri_p, l.prim, p0, p1, p2);
  const f3 n = normalize(cross(p1 - p0, p2 - p0));
  l.nrm[0] = n.x; l.nrm[1] = n.y; l.nrm[2] = n.z;
}

// ================================================================================ light distribution build
// SpatialLightDistribution::compute_distribution (rc/lightdistrib.rs:101-179). Each light's 128-term sum is
// accumulated in the reference's order. The five Halton coordinates of the 128 points are voxel-independent and
// staged once per block in LDS.
// Which voxels can PathIntegrator::li ever look up? Only those holding a surface point (path.rs:154 passes isect.p),
// and isect.p = b0 p0 + b1 p1 + b2 p2 lies in its triangle's bounding box up to rounding. k_lightdist_mark marks,
// per triangle, the voxels of its (slightly inflated) box that its plane crosses - a conservative superset; the
// reference computes voxels lazily on first lookup (lightdistrib.rs:200-296), this is the eager equivalent.
// Only marked voxels are built (list = compacted marks); rt_light_distribution() asks for all of them.
static __global__ void __launch_bounds__(256) k_lightdist_mark(DScene sc, unsigned char* __restrict__ mark) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (sc.n_instances != 0u ? sc.n_top_prims : sc.n_tris)) return;  // object-space primitives follow the top level's: they are reached through their instances
  f3 p0, p1, p2; load_tri(sc.tri_p, (int)t, p0, p1, p2);
  // a quadric or an object instance: every voxel of its box is kept (the instance's slot holds its world box in p0 / p1: min and max below are that box)
  const bool is_instance = (tri_flags(sc.tri_p, (int)t) & RT_FLAG_INSTANCE) != 0u;
  const unsigned slot_index = __float_as_uint(p2.x);  // (a quadric's index in the quadric table / an instance's in the instance table)
  if (is_instance) p2 = p0;
  const bool is_sphere = (tri_flags(sc.tri_p, (int)t) & RT_FLAG_SPHERE) != 0u;
  f3 mn = mk3(fminf(p0.x, fminf(p1.x, p2.x)), fminf(p0.y, fminf(p1.y, p2.y)), fminf(p0.z, fminf(p1.z, p2.z)));
  f3 mx = mk3(fmaxf(p0.x, fmaxf(p1.x, p2.x)), fmaxf(p0.y, fmaxf(p1.y, p2.y)), fmaxf(p0.z, fmaxf(p1.z, p2.z)));
  // The box in a quadric's slot is the one the reference's BVH uses, and Disk::world_bounds is wrong under a rotation (disk.rs:127-134): surface points of a
  // disk can lie outside it. Marking needs a box that truly holds the surface: the 8 mapped corners of the object-space box.
  auto true_box = [&](const DSphere& q, const float* outer) {
    const float zlo = q.kind == 1 ? q.height : fminf(q.z_min, q.z_max), zhi = q.kind == 1 ? q.height : fmaxf(q.z_min, q.z_max);
    for (int c = 0; c < 8; ++c) {
      f3 w = xf34_point(q.o2w, mk3(c & 1 ? q.radius : -q.radius, c & 2 ? q.radius : -q.radius, c & 4 ? zhi : zlo));
      if (outer) w = xf34_point(outer, w);
      mn = mk3(fminf(mn.x, w.x), fminf(mn.y, w.y), fminf(mn.z, w.z)); mx = mk3(fmaxf(mx.x, w.x), fmaxf(mx.y, w.y), fmaxf(mx.z, w.z));
    }
  };
  if (is_sphere) { mn = mk3(kInf, kInf, kInf); mx = mk3(-kInf, -kInf, -kInf); true_box(sc.spheres[slot_index], nullptr); }
  // ... and an instance's world box is the image of its object's root box, which holds the quadrics' boxes of the same kind (found by scripts/fuzz_objects.py: a rotated
  // disk inside an object): widened by the true boxes of the object's quadrics under the instance's transform (objects that hold quadrics are few and small)
  if (is_instance && sc.obj_general) {
    const DInstance& in = sc.instances[slot_index];
    for (unsigned j = 0; j < in.n_prims; ++j) {
      const int g = (int)(in.prim_base + j);
      if (tri_flags(sc.tri_p, g) & RT_FLAG_SPHERE) true_box(sc.spheres[__float_as_uint(sc.tri_p[3 * (size_t)g + 2].x)], in.o2w);
    }
  }
  const f3 o0 = bounds_offset(sc.wb_min, sc.wb_max, mn), o1 = bounds_offset(sc.wb_min, sc.wb_max, mx);
  const float pad = 1e-4f;  // in units of the scene extent: >> the rounding of p and of voxel_of, << one voxel (1/64)
  int lo[3], hi[3];
  const float a0[3] = {o0.x, o0.y, o0.z}, a1[3] = {o1.x, o1.y, o1.z};
  for (int k = 0; k < 3; ++k) {
    lo[k] = clampi(f2i_sat(floorf((a0[k] - pad) * (float)sc.nvox[k])), 0, sc.nvox[k] - 1);
    hi[k] = clampi(f2i_sat(floorf((a1[k] + pad) * (float)sc.nvox[k])), 0, sc.nvox[k] - 1);
  }
  const f3 n = cross(p1 - p0, p2 - p0);
  const f3 ext = sc.wb_max - sc.wb_min;
  const f3 h = mk3(0.51f * ext.x / (float)sc.nvox[0] + pad * ext.x, 0.51f * ext.y / (float)sc.nvox[1] + pad * ext.y, 0.51f * ext.z / (float)sc.nvox[2] + pad * ext.z);  // inflated half extents
  const float r = fabsf(n.x) * h.x + fabsf(n.y) * h.y + fabsf(n.z) * h.z;
  const bool flat = is_sphere || !(len2(n) > 0.0f);  // degenerate triangle: keep its whole box
  for (int z = lo[2]; z <= hi[2]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y)
      for (int x = lo[0]; x <= hi[0]; ++x) {
        const f3 c = mk3(sc.wb_min.x + ((float)x + 0.5f) * ext.x / (float)sc.nvox[0], sc.wb_min.y + ((float)y + 0.5f) * ext.y / (float)sc.nvox[1],
                         sc.wb_min.z + ((float)z + 0.5f) * ext.z / (float)sc.nvox[2]);
        if (flat || fabsf(dot(n, c - p0)) <= r * 1.01f) mark[((size_t)z * sc.nvox[1] + y) * sc.nvox[0] + x] = 1;
      }
}
static __global__ void __launch_bounds__(256) k_lightdist_compact(const unsigned char* __restrict__ mark, unsigned n_vox, int mark_all, unsigned* __restrict__ list, unsigned* __restrict__ n_list) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool take = v < n_vox && (mark_all || mark[v]);
  const unsigned slot = wave_push(n_list, take);
  if (take) list[slot] = v;
}

// Two kernels so that a scene with many lights fills the chip: k_lightdist_contrib has one lane per
// (built voxel, light) and accumulates that light's 128-term sum in the reference's order; k_lightdist_finish has
// one lane per built voxel and runs the voxel's sequential tail (sum over lights, floor, Distribution1D::new).
RT_DEV void voxel_bounds(const DScene& sc, long v, f3& vmn, f3& vmx) {
  const int px = (int)(v % sc.nvox[0]), py = (int)((v / sc.nvox[0]) % sc.nvox[1]), pz = (int)(v / ((long)sc.nvox[0] * sc.nvox[1]));
  f3 p0 = mk3((float)px / (float)sc.nvox[0], (float)py / (float)sc.nvox[1], (float)pz / (float)sc.nvox[2]);
  f3 p1 = mk3(((float)px + 1.0f) / (float)sc.nvox[0], ((float)py + 1.0f) / (float)sc.nvox[1], ((float)pz + 1.0f) / (float)sc.nvox[2]);
  f3 a = bounds_lerp(sc.wb_min, sc.wb_max, p0), b = bounds_lerp(sc.wb_min, sc.wb_max, p1);
  vmn = mk3(min_po(a.x, b.x), min_po(a.y, b.y), min_po(a.z, b.z)); vmx = mk3(max_po(a.x, b.x), max_po(a.y, b.y), max_po(a.z, b.z));
}
// Lanes of a 128-lane block = (voxel slot, light): with lights_pad = n_lights rounded up to a power of two, a block
// covers 128 / lights_pad list entries when lights_pad <= 128, else bpv = ceil(n_lights / 128) consecutive blocks cover one.
template <bool GENERAL>  // GENERAL: an emitter may be an analytic sphere
__global__ void __launch_bounds__(128) k_lightdist_contrib(DScene sc, const unsigned* __restrict__ list, const unsigned* __restrict__ n_list, unsigned lights_pad, unsigned bpv, float* func) {
  __shared__ float halton[128 * 5];
  unsigned li; int j;
  if (lights_pad <= 128u) { const unsigned vpb = 128u / lights_pad; li = blockIdx.x * vpb + threadIdx.x / lights_pad; j = (int)(threadIdx.x % lights_pad); }
  else { li = blockIdx.x / bpv; j = (int)((blockIdx.x % bpv) * 128u + threadIdx.x); }  // bpv = ceil(n_lights / 128)
  if ((lights_pad <= 128u ? blockIdx.x * (128u / lights_pad) : blockIdx.x / bpv) >= *n_list) return;  // whole block idle
  for (unsigned i = threadIdx.x; i < 128u * 5u; i += blockDim.x) halton[i] = radical_inverse((int)(i % 5u), (unsigned long long)(i / 5u));
  __syncthreads();
  if (li >= *n_list || j >= sc.n_lights) return;
  const long v = (long)list[li];
  f3 vmn, vmx; voxel_bounds(sc, v, vmn, vmx);
  const DLight& light = sc.lights[j];
  float contrib = 0.0f;
  if (light.kind == 0 && !(GENERAL && (tri_flags(sc.tri_p, light.prim) & RT_FLAG_SPHERE))) {
    // DiffuseAreaLight::sample_li (area_light_sample_li) with what does not depend on the sample taken out of the 128-sample loop:
    // the emitter's vertices, its normal when the mesh carries no per-vertex normals, 1 / area. Same operations per sample, same sums.
    f3 p0, p1, p2; load_tri(sc.tri_p, light.prim, p0, p1, p2);
    const unsigned flags = tri_flags(sc.tri_p, light.prim);
    const f3 geo_n = normalize(cross(p1 - p0, p2 - p0));
    const float inv_area = 1.0f / light.area;
    f3 n0 = mk3(0, 0, 0), n1 = n0, n2 = n0;
    if (flags & 2u) { const float* q = sc.tri_n + 9 * (size_t)light.prim; n0 = mk3(q[0], q[1], q[2]); n1 = mk3(q[3], q[4], q[5]); n2 = mk3(q[6], q[7], q[8]); }
    for (int i = 0; i < 128; ++i) {
      const f3 ref_p = bounds_lerp(vmn, vmx, mk3(halton[5 * i], halton[5 * i + 1], halton[5 * i + 2]));
      const f2 bq = uniform_sample_triangle(mk2(halton[5 * i + 3], halton[5 * i + 4]));
      const float b2 = 1.0f - bq.x - bq.y;
      const f3 p = (bq.x * p0) + (bq.y * p1) + (b2 * p2);
      f3 normal = geo_n;
      if (flags & 2u) normal = face_forward(normal, bq.x * n0 + bq.y * n1 + b2 * n2);
      else if (flags & 1u) normal = normal * -1.0f;
      float pdf = inv_area;
      f3 wi = p - ref_p;
      if (len2(wi) == 0.0f) pdf = 0.0f;
      else {
        wi = normalize(wi);
        pdf *= distance_squared(ref_p, p) / fabsf(dot(normal, -wi));
        if (isinf(pdf)) pdf = 0.0f;
      }
      const rgb3 li = area_light_l(light, normal, -normalize(p - ref_p));
      if (pdf > 0.0f) contrib += lum_y(li) / pdf;
    }
  } else {
    for (int i = 0; i < 128; ++i) {
      Interaction intr;
      intr.p = bounds_lerp(vmn, vmx, mk3(halton[5 * i], halton[5 * i + 1], halton[5 * i + 2]));
      intr.p_error = mk3(0, 0, 0); intr.wo = mk3(1, 0, 0); intr.n = mk3(0, 0, 0);
      LiSample s = light_sample_li_full<GENERAL, true>(*sc.self, light, intr, mk2(halton[5 * i + 3], halton[5 * i + 4]));  // (exact quotients: these tables are bit-equal to the oracle's)
      if (s.pdf > 0.0f) contrib += lum_y(s.li) / s.pdf;
    }
  }
  func[(size_t)li * sc.n_lights + j] = contrib;
}
static __global__ void __launch_bounds__(128) k_lightdist_finish(DScene sc, const unsigned* __restrict__ list, const unsigned* __restrict__ n_list, float* func, float* cdf, float* fint, int* slot_of,
                                                          unsigned short* guide, int glog) {
  const unsigned li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= *n_list) return;
  const long v = (long)list[li];
  slot_of[v] = (int)li;
  const int nl = sc.n_lights;
  float* fv = func + (size_t)li * nl; float* cv = cdf + (size_t)li * (nl + 1);
  float sum = 0.0f;
  for (int j = 0; j < nl; ++j) sum += fv[j];
  float avg = sum / (float)(128ull * (unsigned long long)nl);
  float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
  // Distribution1D::new (distribution1d.rs:11-42)
  cv[0] = 0.0f;
  for (int j = 0; j < nl; ++j) { float c = fmaxf(fv[j], min_contrib); fv[j] = c; cv[j + 1] = cv[j] + c / (float)nl; }
  float func_int = cv[nl];
  if (func_int == 0.0f) for (int j = 1; j < nl + 1; ++j) cv[j] = (float)j / (float)nl;
  else for (int j = 1; j < nl + 1; ++j) cv[j] /= func_int;
  fint[li] = func_int;
  if (glog >= 0) {  // guide[k] = the number of CDF entries <= k / 2^glog (DScene::ld_guide)
    const int G = 1 << glog; unsigned short* gv = guide + (size_t)li * (G + 1); int i = 0;
    for (int k = 0; k <= G; ++k) { const float x = (float)k / (float)G; while (i <= nl && cv[i] <= x) ++i; gv[k] = (unsigned short)i; }
  }
}
static __global__ void k_lightdist_iota(unsigned n, unsigned* __restrict__ list, unsigned* __restrict__ n_list) {  // every voxel, in order: slot == voxel
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) list[v] = v;
  if (v == 0) *n_list = n;
}

}  // namespace rtx
