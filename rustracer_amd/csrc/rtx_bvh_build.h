// rtx_bvh_build.h — BVH construction on the device. Included by rtx_hip.hip.
//
// The reference builds its tree on one host thread (BVH::new, rc/bvh/mod.rs:80-135: recursive 12-bucket SAH); the host layer
// restates that build (rtx_host.cpp) and it stays the default because its trees trace fastest. This is the other end of the
// trade: a linear BVH in the manner of Karras 2012 ("Maximizing parallelism in the construction of BVHs, octrees and k-d
// trees") - 63-bit Morton keys of the triangle centroids, one radix sort, every internal node found independently from
// the sorted keys, boxes fitted bottom-up - a few milliseconds for a million triangles instead of seconds, for trees
// that cost more node visits per ray. Output is the same rt_bvh_node array (pre-order, first child = i + 1) and primitive
// order the SAH build gives, so nothing downstream knows which builder ran.
#pragma once
#include <hipcub/hipcub.hpp>

namespace rtx {

struct LbvhNode {       // Karras node i of n-1 internal nodes; children >= n_internal are leaves (sorted triangle c - n_internal)
  int left, right, parent;
  int first, last;      // range of sorted triangles below
};

RT_DEV unsigned long long morton_spread21(unsigned v) {  // 21 bits -> every third bit
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// float min / max through ordered integers (valid for all finite values, either sign)
RT_DEV void atomic_min_f(float* a, float v) { v += 0.0f; if (v >= 0.0f) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned*)a, __float_as_uint(v)); }
RT_DEV void atomic_max_f(float* a, float v) { v += 0.0f; if (v >= 0.0f) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned*)a, __float_as_uint(v)); }

__global__ void k_lbvh_centroid_bounds(const float* __restrict__ tri_p, unsigned n, float* __restrict__ cb /* min xyz, max xyz */) {
  __shared__ float lo[3][4], hi[3][4];
  float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = tri_p + 9ull * i;
    for (int a = 0; a < 3; ++a) {
      const float tmn = fminf(p[a], fminf(p[3 + a], p[6 + a])), tmx = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
      const float c = 0.5f * tmn + 0.5f * tmx;  // the centroid the reference bins by: 0.5 * pmin + 0.5 * pmax (bounds.rs:189-191)
      mn[a] = fminf(mn[a], c); mx[a] = fmaxf(mx[a], c);
    }
  }
  for (int a = 0; a < 3; ++a)
    for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_down(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off)); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { lo[a][wave] = mn[a]; hi[a][wave] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x; float l = lo[a][0], h = hi[a][0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { l = fminf(l, lo[a][w]); h = fmaxf(h, hi[a][w]); }
    atomic_min_f(cb + a, l); atomic_max_f(cb + 3 + a, h);
  }
}

__global__ void k_lbvh_keys(const float* __restrict__ tri_p, unsigned n, const float* __restrict__ cb, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = tri_p + 9ull * i;
  unsigned q[3];
  for (int a = 0; a < 3; ++a) {
    const float tmn = fminf(p[a], fminf(p[3 + a], p[6 + a])), tmx = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
    const float c = 0.5f * tmn + 0.5f * tmx, ext = cb[3 + a] - cb[a];
    float u = ext > 0.0f ? (c - cb[a]) / ext : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    q[a] = min((unsigned)(u * 2097152.0f), 2097151u);
  }
  keys[i] = morton_spread21(q[0]) << 2 | morton_spread21(q[1]) << 1 | morton_spread21(q[2]);
  vals[i] = i;
}

// length of the common prefix of sorted keys i and j, -1 outside the array; equal keys are told apart by their positions
RT_DEV int lbvh_delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = k[i], b = k[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

__global__ void k_lbvh_hierarchy(const unsigned long long* __restrict__ keys, int n, LbvhNode* __restrict__ nodes, int* __restrict__ leaf_parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;  // direction of the range
  const int dmin = lbvh_delta(keys, n, i, i - d);
  int lmax = 2;
  while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lbvh_delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {  // split: highest position sharing more than dnode bits with i
    if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const int left = first == gamma ? (n - 1) + gamma : gamma, right = last == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
  nodes[i].left = left; nodes[i].right = right; nodes[i].first = first; nodes[i].last = last;
  if (left >= n - 1) leaf_parent[left - (n - 1)] = i; else nodes[left].parent = i;
  if (right >= n - 1) leaf_parent[right - (n - 1)] = i; else nodes[right].parent = i;
  if (i == 0) nodes[0].parent = -1;
}

struct LbvhBox { float lo[3], hi[3]; };

// Bottom-up fit without in-kernel hand-offs: one launch per tree level. A node is put on the next launch's work list by whichever of
// its children finishes second (one atomic per node), so everything a launch reads was written by an earlier launch and no
// device-scope fence is needed (an in-kernel walk with a fence per step measured 7.1 ms for 10^6 triangles; this takes well under 1).
RT_DEV void lbvh_child_done(int parent, unsigned* __restrict__ visits, int* __restrict__ list_out, unsigned* __restrict__ n_out) {
  if (parent >= 0 && atomicAdd(&visits[parent], 1u) == 1u) list_out[atomicAdd(n_out, 1u)] = parent;
}

__global__ void k_lbvh_fit_leaves(const float* __restrict__ tri_p, const unsigned* __restrict__ order, int n, const int* __restrict__ leaf_parent,
                                  LbvhBox* __restrict__ boxes /* 2n-1: internal then leaves */, float* __restrict__ costs /* 2n-1 */, unsigned* __restrict__ visits,
                                  int* __restrict__ list_out, unsigned* __restrict__ counters /* [3] */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float* p = tri_p + 9ull * order[t];
  LbvhBox b;
  for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(p[a], fminf(p[3 + a], p[6 + a])); b.hi[a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a])); }
  boxes[(n - 1) + t] = b; costs[(n - 1) + t] = 1.0f;
  lbvh_child_done(leaf_parent[t], visits, list_out, &counters[0]);
}

// pass k reads counters[k % 3] entries of list_in, appends to list_out / counters[(k + 1) % 3] and clears counters[(k + 2) % 3]
__global__ void k_lbvh_fit_pass(int n, int max_prims, int pass, const LbvhNode* __restrict__ nodes, LbvhBox* __restrict__ boxes, int* __restrict__ sizes /* n-1 */,
                                unsigned char* __restrict__ axis /* n-1: bits 0-1 axis, 4 = children swapped, 8 = collapsed into a leaf */, float* __restrict__ costs,
                                unsigned* __restrict__ visits, const int* __restrict__ list_in, int* __restrict__ list_out, unsigned* __restrict__ counters) {
  const unsigned count = counters[pass % 3];
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[(pass + 2) % 3] = 0u;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const int node = list_in[i];
    const LbvhNode nd = nodes[node];
    const LbvhBox l = boxes[nd.left], r = boxes[nd.right];
    LbvhBox u;
    for (int a = 0; a < 3; ++a) { u.lo[a] = fminf(l.lo[a], r.lo[a]); u.hi[a] = fmaxf(l.hi[a], r.hi[a]); }
    boxes[node] = u;
    // traversal order: the axis along which the children's box centres are furthest apart; `swap` when the first child is the far one
    float best = -1.0f; int ax = 0; bool swap = false;
    for (int a = 0; a < 3; ++a) { const float dl = (l.lo[a] + l.hi[a]) - (r.lo[a] + r.hi[a]); if (fabsf(dl) > best) { best = fabsf(dl); ax = a; swap = dl > 0.0f; } }
    // leaf or split by the reference's cost model (bvh/mod.rs:233-265): n against 1 + (A_l c_l + A_r c_r) / A, decided bottom-up
    const int cnt = nd.last - nd.first + 1;
    auto area = [](const LbvhBox& x) { const float dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2]; return 2.0f * (dx * dy + dx * dz + dy * dz); };
    const float au = area(u);
    const float split_cost = au > 0.0f ? 1.0f + (area(l) * costs[nd.left] + area(r) * costs[nd.right]) / au : 3.402823466e38f;
    const bool collapse = cnt <= max_prims && (float)cnt <= split_cost;
    costs[node] = collapse ? (float)cnt : split_cost;
    axis[node] = (unsigned char)(ax | (swap ? 4 : 0) | (collapse ? 8 : 0));
    const int sl = nd.left >= n - 1 ? 1 : sizes[nd.left], sr = nd.right >= n - 1 ? 1 : sizes[nd.right];
    sizes[node] = collapse ? 1 : 1 + sl + sr;
    lbvh_child_done(nd.parent, visits, list_out, &counters[(pass + 1) % 3]);
  }
}

// every surviving node finds its pre-order slot by walking to the root: +1 per step down a first child, +1 + size(first) per step down a second child
__global__ void k_lbvh_emit(int n, int max_prims, const LbvhNode* __restrict__ nodes, const int* __restrict__ leaf_parent, const LbvhBox* __restrict__ boxes,
                            const int* __restrict__ sizes, const unsigned char* __restrict__ axis, rt_bvh_node* __restrict__ out, int* __restrict__ max_depth) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n - 1) return;
  const bool is_leaf = t >= n - 1;
  const int parent = is_leaf ? leaf_parent[t - (n - 1)] : nodes[t].parent;
  const int count = is_leaf ? 1 : nodes[t].last - nodes[t].first + 1;
  const bool leaf_out = is_leaf || (axis[t] & 8);
  int slot = 0, depth = 0, child = t, p = parent;
  while (p >= 0) {
    const LbvhNode nd = nodes[p];
    if (axis[p] & 8) return;  // inside a collapsed range
    const bool swap = axis[p] & 4;
    const int first = swap ? nd.right : nd.left;
    if (child == first) slot += 1;
    else slot += 1 + (first >= n - 1 ? 1 : sizes[first]);
    child = p; p = nd.parent; depth += 1;
  }
  rt_bvh_node o;
  const LbvhBox b = boxes[t];
  for (int a = 0; a < 3; ++a) { o.bmin[a] = b.lo[a]; o.bmax[a] = b.hi[a]; }
  o.pad = 0;
  if (leaf_out) { o.offset = (unsigned)(is_leaf ? t - (n - 1) : nodes[t].first); o.n_prims = (unsigned short)count; o.axis = 0; }
  else {
    const bool swap = axis[t] & 4;
    const int first = swap ? nodes[t].right : nodes[t].left;
    o.offset = (unsigned)(slot + 1 + (first >= n - 1 ? 1 : sizes[first])); o.n_prims = 0; o.axis = (unsigned char)(axis[t] & 3);
  }
  out[slot] = o;
  atomicMax(max_depth, depth);
}

}  // namespace rtx

// BVH::new for callers that want the tree now rather than the best tree (rc/bvh/mod.rs:80-135). tri_p: n_tris x 9 floats (world
// space, host memory); nodes: room for 2 * n_tris - 1 records; ordered: n_tris source indices in leaf order.
extern "C" int rt_bvh_build(const float* tri_p, uint32_t n_tris, int32_t max_prims_per_node, rt_bvh_node* nodes, uint32_t* n_nodes, int32_t* ordered, float* ms_device) {
  using namespace rtx;
  if (!tri_p || !nodes || !n_nodes || !ordered || n_tris == 0) return fail(RT_ERR_INVALID, "null argument / empty mesh");
  if (n_tris > (1u << 30)) return fail(RT_ERR_INVALID, "too many triangles");
  if (!rt_device_available()) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; this backend has no CPU fallback");
  const int n = (int)n_tris;
  const int max_prims = max_prims_per_node < 1 ? 1 : (max_prims_per_node > 255 ? 255 : max_prims_per_node);  // bvh/mod.rs:119
  if (n == 1) {
    rt_bvh_node o; memset(&o, 0, sizeof o);
    for (int a = 0; a < 3; ++a) { o.bmin[a] = fminf(tri_p[a], fminf(tri_p[3 + a], tri_p[6 + a])); o.bmax[a] = fmaxf(tri_p[a], fmaxf(tri_p[3 + a], tri_p[6 + a])); }
    o.n_prims = 1; nodes[0] = o; *n_nodes = 1; ordered[0] = 0; if (ms_device) *ms_device = 0.0f;
    return RT_OK;
  }
  DevBuf d_p, d_cb, d_keys, d_vals, d_keys2, d_vals2, d_tmp, d_nodes, d_lpar, d_boxes, d_sizes, d_axis, d_costs, d_visits, d_out, d_depth, d_list[2], d_counters;
  HIP_TRY(d_p.ensure((size_t)n * 36)); HIP_TRY(hipMemcpy(d_p.p, tri_p, (size_t)n * 36, hipMemcpyHostToDevice));
  HIP_TRY(d_cb.ensure(24)); HIP_TRY(d_keys.ensure((size_t)n * 8)); HIP_TRY(d_vals.ensure((size_t)n * 4)); HIP_TRY(d_keys2.ensure((size_t)n * 8)); HIP_TRY(d_vals2.ensure((size_t)n * 4));
  HIP_TRY(d_nodes.ensure((size_t)(n - 1) * sizeof(LbvhNode))); HIP_TRY(d_lpar.ensure((size_t)n * 4)); HIP_TRY(d_boxes.ensure((size_t)(2 * n - 1) * sizeof(LbvhBox)));
  HIP_TRY(d_sizes.ensure((size_t)(n - 1) * 4)); HIP_TRY(d_axis.ensure((size_t)(n - 1))); HIP_TRY(d_visits.ensure((size_t)(n - 1) * 4)); HIP_TRY(d_costs.ensure((size_t)(2 * n - 1) * 4));
  HIP_TRY(d_list[0].ensure((size_t)n * 4)); HIP_TRY(d_list[1].ensure((size_t)n * 4)); HIP_TRY(d_counters.ensure(12));
  HIP_TRY(d_out.ensure((size_t)(2 * n - 1) * sizeof(rt_bvh_node))); HIP_TRY(d_depth.ensure(4));
  hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, 0));
  const float init[6] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
  HIP_TRY(hipMemcpyAsync(d_cb.p, init, 24, hipMemcpyHostToDevice, 0));
  HIP_TRY(hipMemsetAsync(d_visits.p, 0, (size_t)(n - 1) * 4, 0)); HIP_TRY(hipMemsetAsync(d_depth.p, 0, 4, 0)); HIP_TRY(hipMemsetAsync(d_counters.p, 0, 12, 0));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  k_lbvh_centroid_bounds<<<blocks < 2048u ? blocks : 2048u, 256, 0, 0>>>(d_p.as<float>(), (unsigned)n, d_cb.as<float>());
  k_lbvh_keys<<<blocks, 256, 0, 0>>>(d_p.as<float>(), (unsigned)n, d_cb.as<float>(), d_keys.as<unsigned long long>(), d_vals.as<unsigned>());
  size_t tmp_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), d_vals.as<unsigned>(), d_vals2.as<unsigned>(), n, 0, 63, 0));
  HIP_TRY(d_tmp.ensure(tmp_bytes));
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), d_vals.as<unsigned>(), d_vals2.as<unsigned>(), n, 0, 63, 0));
  k_lbvh_hierarchy<<<blocks, 256, 0, 0>>>(d_keys2.as<unsigned long long>(), n, d_nodes.as<LbvhNode>(), d_lpar.as<int>());
  k_lbvh_fit_leaves<<<blocks, 256, 0, 0>>>(d_p.as<float>(), d_vals2.as<unsigned>(), n, d_lpar.as<int>(), d_boxes.as<LbvhBox>(), d_costs.as<float>(), d_visits.as<unsigned>(),
                                           d_list[0].as<int>(), d_counters.as<unsigned>());
  {  // one launch per level until a launch finds nothing to do; the tree is at most 64 + log2(n) levels high (63 key bits, then positions)
    const unsigned pass_blocks = blocks < 1024u ? blocks : 1024u;
    int pass = 0;
    for (bool more = true; more && pass < 192;) {
      for (int k = 0; k < 16; ++k, ++pass)
        k_lbvh_fit_pass<<<pass_blocks, 256, 0, 0>>>(n, max_prims, pass, d_nodes.as<LbvhNode>(), d_boxes.as<LbvhBox>(), d_sizes.as<int>(), d_axis.as<unsigned char>(), d_costs.as<float>(),
                                                    d_visits.as<unsigned>(), d_list[pass & 1].as<int>(), d_list[(pass + 1) & 1].as<int>(), d_counters.as<unsigned>());
      unsigned pending = 0;
      HIP_TRY(hipMemcpyAsync(&pending, d_counters.as<unsigned>() + pass % 3, 4, hipMemcpyDeviceToHost, 0));
      HIP_TRY(hipStreamSynchronize(0));
      more = pending != 0;
    }
    if (pass >= 192) return fail(RT_ERR_UNSUPPORTED, "device-built BVH is too deep: use the SAH build");
  }
  k_lbvh_emit<<<(unsigned)((2 * n - 1 + 255) / 256), 256, 0, 0>>>(n, max_prims, d_nodes.as<LbvhNode>(), d_lpar.as<int>(), d_boxes.as<LbvhBox>(), d_sizes.as<int>(), d_axis.as<unsigned char>(),
                                                                   d_out.as<rt_bvh_node>(), d_depth.as<int>());
  HIP_TRY(hipEventRecord(e1, 0));
  HIP_TRY(hipGetLastError());
  int total = 0, depth = 0;
  HIP_TRY(hipMemcpy(&total, d_sizes.p, 4, hipMemcpyDeviceToHost));  // size of the root's subtree
  HIP_TRY(hipMemcpy(&depth, d_depth.p, 4, hipMemcpyDeviceToHost));
  float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (total < 1 || total > 2 * n - 1) return fail(RT_ERR_HIP, "device BVH build produced an inconsistent tree");
  if (depth + 1 > 64) return fail(RT_ERR_UNSUPPORTED, "device-built BVH is deeper than the 64-entry traversal stack (many coincident centroids): use the SAH build");
  HIP_TRY(hipMemcpy(nodes, d_out.p, (size_t)total * sizeof(rt_bvh_node), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ordered, d_vals2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  *n_nodes = (uint32_t)total;
  if (ms_device) *ms_device = ms;
  return RT_OK;
}
