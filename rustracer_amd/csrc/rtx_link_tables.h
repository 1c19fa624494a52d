// Host side of the LDS walks' tables (rt_scene_create; included by rtx_hip.hip): the skip table of round 4's walks, and round 5's link tables - per direction octant
// and node where the stackless walk goes when the node's box passes and where it carries on after the node's subtree - over all nodes (link_full) and over the nodes
// the calibration below found worth testing (link_kept). Nothing here touches the device: rt_link_tables() exposes it to tests that run without one.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <vector>
#include "../../include/rtx_hip.h"
// (after rtx_kernels.h: RT_LINK_OFF_BITS, RT_MID_NODES, RT_SMALL_NODES)

struct RtLinkTables {
  std::vector<uint16_t> skip8;                 // 8 x n_nodes
  std::vector<uint32_t> link_full, link_kept;  // 9 x n_nodes + 9 each: rows 0 - 7 (closest hit, by octant), their 8 start nodes, row 8 (occlusion rays), its start node
  std::vector<std::vector<char>> kept;         // [9][n_nodes]: the nodes each walk tests
  std::vector<char> any_swap;                  // the occlusion walk's order: second child first at these nodes
  uint32_t nodes_tested = 0;                   // the closest-hit walks' average over the octants
  size_t n_rays[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // calibration rays per set, and their simulated node tests with every node / with the kept ones (want_stats)
  double tests_all[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tests_kept[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// mid: a mid-size scene (k_trace<.., MID>: 11 bits for a leaf's first primitive in its link word); want_stats: fill n_rays / tests_all / tests_kept
static void rt_build_link_tables(const rt_scene_desc* desc, bool mid, bool want_stats, RtLinkTables& out) {
    // The order in which BVH::intersect (bvh/mod.rs:381-425) reaches the nodes depends on the ray only through the signs of its direction (at an interior node
    // the child on the ray's side of the split axis first, the other one pushed): one fixed order per octant. skip[o][i] = the entry on top of the to-visit stack
    // when that walk reaches node i = where it carries on once i's subtree is done; n_nodes when nothing is pending.
    const uint32_t nn = desc->n_nodes;
    std::vector<uint16_t>& skip = out.skip8; skip.assign((size_t)8 * nn, (uint16_t)nn);
    std::vector<uint32_t> st; st.reserve(64);
    for (uint32_t o = 0; o < 8; ++o) {
      st.clear();
      uint32_t cur = 0;
      for (;;) {
        skip[(size_t)o * nn + cur] = (uint16_t)(st.empty() ? nn : st.back());
        const rt_bvh_node& n = desc->nodes[cur];
        if (n.n_prims == 0) {
          const bool neg = ((o >> (n.axis < 2 ? n.axis : 2)) & 1u) != 0u;  // (the walks read an axis other than 0 / 1 as z)
          if (neg) { st.push_back(cur + 1); cur = n.offset; } else { st.push_back(n.offset); cur = cur + 1; }
        } else {
          if (st.empty()) break;
          cur = st.back(); st.pop_back();
        }
      }
    }
    // ---- round 5: link tables of the stackless walks (rtx_kernels.h, closest_small_links), over all nodes and over the nodes worth testing.
    // A box contains its children's boxes, so an interior node's test decides nothing - it saves its subtree's tests when it fails and costs one test when it does not. Which
    // interior nodes earn their test is measured here on synthetic rays of the kind a path tracer casts (origins on the surfaces, cosine-distributed directions): a greedy pass
    // drops a node when the walk without it tests fewer nodes. Leaves always stay (a leaf's own test is what decides whether its primitives are tested). The choice only moves time.
    // kept[o]: the nodes the closest-hit walk of octant o tests; kept[8]: the nodes the occlusion walk (octant 0's order for every ray) tests
    std::vector<std::vector<char>>& kept = out.kept; kept.assign(9, std::vector<char>(nn, 1));
    std::vector<char>& any_swap = out.any_swap; any_swap.assign(nn, 0);  // the occlusion walk's own order: second child first at these nodes (any order gives intersect_p's answer)
    bool nested = nn >= 3;  // (s->small / mid: no object instances; quadrics and masked triangles walk the same tables, k_trace's LINKS_G)
    for (uint32_t i = 0; i < nn && nested; ++i) {
      const rt_bvh_node& n = desc->nodes[i];
      if (n.n_prims != 0) continue;
      const uint32_t kids[2] = {i + 1, n.offset};
      for (uint32_t c : kids) for (int k = 0; k < 3; ++k) if (!(desc->nodes[c].bmin[k] >= n.bmin[k] && desc->nodes[c].bmax[k] <= n.bmax[k])) nested = false;  // (NaN bounds: no)
    }
    const char* prune_env = getenv("RTX_LDS_PRUNE");  // measurement / test knob, read per scene: 0 = every node is tested
    out.nodes_tested = nn;
    if (nested && !(prune_env && prune_env[0] == '0')) {
      const uint32_t K = nn <= 64 ? 8192u : 4096u;  // (the tables below: rays x (nodes + primitives) floats, 62 MB at the mid-size limit)
      struct CalRay { double o[3], d[3], t_max; };
      std::vector<CalRay> rays[9];  // closest-hit rays by octant; [8]: occlusion segments
      std::vector<double> cum(desc->n_tris + 1, 0.0);
      auto P = [&](uint32_t t, int v, int k) { return (double)desc->tri_p[9 * (size_t)t + 3 * v + k]; };
      // a quadric slot: whole spheres are calibrated as the spheres they are (centre and radius in world space, a uniform scale assumed); any other quadric starts no ray
      // and stops none, a masked triangle counts as opaque - the calibration is a cost model, every set of nodes it may choose leaves the hit records as they are
      auto quadric = [&](uint32_t t, double* c, double* rad) {
        if (!(desc->tri_meta[t].flags & RT_PRIM_SPHERE)) return 0;
        uint32_t k; memcpy(&k, desc->tri_p + 9 * (size_t)t + 6, 4);
        const rt_sphere& q = desc->spheres[k];
        if (q.kind != 0 || q.z_min > -q.radius || q.z_max < q.radius || q.phi_max < 6.28f) return 2;
        for (int a = 0; a < 3; ++a) c[a] = q.o2w[4 * a + 3];
        *rad = (double)q.radius * std::sqrt((double)q.o2w[0] * q.o2w[0] + (double)q.o2w[4] * q.o2w[4] + (double)q.o2w[8] * q.o2w[8]);
        return 1;
      };
      for (uint32_t t = 0; t < desc->n_tris; ++t) {
        double c[3], rad;
        const int qk = quadric(t, c, &rad);
        if (qk) { cum[t + 1] = cum[t] + (qk == 1 ? 12.566370614359172 * rad * rad : 0.0); continue; }
        const double e1[3] = {P(t, 1, 0) - P(t, 0, 0), P(t, 1, 1) - P(t, 0, 1), P(t, 1, 2) - P(t, 0, 2)}, e2[3] = {P(t, 2, 0) - P(t, 0, 0), P(t, 2, 1) - P(t, 0, 1), P(t, 2, 2) - P(t, 0, 2)};
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        const double ar = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        cum[t + 1] = cum[t] + (ar < 1e300 ? ar : 0.0);  // (NaN / inf areas: none)
      }
      std::vector<uint32_t> emitters;  // emitting triangles: where shadow rays go
      for (uint32_t i = 0; i < desc->n_lights; ++i) if (desc->lights[i].kind == RT_LIGHT_DIFFUSE_AREA && desc->lights[i].prim >= 0 && (uint32_t)desc->lights[i].prim < desc->n_tris) emitters.push_back((uint32_t)desc->lights[i].prim);
      unsigned long long rs = 0x9e3779b97f4a7c15ull;  // splitmix64: a fixed stream, the same tables for the same scene
      auto rnd = [&]() { rs += 0x9e3779b97f4a7c15ull; unsigned long long z = rs; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0); };
      const rt_bvh_node& root = desc->nodes[0];
      const double ctr[3] = {0.5 * ((double)root.bmin[0] + root.bmax[0]), 0.5 * ((double)root.bmin[1] + root.bmax[1]), 0.5 * ((double)root.bmin[2] + root.bmax[2])};
      // a point of primitive t and the normal there (false: a primitive the calibration has no points on)
      auto surface_point = [&](uint32_t t, double* p, double* n) {
        double c[3], rad;
        const int qk = quadric(t, c, &rad);
        if (qk == 2) return false;
        if (qk == 1) {
          const double z = 1.0 - 2.0 * rnd(), rr = std::sqrt(std::max(0.0, 1.0 - z * z)), ph = 6.283185307179586 * rnd();
          n[0] = rr * std::cos(ph); n[1] = rr * std::sin(ph); n[2] = z;
          for (int k = 0; k < 3; ++k) p[k] = c[k] + rad * n[k];
          return true;
        }
        const double su = std::sqrt(rnd()), b0 = 1.0 - su, b1 = rnd() * su, b2 = 1.0 - b0 - b1;
        for (int k = 0; k < 3; ++k) p[k] = b0 * P(t, 0, k) + b1 * P(t, 1, k) + b2 * P(t, 2, k);
        const double e1[3] = {P(t, 1, 0) - P(t, 0, 0), P(t, 1, 1) - P(t, 0, 1), P(t, 1, 2) - P(t, 0, 2)}, e2[3] = {P(t, 2, 0) - P(t, 0, 0), P(t, 2, 1) - P(t, 0, 1), P(t, 2, 2) - P(t, 0, 2)};
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nl > 0.0)) return false;
        for (int k = 0; k < 3; ++k) n[k] /= nl;
        return true;
      };
      // the parameter at which a calibration ray hits primitive t (Moeller-Trumbore in double; whole spheres as spheres), +inf: it does not
      auto prim_hit = [&](const CalRay& cr, uint32_t t) -> double {
        const double none = std::numeric_limits<double>::infinity();
        double sc_[3], srad;
        const int qk = quadric(t, sc_, &srad);
        if (qk == 2) return none;
        if (qk == 1) {
          const double oc[3] = {cr.o[0] - sc_[0], cr.o[1] - sc_[1], cr.o[2] - sc_[2]};
          const double A = cr.d[0] * cr.d[0] + cr.d[1] * cr.d[1] + cr.d[2] * cr.d[2], B = 2.0 * (oc[0] * cr.d[0] + oc[1] * cr.d[1] + oc[2] * cr.d[2]), Cq = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - srad * srad;
          const double disc = B * B - 4.0 * A * Cq;
          if (disc < 0.0) return none;
          const double sq = std::sqrt(disc), t0 = (-B - sq) / (2.0 * A), t1 = (-B + sq) / (2.0 * A), eps = 1e-6 * srad / std::sqrt(A);
          const double tt = t0 > eps ? t0 : t1;
          return tt > eps ? tt : none;
        }
        const double e1[3] = {P(t, 1, 0) - P(t, 0, 0), P(t, 1, 1) - P(t, 0, 1), P(t, 1, 2) - P(t, 0, 2)}, e2[3] = {P(t, 2, 0) - P(t, 0, 0), P(t, 2, 1) - P(t, 0, 1), P(t, 2, 2) - P(t, 0, 2)};
        const double pv[3] = {cr.d[1] * e2[2] - cr.d[2] * e2[1], cr.d[2] * e2[0] - cr.d[0] * e2[2], cr.d[0] * e2[1] - cr.d[1] * e2[0]};
        const double det = pv[0] * e1[0] + pv[1] * e1[1] + pv[2] * e1[2];
        if (det == 0.0) return none;
        const double tv[3] = {cr.o[0] - P(t, 0, 0), cr.o[1] - P(t, 0, 1), cr.o[2] - P(t, 0, 2)};
        const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
        if (u < 0.0 || u > 1.0) return none;
        const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
        const double v = (cr.d[0] * qv[0] + cr.d[1] * qv[1] + cr.d[2] * qv[2]) / det;
        if (v < 0.0 || u + v > 1.0) return none;
        const double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
        return tt > 1e-9 ? tt : none;
      };
      // the nearest hit of a calibration ray (a plain stack walk of the tree in double): the next vertex of the random walk below. false: it leaves the scene
      auto closest_hit = [&](const CalRay& cr, double& t_hit, uint32_t& prim) {
        t_hit = std::numeric_limits<double>::infinity(); prim = 0;
        std::vector<uint32_t> stk; stk.reserve(64); stk.push_back(0);
        const double inv[3] = {1.0 / cr.d[0], 1.0 / cr.d[1], 1.0 / cr.d[2]};
        while (!stk.empty()) {
          const uint32_t i = stk.back(); stk.pop_back();
          const rt_bvh_node& n = desc->nodes[i];
          double tn = -1e300, tf = 1e300;
          for (int k = 0; k < 3; ++k) { double a0 = ((double)n.bmin[k] - cr.o[k]) * inv[k], a1 = ((double)n.bmax[k] - cr.o[k]) * inv[k]; if (a0 > a1) std::swap(a0, a1); tn = std::max(tn, a0); tf = std::min(tf, a1); }
          if (!(tn <= tf && tf > 0.0 && tn < t_hit)) continue;
          if (n.n_prims != 0) { for (uint32_t t = n.offset; t < n.offset + n.n_prims; ++t) { const double tt = prim_hit(cr, t); if (tt < t_hit) { t_hit = tt; prim = t; } } }
          else { stk.push_back(i + 1); stk.push_back(n.offset); }
        }
        return t_hit < 1e300;
      };
      auto pick_tri = [&]() { const double x = rnd() * cum[desc->n_tris]; return (uint32_t)std::min<size_t>(desc->n_tris - 1, (size_t)(std::upper_bound(cum.begin(), cum.end(), x) - cum.begin()) - 1); };
      // Where the rays start: a RANDOM WALK, as the paths they stand for - the first vertex of a chain an area-weighted surface point (three of four chains leave it
      // towards the middle of the scene, where a path's next vertex usually lies), each further vertex where the chain's last ray hit, up to five deep or until a ray
      // leaves the scene; surfaces are thereby weighted by how much of the scene sees them (RTX_CAL_WALK=0, measurement knob: every ray from an area-weighted point).
      static const char* walk_env = getenv("RTX_CAL_WALK"); static const bool walk_on = !(walk_env && walk_env[0] == '0');
      bool have_vertex = false; int depth = 0; double vp[3] = {0, 0, 0}, vn[3] = {0, 0, 1};
      for (uint32_t r = 0; r < K && cum[desc->n_tris] > 0.0; ++r) {
        CalRay cr; cr.t_max = 1e300;
        double n[3];
        if (walk_on && have_vertex && depth < 5) { for (int k = 0; k < 3; ++k) { n[k] = vn[k]; cr.o[k] = vp[k] + n[k] * 1e-6 * (std::fabs(vp[k]) + 1.0); } }
        else {
          const uint32_t t = pick_tri();
          if (!surface_point(t, cr.o, n)) continue;
          const double side = ((ctr[0] - cr.o[0]) * n[0] + (ctr[1] - cr.o[1]) * n[1] + (ctr[2] - cr.o[2]) * n[2]) < 0.0 ? -1.0 : 1.0;
          const double flip = (rnd() < 0.25 ? -1.0 : 1.0) * side;
          for (int k = 0; k < 3; ++k) { n[k] *= flip; cr.o[k] += n[k] * 1e-6 * (std::fabs(cr.o[k]) + 1.0); }
          depth = 0; have_vertex = false;
          if (walk_on) { for (int k = 0; k < 3; ++k) { vp[k] = cr.o[k]; vn[k] = n[k]; } have_vertex = true; }
        }
        if ((r & 3u) == 3u) {  // every fourth ray: a shadow segment towards a point on an emitter (any surface point where nothing emits)
          double q[3], qn[3];
          if (!surface_point(emitters.empty() ? pick_tri() : emitters[(size_t)(rnd() * emitters.size()) % emitters.size()], q, qn)) continue;
          for (int k = 0; k < 3; ++k) cr.d[k] = q[k] - cr.o[k];
          cr.t_max = 1.0 - 1e-4;
          if (cr.d[0] == 0.0 || cr.d[1] == 0.0 || cr.d[2] == 0.0) continue;
          rays[8].push_back(cr);
          continue;
        }
        const double a[3] = {std::fabs(n[0]) > 0.9 ? 0.0 : 1.0, std::fabs(n[0]) > 0.9 ? 1.0 : 0.0, 0.0};
        double tx[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0]};
        const double tl = std::sqrt(tx[0] * tx[0] + tx[1] * tx[1] + tx[2] * tx[2]);
        for (int k = 0; k < 3; ++k) tx[k] /= tl;
        const double ty[3] = {n[1] * tx[2] - n[2] * tx[1], n[2] * tx[0] - n[0] * tx[2], n[0] * tx[1] - n[1] * tx[0]};
        const double r1 = rnd(), ph = 6.283185307179586 * rnd(), rr = std::sqrt(r1), cz = std::sqrt(1.0 - r1);
        for (int k = 0; k < 3; ++k) cr.d[k] = tx[k] * rr * std::cos(ph) + ty[k] * rr * std::sin(ph) + n[k] * cz;
        if (cr.d[0] == 0.0 || cr.d[1] == 0.0 || cr.d[2] == 0.0) continue;
        rays[(cr.d[0] < 0.0 ? 1 : 0) | (cr.d[1] < 0.0 ? 2 : 0) | (cr.d[2] < 0.0 ? 4 : 0)].push_back(cr);
        if (walk_on) {  // the chain's next vertex: where this ray hits, the normal there turned against the ray
          double th; uint32_t hp;
          have_vertex = closest_hit(cr, th, hp);
          if (have_vertex) {
            double c[3], rad;
            for (int k = 0; k < 3; ++k) vp[k] = cr.o[k] + th * cr.d[k];
            if (quadric(hp, c, &rad) == 1) { for (int k = 0; k < 3; ++k) vn[k] = (vp[k] - c[k]) / rad; }
            else {
              const double e1[3] = {P(hp, 1, 0) - P(hp, 0, 0), P(hp, 1, 1) - P(hp, 0, 1), P(hp, 1, 2) - P(hp, 0, 2)}, e2[3] = {P(hp, 2, 0) - P(hp, 0, 0), P(hp, 2, 1) - P(hp, 0, 1), P(hp, 2, 2) - P(hp, 0, 2)};
              vn[0] = e1[1] * e2[2] - e1[2] * e2[1]; vn[1] = e1[2] * e2[0] - e1[0] * e2[2]; vn[2] = e1[0] * e2[1] - e1[1] * e2[0];
              const double nl = std::sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
              if (nl > 0.0) { for (int k = 0; k < 3; ++k) vn[k] /= nl; } else have_vertex = false;
            }
            if (vn[0] * cr.d[0] + vn[1] * cr.d[1] + vn[2] * cr.d[2] > 0.0) for (int k = 0; k < 3; ++k) vn[k] = -vn[k];
            depth += 1;
          }
        }
      }
      // What a ray's walk meets does not depend on which nodes are tested: per set, ray and node the parameter at which the ray enters the node's box (+inf: it misses the
      // box), per ray and primitive the parameter of its hit (+inf: none; Moeller-Trumbore in double) - computed when first asked for and kept (NaN: not yet), so that a
      // candidate set of nodes costs table lookups and a ray pays only for the nodes some walk of it reaches.
      const float kNone = std::numeric_limits<float>::infinity(), kUnknown = std::numeric_limits<float>::quiet_NaN();
      std::vector<float> t_node[9], t_prim[9];
      for (int w = 0; w < 9; ++w) { t_node[w].assign(rays[w].size() * nn, kUnknown); t_prim[w].assign(rays[w].size() * (size_t)desc->n_tris, kUnknown); }
      auto node_t = [&](int w, size_t r, uint32_t i) -> float {
        float& out = t_node[w][r * nn + i];
        if (out == out) return out;
        const CalRay& cr = rays[w][r];
        const rt_bvh_node& n = desc->nodes[i];
        double tn = -1e300, tf = 1e300;
        for (int k = 0; k < 3; ++k) { const double iv = 1.0 / cr.d[k]; double a0 = ((double)n.bmin[k] - cr.o[k]) * iv, a1 = ((double)n.bmax[k] - cr.o[k]) * iv; if (a0 > a1) std::swap(a0, a1); tn = std::max(tn, a0); tf = std::min(tf, a1); }
        return out = (tn <= tf && tf > 0.0) ? (float)tn : kNone;
      };
      auto prim_t = [&](int w, size_t r, uint32_t t) -> float {
        float& out = t_prim[w][r * (size_t)desc->n_tris + t];
        if (out == out) return out;
        const double tt = prim_hit(rays[w][r], t);
        return out = tt < 3.0e38 ? (float)tt : kNone;
      };
      // The walk over the rays of set `w` when only kept[w] nodes are tested (a node that is not tested counts as passed). w < 8: BVH::intersect's order and its shrinking
      // t_max; w == 8: the occlusion walk - first child first at every node, over at the first primitive hit. Returns the node tests; n_pass / n_fail: per node, the rays
      // that reach it and would pass / fail its test (whether it is tested or not).
      auto simulate = [&](int w, std::vector<uint32_t>& n_pass, std::vector<uint32_t>& n_fail) {
        unsigned long long tests = 0;
        std::fill(n_pass.begin(), n_pass.end(), 0u); std::fill(n_fail.begin(), n_fail.end(), 0u);
        std::vector<uint32_t> stk; stk.reserve(64);
        const std::vector<char>& kp = kept[w];
        const unsigned oct = w < 8 ? (unsigned)w : 0u;  // (every ray of set w < 8 lies in octant w)
        for (size_t r = 0; r < rays[w].size(); ++r) {
          float t_max = (float)std::min(rays[w][r].t_max, 3.0e38); stk.clear(); uint32_t cur = 0; bool done = false;
          while (!done) {
            const rt_bvh_node& n = desc->nodes[cur];
            const bool would = node_t(w, r, cur) < t_max;
            (would ? n_pass : n_fail)[cur] += 1u;
            bool hit = true;
            if (kp[cur]) { tests += 1; hit = would; }
            if (hit && n.n_prims != 0) {
              for (uint32_t t = n.offset; t < n.offset + n.n_prims; ++t) {
                const float tp = prim_t(w, r, t);
                if (tp < t_max) { if (w == 8) { done = true; break; } t_max = tp; }
              }
            }
            if (done) break;
            if (hit && n.n_prims == 0) {
              const bool neg = w < 8 ? ((oct >> (n.axis < 2 ? n.axis : 2)) & 1u) != 0u : any_swap[cur] != 0;
              if (neg) { stk.push_back(cur + 1); cur = n.offset; } else { stk.push_back(n.offset); cur = cur + 1; }
            } else { if (stk.empty()) break; cur = stk.back(); stk.pop_back(); }
          }
        }
        return tests;
      };
      // Which interior nodes to test. A ray that fails node i's box fails every box below it (nested boxes, ordered products, no hit in between to move t_max), so without
      // i's test it runs into exactly the TESTED nodes nearest below i - frontier(i) of them, whatever the ray - and fails each; a ray that passes saves the one test. Dropping
      // i therefore changes the count by fail_i x (frontier(i) - 1) - pass_i over the rays that reach i: decided bottom-up (children first: their decisions are frontier(i)),
      // from the counts of one simulated walk; which rays reach i depends on the nodes above it, so the sweep is repeated on the new set until nothing changes (<= 6 times).
      // The occlusion walk's ORDER. intersect_p's answer does not depend on it, the work does: an occluded ray stops at its first hit. Per interior node and child X: p_X =
      // the share of the rays through the node that have an occluder below X, c_X = the node tests below X of a ray that looks through all of it (every node tested: the
      // order is chosen before the pruning); searching X first costs c_X + (1 - p_X) c_Y, so the child with the larger p / c goes first (array order on ties).
      static const char* order_env = getenv("RTX_LDS_ANY_ORDER");  // measurement knob: 0 = array order
      if (rays[8].size() >= 64 && !(order_env && order_env[0] == '0')) {
        std::vector<double> sum_tests(nn, 0.0); std::vector<uint32_t> occ(nn, 0u);
        struct Res { uint32_t tests; bool occ; };
        for (size_t r = 0; r < rays[8].size(); ++r) {
          const float t_max = (float)std::min(rays[8][r].t_max, 3.0e38);
          std::function<Res(uint32_t)> explore = [&](uint32_t i) -> Res {
            Res res{1u, false};
            if (!(node_t(8, r, i) < t_max)) return res;
            const rt_bvh_node& n = desc->nodes[i];
            if (n.n_prims != 0) { for (uint32_t t = n.offset; t < n.offset + n.n_prims; ++t) if (prim_t(8, r, t) < t_max) res.occ = true; return res; }
            const Res a = explore(i + 1), b = explore(n.offset);
            sum_tests[i + 1] += a.tests; sum_tests[n.offset] += b.tests; occ[i + 1] += a.occ ? 1u : 0u; occ[n.offset] += b.occ ? 1u : 0u;
            res.tests += a.tests + b.tests; res.occ = a.occ || b.occ;
            return res;
          };
          (void)explore(0);
        }
        for (uint32_t i = 0; i < nn; ++i) {
          const rt_bvh_node& n = desc->nodes[i];
          if (n.n_prims == 0) any_swap[i] = (double)occ[n.offset] * sum_tests[i + 1] > (double)occ[i + 1] * sum_tests[n.offset] ? 1 : 0;
        }
      }
      uint32_t nt = 0;
      std::vector<uint32_t> n_pass(nn), n_fail(nn), frontier(nn);
      for (int w = 0; w < 9; ++w) {
        if (rays[w].size() >= 64) {
          for (int sweep = 0; sweep < 6; ++sweep) {
            (void)simulate(w, n_pass, n_fail);
            bool changed = false;
            for (uint32_t i = nn; i-- > 0;) {
              const rt_bvh_node& n = desc->nodes[i];
              if (n.n_prims != 0) { frontier[i] = 1u; continue; }
              const uint32_t a = i + 1, b2 = n.offset;
              frontier[i] = (kept[w][a] ? 1u : frontier[a]) + (kept[w][b2] ? 1u : frontier[b2]);
              // (fewer than 8 rays through the node: no evidence - it stays tested, as the tree's builder meant it)
              const char keep = (n_pass[i] + n_fail[i] < 8u || (unsigned long long)n_fail[i] * (frontier[i] - 1u) > (unsigned long long)n_pass[i]) ? 1 : 0;
              if (keep != kept[w][i]) { kept[w][i] = keep; changed = true; }
            }
            if (!changed) break;
          }
        }
        if (w < 8) for (uint32_t i = 0; i < nn; ++i) nt += kept[w][i] ? 1u : 0u;
      }
      static const bool prune_report = getenv("RTX_PRUNE_REPORT") != nullptr;
      if (prune_report || want_stats) {  // measurement knob / rt_link_tables: simulated node tests per calibration ray, all nodes against the chosen ones
        for (int w = 0; w < 9; ++w) if (rays[w].size() >= 64) {
          std::vector<char> sel = kept[w]; kept[w].assign(nn, 1);
          const unsigned long long full = simulate(w, n_pass, n_fail); kept[w] = sel;
          const unsigned long long now = simulate(w, n_pass, n_fail);
          uint32_t k = 0; for (uint32_t i = 0; i < nn; ++i) k += sel[i] ? 1u : 0u;
          out.n_rays[w] = rays[w].size(); out.tests_all[w] = (double)full; out.tests_kept[w] = (double)now;
          if (prune_report) fprintf(stderr, "[rtx] prune set %d: %zu rays, %.2f -> %.2f node tests per ray, %u of %u nodes tested\n", w, rays[w].size(), (double)full / rays[w].size(), (double)now / rays[w].size(), k, nn);
        }
      }
      out.nodes_tested = (nt + 4u) / 8u;  // (the closest-hit walks' average over the octants)
    }
    {
      // the octant's order over ALL nodes (pos), each node's subtree as a range of it; then per node: the first TESTED node inside its subtree (enter) and the first after it (skip).
      // Layout: rows 0 - 7 (closest hit, one per octant), their 8 start nodes, row 8 (occlusion rays: octant 0's order, their own set of tested nodes), its start node.
      const size_t tab = (size_t)9 * nn + 9;
      std::vector<uint32_t>& link_full = out.link_full; std::vector<uint32_t>& link_kept = out.link_kept; link_full.assign(tab, 0u); link_kept.assign(tab, 0u);
      std::vector<uint32_t> order; std::vector<uint32_t> pos(nn), end_(nn);
      std::vector<uint32_t> size(nn, 1u);
      for (uint32_t i = nn; i-- > 0;) if (desc->nodes[i].n_prims == 0) size[i] = 1u + size[i + 1] + size[desc->nodes[i].offset];
      for (uint32_t row = 0; row < 9; ++row) {
        const uint32_t o = row < 8 ? row : 0u;
        order.clear(); st.clear();
        uint32_t cur = 0;
        for (;;) {
          pos[cur] = (uint32_t)order.size(); order.push_back(cur);
          const rt_bvh_node& n = desc->nodes[cur];
          if (n.n_prims == 0) {
            const bool neg = row < 8 ? ((o >> (n.axis < 2 ? n.axis : 2)) & 1u) != 0u : any_swap[cur] != 0;
            if (neg) { st.push_back(cur + 1); cur = n.offset; } else { st.push_back(n.offset); cur = cur + 1; }
          } else { if (st.empty()) break; cur = st.back(); st.pop_back(); }
        }
        for (uint32_t i = 0; i < nn; ++i) end_[i] = pos[i] + size[i];  // (a node's subtree is contiguous in any depth-first order)
        const size_t base = row < 8 ? (size_t)row * nn : (size_t)8 * nn + 8, start_at = row < 8 ? (size_t)8 * nn + row : (size_t)9 * nn + 8;
        for (int which = 0; which < 2; ++which) {
          std::vector<uint32_t>& L = which == 0 ? link_full : link_kept;
          auto tested = [&](uint32_t i) { return which == 0 || kept[row][i] != 0; };
          std::vector<uint32_t> next_tested(nn + 1, nn);  // by position: the first tested node at or after it
          for (uint32_t p_ = nn; p_-- > 0;) next_tested[p_] = tested(order[p_]) ? order[p_] : next_tested[p_ + 1];
          for (uint32_t i = 0; i < nn; ++i) {
            const uint32_t enter = pos[i] + 1 < end_[i] ? next_tested[pos[i] + 1] : nn;  // (inside the subtree a tested node always exists: its leaves)
            const uint32_t skp = end_[i] < nn ? next_tested[end_[i]] : nn;
            const rt_bvh_node& nd = desc->nodes[i];
            const int off_bits = mid ? RT_LINK_OFF_BITS(RT_MID_NODES) : RT_LINK_OFF_BITS(RT_SMALL_NODES);
            L[base + i] = nd.n_prims != 0 ? (0x80000000u | ((uint32_t)nd.n_prims << (16 + off_bits)) | ((uint32_t)nd.offset << 16) | skp) : ((enter << 16) | skp);
          }
          L[start_at] = next_tested[0];
        }
      }
    }
}
