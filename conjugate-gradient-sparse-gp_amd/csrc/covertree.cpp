// covertree.cpp -- next row F3: the cover-tree clustering of cggp/covertree.py:26-179 (host code).
//
// The construction is a sequential greedy r-net per level (each new centre depends on which rows the
// previous centres removed), pointer-heavy and run once per inducing-point update, so it lives on the
// host: plain C++ over row indices (the reference copies and concatenates row blocks per node; here
// a node holds an index list into the caller's X and every pass is a stable partition of such lists).
// Semantics kept from the reference: root = mean of X with the largest distance as radius (rounded up
// to spatial_resolution * 2^(levels-1) when a resolution is given, :54-56); per level the radius
// halves; a parent seeds children from its first remaining row, optionally re-centred on the mean of
// the seed's radius-ball unless that lands within `radius` of an existing nearby centre (:72-84); a
// new centre takes every remaining row within `radius` from the parent and its r-neighbours, in
// r-neighbour order (:89-99); children become r-neighbours when their centres are within
// 4 (1 - 2^-(levels-level)) * radius (:65,105-116); and, with voronoi=1, the rows of each parent are
// then reassigned to the nearest centre among the children of its r-neighbours (first on ties, :120-158).
// Distances are Euclidean on the raw inputs (the reference ignores its distance argument, :36-44).
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <new>
#include <vector>

#include "covertree.h"

namespace {

thread_local char g_err[256] = "";

using Node = MgpCtNode;

inline double dist(const double* p, const double* q, int D) { return mgp_ct_dist(p, q, D); }

}  // namespace

int mgp_ct_host_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define host_fail mgp_ct_host_fail

int mgp_ct_make_root(mgp_covertree* t, const double* x, int64_t N, int D, double spatial_resolution, int* num_levels_io,
                     int voronoi) {
  int num_levels = *num_levels_io;
  t->D = D;
  t->N = N;
  t->nodes.emplace_back();
  Node& root = t->nodes.back();
  root.point.assign(D, 0.0);
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < D; ++d) root.point[d] += x[i * D + d];
  for (int d = 0; d < D; ++d) root.point[d] /= (double)N;
  double max_radius = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    const double r = dist(root.point.data(), x + i * D, D);
    if (r > max_radius) max_radius = r;
  }
  if (spatial_resolution > 0.0) {
    if (!(max_radius > 0.0))
      return host_fail(MGP_E_BADARG, "covertree: all rows coincide, no level count for a resolution");
    num_levels = (int)std::ceil(std::log2(max_radius / spatial_resolution)) + 1;
    max_radius = spatial_resolution * std::ldexp(1.0, num_levels - 1);
  }
  if (num_levels < 1 || num_levels > 60)
    return host_fail(MGP_E_BADARG, "covertree: %d levels (resolution larger than the data radius?)", num_levels);
  t->max_radius = max_radius;
  root.rows.resize(N);
  for (int64_t i = 0; i < N; ++i) root.rows[i] = i;
  root.rnb.push_back(0);
  if (voronoi) {
    root.vor = root.rows;
    root.has_vor = true;
  }
  t->levels.assign(num_levels, {});
  t->levels[0].push_back(0);
  *num_levels_io = num_levels;
  return MGP_OK;
}

extern "C" const char* mgp_host_last_error(void) { return g_err; }

extern "C" int mgp_covertree_build(const double* x, int64_t N, int D, double spatial_resolution, int num_levels,
                                   int lloyds, int voronoi, mgp_covertree** out) {
  if (!out) return host_fail(MGP_E_BADARG, "covertree: NULL out");
  *out = nullptr;
  if (!x || N <= 0 || D <= 0) return host_fail(MGP_E_SHAPE, "covertree: needs N > 0 rows of D > 0 columns");
  mgp_covertree* t = new (std::nothrow) mgp_covertree;
  if (!t) return host_fail(MGP_E_HIP, "covertree: out of memory");
  try {
    const int rc_root = mgp_ct_make_root(t, x, N, D, spatial_resolution, &num_levels, voronoi);
    if (rc_root != MGP_OK) {
      delete t;
      return rc_root;
    }
    const double max_radius = t->max_radius;

    std::vector<double> point(D);
    std::vector<int64_t> keep;
    for (int level = 1; level < num_levels; ++level) {
      const double radius = max_radius / std::ldexp(1.0, level);
      const double reach = 4.0 * (1.0 - 1.0 / std::ldexp(1.0, num_levels - level)) * radius;
      const std::vector<int>& parents = t->levels[level - 1];
      for (int pid : parents) {
        while (!t->nodes[pid].rows.empty()) {
          Node& P = t->nodes[pid];
          const double* seed = x + P.rows[0] * D;
          for (int d = 0; d < D; ++d) point[d] = seed[d];
          if (lloyds) {
            int64_t cnt = 0;
            std::vector<double> mean(D, 0.0);
            for (int64_t r : P.rows)
              if (dist(seed, x + r * D, D) <= radius) {
                for (int d = 0; d < D; ++d) mean[d] += x[r * D + d];
                ++cnt;
              }
            for (int d = 0; d < D; ++d) mean[d] /= (double)cnt;  // cnt >= 1: the seed itself
            bool clash = false;
            for (size_t a = 0; a < P.rnb.size() && !clash; ++a)
              for (int c : t->nodes[P.rnb[a]].children)
                if (dist(mean.data(), t->nodes[c].point.data(), D) < radius) {
                  clash = true;
                  break;
                }
            if (!clash) point = mean;
          }
          const int cid = (int)t->nodes.size();
          t->nodes.emplace_back();
          Node& C = t->nodes.back();
          C.point = point;
          C.parent = pid;
          C.rnb.push_back(cid);
          Node& P2 = t->nodes[pid];
          for (int nb : P2.rnb) {
            std::vector<int64_t>& src = t->nodes[nb].rows;
            keep.clear();
            for (int64_t r : src) {
              if (dist(point.data(), x + r * D, D) <= radius) C.rows.push_back(r);
              else keep.push_back(r);
            }
            src.swap(keep);
          }
          t->levels[level].push_back(cid);
          P2.children.push_back(cid);
        }
      }
      std::vector<int> nearby;
      for (int pid : parents) {
        const Node& P = t->nodes[pid];
        nearby.clear();
        for (int nb : P.rnb)
          for (int c : t->nodes[nb].children) nearby.push_back(c);
        for (int c : P.children) {
          Node& C = t->nodes[c];
          C.rnb.clear();
          for (int o : nearby)
            if (dist(t->nodes[o].point.data(), C.point.data(), D) <= reach) C.rnb.push_back(o);
        }
      }
      if (voronoi) {
        for (int pid : parents) {
          const Node& P = t->nodes[pid];
          if (!P.has_vor || P.vor.empty()) continue;
          nearby.clear();
          for (int nb : P.rnb)
            for (int c : t->nodes[nb].children) nearby.push_back(c);
          if (nearby.empty()) {
            delete t;
            return host_fail(MGP_E_BADARG, "covertree: a parent with rows has no candidate children");
          }
          std::vector<std::vector<int64_t>> bucket(nearby.size());
          for (int64_t r : P.vor) {
            size_t best = 0;
            double bd = INFINITY;
            for (size_t k = 0; k < nearby.size(); ++k) {
              const double dd = dist(t->nodes[nearby[k]].point.data(), x + r * D, D);
              if (dd < bd) {
                bd = dd;
                best = k;
              }
            }
            bucket[best].push_back(r);
          }
          for (size_t k = 0; k < nearby.size(); ++k) {
            Node& C = t->nodes[nearby[k]];
            C.has_vor = true;
            C.vor.insert(C.vor.end(), bucket[k].begin(), bucket[k].end());
            C.rows = C.vor;
          }
        }
      }
    }
  } catch (const std::bad_alloc&) {
    delete t;
    return host_fail(MGP_E_HIP, "covertree: out of memory");
  }
  *out = t;
  return MGP_OK;
}

extern "C" void mgp_covertree_destroy(mgp_covertree* t) { delete t; }

extern "C" int mgp_covertree_num_levels(const mgp_covertree* t) { return t ? (int)t->levels.size() : 0; }

extern "C" int64_t mgp_covertree_level_size(const mgp_covertree* t, int level) {
  if (!t || level < 0 || level >= (int)t->levels.size()) return -1;
  return (int64_t)t->levels[level].size();
}

extern "C" double mgp_covertree_level_radius(const mgp_covertree* t, int level) {
  if (!t || level < 0 || level >= (int)t->levels.size()) return -1.0;
  return t->max_radius / std::ldexp(1.0, level);
}

extern "C" int mgp_covertree_level_nodes(const mgp_covertree* t, int level, double* points, int64_t* parent,
                                         int64_t* counts) {
  if (!t || level < 0 || level >= (int)t->levels.size()) return host_fail(MGP_E_BADARG, "covertree: bad level");
  // a parent is reported by its position inside the level above (-1 for the root)
  std::vector<int64_t> pos;
  if (level > 0) {
    pos.assign(t->nodes.size(), -1);
    const std::vector<int>& up = t->levels[level - 1];
    for (size_t k = 0; k < up.size(); ++k) pos[up[k]] = (int64_t)k;
  }
  const std::vector<int>& ids = t->levels[level];
  for (size_t k = 0; k < ids.size(); ++k) {
    const Node& nd = t->nodes[ids[k]];
    if (points)
      for (int d = 0; d < t->D; ++d) points[k * t->D + d] = nd.point[d];
    if (parent) parent[k] = level > 0 ? pos[nd.parent] : -1;
    if (counts) counts[k] = (int64_t)nd.rows.size();
  }
  return MGP_OK;
}

extern "C" int mgp_covertree_level_rows(const mgp_covertree* t, int level, int64_t* offsets, int64_t* rows) {
  if (!t || level < 0 || level >= (int)t->levels.size()) return host_fail(MGP_E_BADARG, "covertree: bad level");
  if (!offsets) return host_fail(MGP_E_BADARG, "covertree: NULL offsets");
  const std::vector<int>& ids = t->levels[level];
  int64_t o = 0;
  for (size_t k = 0; k < ids.size(); ++k) {
    const Node& nd = t->nodes[ids[k]];
    offsets[k] = o;
    if (rows)
      for (int64_t r : nd.rows) rows[o++] = r;
    else
      o += (int64_t)nd.rows.size();
  }
  offsets[ids.size()] = o;
  return MGP_OK;
}
