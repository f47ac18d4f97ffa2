// covertree.h -- the cover-tree object shared by the host construction (covertree.cpp) and the GPU-assisted one
// (covertree_dev.hip): nodes hold row-index lists into the caller's X.
#pragma once
#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

#include "../../include/mgp.h"

struct MgpCtNode {
  std::vector<double> point;
  int parent = -1;             // node id
  std::vector<int> children;   // node ids, creation order
  std::vector<int> rnb;        // r-neighbours (node ids of the same level)
  std::vector<int64_t> rows;   // data rows held
  std::vector<int64_t> vor;    // accumulated Voronoi rows
  bool has_vor = false;
};

struct mgp_covertree {
  int D = 0;
  int64_t N = 0;
  double max_radius = 0.0;
  std::deque<MgpCtNode> nodes;           // stable references while growing
  std::vector<std::vector<int>> levels;  // node ids per level
};

// Euclidean distance exactly as every decision of the construction forms it: differences, squares and sums one by one
// in dimension order, then the square root (no fused multiply-add -- the device kernels repeat these operations)
inline double mgp_ct_dist(const double* p, const double* q, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = p[d] - q[d];
    s += t * t;
  }
  return std::sqrt(s);
}

int mgp_ct_host_fail(int code, const char* fmt, ...);
// root node, level count and radius (cggp/covertree.py:50-66); shared by both constructions
int mgp_ct_make_root(mgp_covertree* t, const double* x, int64_t N, int D, double spatial_resolution, int* num_levels,
                     int voronoi);
