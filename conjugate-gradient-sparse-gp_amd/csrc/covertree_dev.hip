// covertree_dev.hip -- row F3 with the O(N x centres) passes on the GPU: the cover-tree clustering of
// cggp/covertree.py:26-179 (the paper's inducing-point method), node for node what covertree.cpp builds.
//
// The construction is a sequential greedy r-net per level: which row seeds the next centre depends on what the
// previous centres removed, so the ACCEPTANCE stays on the host, in the reference's order.  What made the host
// version slow at realistic dimension (75 s for N = 2e5, D = 8: there almost every node is an r-neighbour of every
// other and each new centre scans nearly all remaining rows) are four all-pairs-shaped passes, and those run here
// as device filters over the resident X:
//   ball     rows of the parent within `radius` of the seed                     (Lloyd re-centring, :72-84)
//   take     rows of the parent's r-neighbours within `radius` of the new centre (:89-99)
//   voronoi  nearest candidate centre of every row, all parents of a level in one launch (:120-158)
//   reach    which candidate centres lie within `reach` of a new child, all children of a level in one launch (:105-116)
// The device only FILTERS: it returns row / candidate indices, the host forms every mean and keeps every list in the
// reference's order (rows of a child = its parent's r-neighbours in list order, each one's rows in list order), so
// the arithmetic that defines the tree is the host's.  The distance test is the host's too, operation for operation
// (differences, squares and sums in dimension order without fused multiply-add, IEEE square root, `<=` / `<` on the
// root) -- a row sits inside a ball on the device iff it does on the host, and the two constructions give the same
// tree bit for bit (tests/test_gpu_covertree.py).
#include <algorithm>
#include <new>

#include "covertree.h"
#include "mgp_common.h"

namespace {

using Node = MgpCtNode;

__device__ __forceinline__ double ct_dist_dev(const double* __restrict__ p, const double* __restrict__ q, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = __dsub_rn(p[d], q[d]);
    s = __dadd_rn(s, __dmul_rn(t, t));
  }
  return __dsqrt_rn(s);
}

// mode 0 (ball): rows i with state[i] == pp;  mode 1 (take): rows with state[i] >= 0 and flag[state[i]] set -- those
// within `radius` of pt are appended to out[1..] (out[0] = count; order arbitrary, the host sorts) and, in mode 1,
// leave the level (state = -1)
__global__ __launch_bounds__(256) void ct_filter_kernel(const double* __restrict__ X, long N, int D,
                                                        const double* __restrict__ pt, double radius,
                                                        int* __restrict__ state, int mode, int pp,
                                                        const unsigned char* __restrict__ flag,
                                                        unsigned long long* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int st = state[i];
  const bool elig = mode == 0 ? st == pp : (st >= 0 && flag[st] != 0);
  if (!elig) return;
  if (ct_dist_dev(pt, X + i * D, D) <= radius) {
    const unsigned long long k = atomicAdd(out, 1ull);
    out[1 + k] = (unsigned long long)i;
    if (mode == 1) state[i] = -1;
  }
}

// one workgroup per row: nearest of the row's candidate list (first on ties: `<` on the distance while the list
// position ascends, then (distance, position) over the workgroup)
__global__ __launch_bounds__(256) void ct_voronoi_kernel(const double* __restrict__ X, int D,
                                                         const long* __restrict__ vrows,
                                                         const long* __restrict__ row_off,
                                                         const int* __restrict__ row_cnt,
                                                         const int* __restrict__ cand,
                                                         const double* __restrict__ cpts, int* __restrict__ best) {
  __shared__ double sd[256];
  __shared__ int sk[256];
  const long rix = blockIdx.x;
  const double* x = X + vrows[rix] * D;
  const int* list = cand + row_off[rix];
  const int cnt = row_cnt[rix];
  double bd = INFINITY;
  int bk = 0x7fffffff;
  for (int k = threadIdx.x; k < cnt; k += 256) {
    const double dd = ct_dist_dev(cpts + (long)list[k] * D, x, D);
    if (dd < bd) {
      bd = dd;
      bk = k;
    }
  }
  sd[threadIdx.x] = bd;
  sk[threadIdx.x] = bk;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double od = sd[threadIdx.x + s];
      const int ok = sk[threadIdx.x + s];
      if (od < sd[threadIdx.x] || (od == sd[threadIdx.x] && ok < sk[threadIdx.x])) {
        sd[threadIdx.x] = od;
        sk[threadIdx.x] = ok;
      }
    }
    __syncthreads();
  }
  // every candidate at distance inf / NaN: the host loop keeps its initial best = 0
  if (threadIdx.x == 0) best[rix] = sk[0] == 0x7fffffff ? 0 : sk[0];
}

// one workgroup per child: mask[k] = its candidate k lies within `reach`
__global__ __launch_bounds__(256) void ct_reach_kernel(int D, const int* __restrict__ child,
                                                       const long* __restrict__ c_off, const int* __restrict__ c_cnt,
                                                       const int* __restrict__ cand, const double* __restrict__ cpts,
                                                       double reach, const long* __restrict__ m_off,
                                                       unsigned char* __restrict__ mask) {
  const long cix = blockIdx.x;
  const double* me = cpts + (long)child[cix] * D;
  const int* list = cand + c_off[cix];
  unsigned char* m = mask + m_off[cix];
  for (int k = threadIdx.x; k < c_cnt[cix]; k += 256)
    m[k] = ct_dist_dev(cpts + (long)list[k] * D, me, D) <= reach ? 1 : 0;
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    if (hipMalloc(&p, need + need / 4 + 4096) != hipSuccess) return -1;
    bytes = need + need / 4 + 4096;
    return 0;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

#define CT_HIP(call)                                                                                    \
  do {                                                                                                  \
    const hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                             \
      delete t;                                                                                         \
      return mgp_ct_host_fail(MGP_E_HIP, "covertree (device): %s failed: %s", #call, hipGetErrorString(e_)); \
    }                                                                                                   \
  } while (0)

}  // namespace

extern "C" int mgp_covertree_build_device(mgp_handle* h, const double* x, const double* x_dev, int64_t N, int D,
                                          double spatial_resolution, int num_levels, int lloyds, int voronoi,
                                          mgp_covertree** out) {
  if (!out) return mgp_ct_host_fail(MGP_E_BADARG, "covertree: NULL out");
  *out = nullptr;
  if (!h) return mgp_ct_host_fail(MGP_E_BADARG, "covertree (device): NULL handle");
  if (!x || !x_dev || N <= 0 || D <= 0)
    return mgp_ct_host_fail(MGP_E_SHAPE, "covertree: needs N > 0 rows of D > 0 columns (host and device copy)");
  if (N > 2147483647L) return mgp_ct_host_fail(MGP_E_SHAPE, "covertree (device): N too large");
  mgp_covertree* t = new (std::nothrow) mgp_covertree;
  if (!t) return mgp_ct_host_fail(MGP_E_HIP, "covertree: out of memory");
  hipStream_t s = h->stream;
  try {
    const int rc_root = mgp_ct_make_root(t, x, N, D, spatial_resolution, &num_levels, voronoi);
    if (rc_root != MGP_OK) {
      delete t;
      return rc_root;
    }
    const double max_radius = t->max_radius;
    DevBuf d_state, d_flag, d_out, d_pt, d_big;
    constexpr size_t kHead = 1024;  // rows of a filter result fetched with its count; more -> a second copy
    if (d_state.ensure((size_t)N * sizeof(int)) ||
        d_out.ensure(((size_t)N > kHead ? (size_t)N + 1 : kHead + 1) * sizeof(unsigned long long)) ||
        d_pt.ensure((size_t)D * sizeof(double))) {
      delete t;
      return mgp_ct_host_fail(MGP_E_NOMEM, "covertree (device): out of device memory");
    }
    unsigned long long* pin_out = nullptr;  // pinned: count + head of the list
    double* pin_pt = nullptr;
    CT_HIP(hipHostMalloc((void**)&pin_out, (kHead + 1) * sizeof(unsigned long long), hipHostMallocDefault));
    CT_HIP(hipHostMalloc((void**)&pin_pt, (size_t)D * sizeof(double), hipHostMallocDefault));
    struct PinFree {
      void *a, *b;
      ~PinFree() {
        (void)hipHostFree(a);
        (void)hipHostFree(b);
      }
    } pin_free{pin_out, pin_pt};

    // host mirror of the level's bookkeeping: owner[i] = position of the node holding row i in the parent level
    // (-1: taken into a child of the level under construction), pos[i] = its place in that node's row list
    std::vector<int> owner((size_t)N, 0);
    std::vector<int64_t> pos((size_t)N);
    for (int64_t i = 0; i < N; ++i) pos[(size_t)i] = i;
    std::vector<int64_t> got, rest;
    std::vector<double> point(D), mean(D);
    const unsigned nblk = (unsigned)((N + 255) / 256);

    // device filter -> `got` (unordered row indices)
    hipError_t ferr = hipSuccess;
    const char* fstep = "";
#define CT_F(call, name)            \
  do {                              \
    ferr = (call);                  \
    if (ferr != hipSuccess) {       \
      fstep = name;                 \
      return -1;                    \
    }                               \
  } while (0)
    auto filter = [&](int mode, int pp, const double* pt, double radius) -> int {
      memcpy(pin_pt, pt, (size_t)D * sizeof(double));
      CT_F(hipMemcpyAsync(d_pt.p, pin_pt, (size_t)D * sizeof(double), hipMemcpyHostToDevice, s), "copy of the point");
      CT_F(hipMemsetAsync(d_out.p, 0, sizeof(unsigned long long), s), "counter reset");
      hipLaunchKernelGGL(ct_filter_kernel, dim3(nblk), dim3(256), 0, s, x_dev, (long)N, D, (const double*)d_pt.p,
                         radius, (int*)d_state.p, mode, pp, (const unsigned char*)d_flag.p,
                         (unsigned long long*)d_out.p);
      CT_F(hipGetLastError(), "filter launch");
      CT_F(hipMemcpyAsync(pin_out, d_out.p, (kHead + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s),
           "copy of the result head");
      CT_F(hipStreamSynchronize(s), "synchronize");
      const size_t cnt = (size_t)pin_out[0];
      got.resize(cnt);
      const size_t head = cnt < kHead ? cnt : kHead;
      for (size_t k = 0; k < head; ++k) got[k] = (int64_t)pin_out[1 + k];
      if (cnt > kHead) {
        rest.resize(cnt - kHead);
        CT_F(hipMemcpy(rest.data(), (unsigned long long*)d_out.p + 1 + kHead, (cnt - kHead) * sizeof(int64_t),
                       hipMemcpyDeviceToHost), "copy of the result tail");
        for (size_t k = kHead; k < cnt; ++k) got[k] = rest[k - kHead];
      }
      return 0;
    };
#undef CT_F

    std::vector<int> rnb_rank;
    std::vector<unsigned char> flag_host;
    for (int level = 1; level < num_levels; ++level) {
      const double radius = max_radius / std::ldexp(1.0, level);
      const double reach = 4.0 * (1.0 - 1.0 / std::ldexp(1.0, num_levels - level)) * radius;
      const std::vector<int> parents = t->levels[level - 1];
      const size_t np = parents.size();
      // node id -> position in the parent level
      std::vector<int> ppos(t->nodes.size(), -1);
      for (size_t k = 0; k < np; ++k) ppos[(size_t)parents[k]] = (int)k;
      for (size_t k = 0; k < np; ++k) {
        const std::vector<int64_t>& rows = t->nodes[parents[k]].rows;
        for (size_t j = 0; j < rows.size(); ++j) {
          owner[(size_t)rows[j]] = (int)k;
          pos[(size_t)rows[j]] = (int64_t)j;
        }
      }
      CT_HIP(hipMemcpyAsync(d_state.p, owner.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, s));
      if (d_flag.ensure(np)) {
        delete t;
        return mgp_ct_host_fail(MGP_E_NOMEM, "covertree (device): out of device memory");
      }
      flag_host.assign(np, 0);
      rnb_rank.assign(np, -1);
      // the parents' row lists stay as they were at the start of the level; a cursor skips rows taken since
      std::vector<std::vector<int64_t>> plist(np);
      for (size_t k = 0; k < np; ++k) plist[k].swap(t->nodes[parents[k]].rows);

      for (size_t pk = 0; pk < np; ++pk) {
        const int pid = parents[pk];
        {  // r-neighbour flags and ranks of this parent
          const Node& P = t->nodes[pid];
          std::fill(flag_host.begin(), flag_host.end(), 0);
          for (size_t a = 0; a < P.rnb.size(); ++a) {
            flag_host[(size_t)ppos[(size_t)P.rnb[a]]] = 1;
            rnb_rank[(size_t)ppos[(size_t)P.rnb[a]]] = (int)a;
          }
          CT_HIP(hipMemcpyAsync(d_flag.p, flag_host.data(), np, hipMemcpyHostToDevice, s));
          CT_HIP(hipStreamSynchronize(s));  // flag_host is reused
        }
        size_t cursor = 0;
        const std::vector<int64_t>& mine = plist[pk];
        while (true) {
          while (cursor < mine.size() && owner[(size_t)mine[cursor]] != (int)pk) ++cursor;
          if (cursor >= mine.size()) break;
          const double* seed = x + mine[cursor] * D;
          for (int d = 0; d < D; ++d) point[d] = seed[d];
          if (lloyds) {
            if (filter(0, (int)pk, seed, radius)) {
              delete t;
              return mgp_ct_host_fail(MGP_E_HIP, "covertree (device): ball query failed at %s: %s", fstep, hipGetErrorString(ferr));
            }
            std::sort(got.begin(), got.end(), [&](int64_t a, int64_t b) { return pos[(size_t)a] < pos[(size_t)b]; });
            std::fill(mean.begin(), mean.end(), 0.0);
            for (int64_t r : got)
              for (int d = 0; d < D; ++d) mean[d] += x[r * D + d];
            const int64_t cnt = (int64_t)got.size();
            for (int d = 0; d < D; ++d) mean[d] /= (double)cnt;  // cnt >= 1: the seed itself
            bool clash = false;
            const Node& P = t->nodes[pid];
            for (size_t a = 0; a < P.rnb.size() && !clash; ++a)
              for (int c : t->nodes[P.rnb[a]].children)
                if (mgp_ct_dist(mean.data(), t->nodes[c].point.data(), D) < radius) {
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
          if (filter(1, (int)pk, point.data(), radius)) {
            delete t;
            return mgp_ct_host_fail(MGP_E_HIP, "covertree (device): take query failed at %s: %s", fstep, hipGetErrorString(ferr));
          }
          // the reference's order: the parent's r-neighbours in list order, each one's rows in list order
          std::sort(got.begin(), got.end(), [&](int64_t a, int64_t b) {
            const int ra = rnb_rank[(size_t)owner[(size_t)a]], rb = rnb_rank[(size_t)owner[(size_t)b]];
            return ra != rb ? ra < rb : pos[(size_t)a] < pos[(size_t)b];
          });
          C.rows.assign(got.begin(), got.end());
          for (int64_t r : got) owner[(size_t)r] = -1;
          t->levels[level].push_back(cid);
          t->nodes[pid].children.push_back(cid);
        }
        for (int nb : t->nodes[pid].rnb) rnb_rank[(size_t)ppos[(size_t)nb]] = -1;
      }

      // ---- candidate lists of the level: nearby(P) = children of P's r-neighbours, in that order (:105-128)
      const std::vector<int>& kids = t->levels[level];
      const size_t nk = kids.size();
      std::vector<int> kpos(t->nodes.size(), -1);
      for (size_t k = 0; k < nk; ++k) kpos[(size_t)kids[k]] = (int)k;
      std::vector<double> cpts(nk * (size_t)D);
      for (size_t k = 0; k < nk; ++k)
        for (int d = 0; d < D; ++d) cpts[k * D + d] = t->nodes[kids[k]].point[d];
      std::vector<long> near_off(np + 1, 0);
      for (size_t pk = 0; pk < np; ++pk) {
        size_t c = 0;
        for (int nb : t->nodes[parents[pk]].rnb) c += t->nodes[nb].children.size();
        near_off[pk + 1] = near_off[pk] + (long)c;
      }
      std::vector<int> near((size_t)near_off[np]);
      for (size_t pk = 0; pk < np; ++pk) {
        long o = near_off[pk];
        for (int nb : t->nodes[parents[pk]].rnb)
          for (int c : t->nodes[nb].children) near[(size_t)o++] = kpos[(size_t)c];
      }
      // ---- r-neighbours of the children (:105-116): one launch, one mask byte per (child, candidate)
      {
        std::vector<int> child(nk);
        std::vector<long> c_off(nk), m_off(nk + 1, 0);
        std::vector<int> c_cnt(nk);
        size_t q = 0;
        for (size_t pk = 0; pk < np; ++pk)
          for (int c : t->nodes[parents[pk]].children) {
            child[q] = kpos[(size_t)c];
            c_off[q] = near_off[pk];
            c_cnt[q] = (int)(near_off[pk + 1] - near_off[pk]);
            m_off[q + 1] = m_off[q] + c_cnt[q];
            ++q;
          }
        const size_t mbytes = (size_t)m_off[nk];
        const size_t b_cpts = nk * (size_t)D * sizeof(double), b_near = near.size() * sizeof(int),
                     b_child = nk * sizeof(int), b_off = nk * sizeof(long), b_cnt = nk * sizeof(int),
                     b_moff = nk * sizeof(long);
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t total = al(b_cpts) + al(b_near) + al(b_child) + al(b_off) + al(b_cnt) + al(b_moff) + al(mbytes);
        if (d_big.ensure(total)) {
          delete t;
          return mgp_ct_host_fail(MGP_E_NOMEM, "covertree (device): out of device memory (%zu bytes of candidate lists)",
                                  total);
        }
        char* base = (char*)d_big.p;
        double* g_cpts = (double*)base;
        int* g_near = (int*)(base + al(b_cpts));
        int* g_child = (int*)((char*)g_near + al(b_near));
        long* g_off = (long*)((char*)g_child + al(b_child));
        int* g_cnt = (int*)((char*)g_off + al(b_off));
        long* g_moff = (long*)((char*)g_cnt + al(b_cnt));
        unsigned char* g_mask = (unsigned char*)((char*)g_moff + al(b_moff));
        CT_HIP(hipMemcpyAsync(g_cpts, cpts.data(), b_cpts, hipMemcpyHostToDevice, s));
        if (b_near) CT_HIP(hipMemcpyAsync(g_near, near.data(), b_near, hipMemcpyHostToDevice, s));
        CT_HIP(hipMemcpyAsync(g_child, child.data(), b_child, hipMemcpyHostToDevice, s));
        CT_HIP(hipMemcpyAsync(g_off, c_off.data(), b_off, hipMemcpyHostToDevice, s));
        CT_HIP(hipMemcpyAsync(g_cnt, c_cnt.data(), b_cnt, hipMemcpyHostToDevice, s));
        CT_HIP(hipMemcpyAsync(g_moff, m_off.data(), b_moff, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ct_reach_kernel, dim3((unsigned)nk), dim3(256), 0, s, D, (const int*)g_child,
                           (const long*)g_off, (const int*)g_cnt, (const int*)g_near, (const double*)g_cpts, reach,
                           (const long*)g_moff, g_mask);
        std::vector<unsigned char> mask(mbytes);
        if (mbytes) CT_HIP(hipMemcpyAsync(mask.data(), g_mask, mbytes, hipMemcpyDeviceToHost, s));
        CT_HIP(hipStreamSynchronize(s));
        q = 0;
        for (size_t pk = 0; pk < np; ++pk)
          for (int c : t->nodes[parents[pk]].children) {
            Node& C = t->nodes[c];
            C.rnb.clear();
            const unsigned char* m = mask.data() + m_off[q];
            const int* list = near.data() + near_off[pk];
            for (int k = 0; k < c_cnt[q]; ++k)
              if (m[k]) C.rnb.push_back(kids[(size_t)list[k]]);
            ++q;
          }
        // ---- Voronoi reassignment (:120-158): every row of every parent against its parent's candidates, one launch
        if (voronoi) {
          size_t nrows = 0;
          for (size_t pk = 0; pk < np; ++pk) {
            const Node& P = t->nodes[parents[pk]];
            if (P.has_vor && !P.vor.empty()) {
              if (near_off[pk + 1] == near_off[pk]) {
                delete t;
                return mgp_ct_host_fail(MGP_E_BADARG, "covertree: a parent with rows has no candidate children");
              }
              nrows += P.vor.size();
            }
          }
          if (nrows) {
            std::vector<long> vrows(nrows), r_off(nrows);
            std::vector<int> r_cnt(nrows), r_par(nrows);
            size_t w = 0;
            for (size_t pk = 0; pk < np; ++pk) {
              const Node& P = t->nodes[parents[pk]];
              if (!P.has_vor || P.vor.empty()) continue;
              for (int64_t r : P.vor) {
                vrows[w] = (long)r;
                r_off[w] = near_off[pk];
                r_cnt[w] = (int)(near_off[pk + 1] - near_off[pk]);
                r_par[w] = (int)pk;
                ++w;
              }
            }
            DevBuf d_v;
            const size_t b_rows = nrows * sizeof(long), b_roff = nrows * sizeof(long), b_rcnt = nrows * sizeof(int),
                         b_best = nrows * sizeof(int);
            if (d_v.ensure(al(b_rows) + al(b_roff) + al(b_rcnt) + al(b_best))) {
              delete t;
              return mgp_ct_host_fail(MGP_E_NOMEM, "covertree (device): out of device memory");
            }
            char* vb = (char*)d_v.p;
            long* g_rows = (long*)vb;
            long* g_roff = (long*)(vb + al(b_rows));
            int* g_rcnt = (int*)((char*)g_roff + al(b_roff));
            int* g_best = (int*)((char*)g_rcnt + al(b_rcnt));
            CT_HIP(hipMemcpyAsync(g_rows, vrows.data(), b_rows, hipMemcpyHostToDevice, s));
            CT_HIP(hipMemcpyAsync(g_roff, r_off.data(), b_roff, hipMemcpyHostToDevice, s));
            CT_HIP(hipMemcpyAsync(g_rcnt, r_cnt.data(), b_rcnt, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(ct_voronoi_kernel, dim3((unsigned)nrows), dim3(256), 0, s, x_dev, D, (const long*)g_rows,
                               (const long*)g_roff, (const int*)g_rcnt, (const int*)g_near, (const double*)g_cpts,
                               g_best);
            std::vector<int> best(nrows);
            CT_HIP(hipMemcpyAsync(best.data(), g_best, b_best, hipMemcpyDeviceToHost, s));
            CT_HIP(hipStreamSynchronize(s));
            // buckets in the reference's order: parents in level order, rows of a parent in list order
            size_t a = 0;
            while (a < nrows) {
              const int pk = r_par[a];
              size_t b = a;
              while (b < nrows && r_par[b] == pk) ++b;
              const int* list = near.data() + near_off[(size_t)pk];
              const int cnt = (int)(near_off[(size_t)pk + 1] - near_off[(size_t)pk]);
              // every candidate of this parent becomes a Voronoi node (an empty bucket too), rows appended per bucket
              std::vector<std::vector<int64_t>> bucket((size_t)cnt);
              for (size_t q2 = a; q2 < b; ++q2) bucket[(size_t)best[q2]].push_back((int64_t)vrows[q2]);
              for (int k = 0; k < cnt; ++k) {
                Node& C = t->nodes[kids[(size_t)list[k]]];
                C.has_vor = true;
                C.vor.insert(C.vor.end(), bucket[(size_t)k].begin(), bucket[(size_t)k].end());
                C.rows = C.vor;
              }
              a = b;
            }
          }
        }
      }
    }
  } catch (const std::bad_alloc&) {
    delete t;
    return mgp_ct_host_fail(MGP_E_HIP, "covertree: out of memory");
  }
  *out = t;
  return MGP_OK;
}
