// cluster.hip -- next row F1: nearest-centre assignment and cluster statistics
// (cggp/optimize.py:41-98, cggp/selection.py:14-32, built on cggp/distance.py).
//
// Every distance the reference offers is a monotone function of the (lengthscale-scaled)
// squared distance, so one fused N x M sweep finds argmin_m ||a_i - b_m||^2 per row (first index
// on ties: strict '<' while m ascends) and the epilogue maps the winning value to the requested
// distance: squared euclidean on raw inputs (ops.square_distance, optimize.py:50), euclidean
// (distance.py:9-11), covariance 2 var (1 - rho) (:15-22) or correlation 1 - rho (:24-30).
// Cluster sums/counts use the transpose sweep (lane = cluster, rows broadcast from LDS, ordered
// two-stage reduction): deterministic, no float atomics.
#include "mgp_common.h"

namespace {

constexpr int NT = 256;

// RPT rows per thread: one LDS read of a centre feeds RPT distance chains (with one row per thread the
// kernel was LDS-issue bound: 5 operand reads per 13 VALU instructions)
// DIRECT (dist_type 1, the reference's `euclid_distance`, distance.py:9-11): the squared distance is
// the sum of squared differences, exactly 0 for coincident points, as `norm(x - y)` gives; the other
// types use GPflow's expansion |a|^2 + |b|^2 - 2 a.b, as `ops.square_distance` and the kernels do.
template <typename T, int DP, int KIND, int RPT, bool DIRECT>
__global__ __launch_bounds__(NT) void nearest_kernel(const T* __restrict__ X, long N, const T* __restrict__ Z,
                                                     long M, int D, SweepParams prm, int dist_type,
                                                     long* __restrict__ idx, T* __restrict__ best) {
  constexpr int TB = 128;
  constexpr int PS = (DP + 1 + 1) & ~1;
  __shared__ __attribute__((aligned(16))) T tile[TB * PS];
  const int t = threadIdx.x;
  const long i0 = (long)blockIdx.x * (NT * RPT) + t;
  T a[RPT][DP];
  T a2[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    long i = i0 + (long)q * NT;
    if (i >= N) i = N - 1;
    a2[q] = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      T v = d < D ? X[i * D + d] * (T)prm.inv_ls[d] : (T)0;
      a[q][d] = v;
      a2[q] = mgp_fma(v, v, a2[q]);
    }
  }
  T bs[RPT];
  int bj[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    bs[q] = (T)INFINITY;
    bj[q] = 0;
  }
  for (long j0 = 0; j0 < M; j0 += TB) {
    __syncthreads();
    if (t < TB) {
      const long j = j0 + t;
      T* p = &tile[t * PS];
      T s = 0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        T v = (d < D && j < M) ? Z[j * D + d] * (T)prm.inv_ls[d] : (T)0;
        s = mgp_fma(v, v, s);
        p[d] = DIRECT ? v : v + v;
      }
      p[DP] = j < M ? (DIRECT ? (T)0 : s) : (T)INFINITY;
    }
    __syncthreads();
    const int lim = (M - j0) < TB ? (int)(M - j0) : TB;
    for (int jj = 0; jj < lim; ++jj) {
      const T* p = &tile[jj * PS];
      T pv[DP + 1];
#pragma unroll
      for (int d = 0; d <= DP; ++d) pv[d] = p[d];
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        T s;
        if (DIRECT) {
          s = pv[DP];  // 0, or +inf past the last centre
#pragma unroll
          for (int d = 0; d < DP; ++d) {
            const T df = a[q][d] - pv[d];
            s = mgp_fma(df, df, s);
          }
        } else {
          s = pv[DP] + a2[q];  // |a|^2 + |b|^2 - 2 a.b  (GPflow's expansion)
#pragma unroll
          for (int d = 0; d < DP; ++d) s = mgp_fma(-a[q][d], pv[d], s);
        }
        if (s < bs[q]) {  // strict: first index on ties while j ascends
          bs[q] = s;
          bj[q] = (int)j0 + jj;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long i = i0 + (long)q * NT;
    if (i >= N) continue;
    idx[i] = bj[q];
    if (best != nullptr) {
      T o;
      if (dist_type == 0) {
        o = bs[q];
      } else if (dist_type == 1) {
        o = mgp_sqrt(bs[q] > 0 ? bs[q] : (T)0);
      } else {
        const T rho = mgp_profile<KIND, T>(-bs[q], (T)prm.clamp);  // k / variance
        o = dist_type == 2 ? (T)2 * (T)prm.variance * ((T)1 - rho) : (T)1 - rho;
      }
      best[i] = o;
    }
  }
}

// partial[chunk][2][M]: sums and counts of the rows of one chunk; lane = cluster
template <typename T>
__global__ __launch_bounds__(NT) void cluster_stats_kernel(const long* __restrict__ idx, const T* __restrict__ y,
                                                           long N, long M, long rows_per_chunk,
                                                           T* __restrict__ part) {
  constexpr int TB = 1024;
  __shared__ int sidx[TB];
  __shared__ T sy[TB];
  const int t = threadIdx.x;
  const long m = (long)blockIdx.x * NT + t;
  const long ib = (long)blockIdx.y * rows_per_chunk;
  const long ie = ib + rows_per_chunk < N ? ib + rows_per_chunk : N;
  T s = 0, cnt = 0;
  const int mm = (int)m;
  for (long i0 = ib; i0 < ie; i0 += TB) {
    __syncthreads();
    for (int e = t; e < TB; e += NT) {
      const long i = i0 + e;
      sidx[e] = i < ie ? (int)idx[i] : -1;
      sy[e] = i < ie ? y[i] : (T)0;
    }
    __syncthreads();
#pragma unroll 8
    for (int e = 0; e < TB; ++e) {
      const bool hit = sidx[e] == mm;
      s += hit ? sy[e] : (T)0;
      cnt += hit ? (T)1 : (T)0;
    }
  }
  if (m < M) {
    T* o = part + (long)blockIdx.y * 2 * M;
    o[m] = s;
    o[M + m] = cnt;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void cluster_reduce_kernel(const T* __restrict__ part, int nchunks, long M,
                                                            T* __restrict__ sums, T* __restrict__ counts) {
  const long m = (long)blockIdx.x * NT + threadIdx.x;
  if (m >= M) return;
  T s = 0, c = 0;
  for (int k = 0; k < nchunks; ++k) {
    s += part[(long)k * 2 * M + m];
    c += part[(long)k * 2 * M + M + m];
  }
  sums[m] = s;
  counts[m] = c;
}

template <typename T, int KIND>
int nearest_dp(mgp_handle* h, const SweepParams& prm, int D, int dist_type, const T* X, long N, const T* Z, long M,
               long* idx, T* best) {
  // rows per thread: as many as still leave two workgroups per CU (C2's 10^5 rows stay at one)
  int rpt = D <= 8 ? 4 : 2;
  while (rpt > 1 && (N + (long)NT * rpt - 1) / ((long)NT * rpt) < 2L * h->num_cus) rpt >>= 1;
  dim3 grid((unsigned)((N + (long)NT * rpt - 1) / ((long)NT * rpt)));
#define MGP_NC1(DPV, RV)                                                                                        \
  do {                                                                                                          \
    if (dist_type == 1)                                                                                         \
      hipLaunchKernelGGL((nearest_kernel<T, DPV, KIND, RV, true>), grid, dim3(NT), 0, h->stream, X, N, Z, M, D, prm, \
                         dist_type, idx, best);                                                                 \
    else                                                                                                        \
      hipLaunchKernelGGL((nearest_kernel<T, DPV, KIND, RV, false>), grid, dim3(NT), 0, h->stream, X, N, Z, M, D,  \
                         prm, dist_type, idx, best);                                                            \
  } while (0)
#define MGP_NC(DPV)          \
  do {                       \
    if (rpt == 4) {          \
      if (DPV <= 8) MGP_NC1(DPV, 4); \
    } else if (rpt == 2)     \
      MGP_NC1(DPV, 2);       \
    else                     \
      MGP_NC1(DPV, 1);       \
  } while (0)
  if (D <= 2) MGP_NC(2);
  else if (D <= 4) MGP_NC(4);
  else if (D <= 8) MGP_NC(8);
  else if (D <= 16) MGP_NC(16);
  else MGP_NC(32);
#undef MGP_NC1
#undef MGP_NC
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
int nearest_t(mgp_handle* h, const mgp_kernel* k, int dist_type, const T* X, long N, const T* Z, long M, long* idx,
              T* best) {
  SweepParams prm = mgp_make_params(k);
  if (dist_type <= 1) {  // raw inputs: no lengthscale, no profile scale
    for (int d = 0; d < MGP_FUSED_MAX_D; ++d) prm.inv_ls[d] = d < k->D ? 1.0 : 0.0;
  }
  switch (k->kind) {
    case MGP_SE: return nearest_dp<T, 0>(h, prm, k->D, dist_type, X, N, Z, M, idx, best);
    case MGP_MATERN12: return nearest_dp<T, 1>(h, prm, k->D, dist_type, X, N, Z, M, idx, best);
    case MGP_MATERN32: return nearest_dp<T, 2>(h, prm, k->D, dist_type, X, N, Z, M, idx, best);
    default: return nearest_dp<T, 3>(h, prm, k->D, dist_type, X, N, Z, M, idx, best);
  }
}

template <typename T>
int cluster_stats_t(mgp_handle* h, const long* idx, const T* y, long N, long M, T* sums, T* counts) {
  const long nblk = (M + NT - 1) / NT;
  long nchunks = (8L * h->num_cus + nblk - 1) / nblk;
  const long max_chunks = (N + 1023) / 1024;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  long rows = (N + nchunks - 1) / nchunks;
  rows = (rows + 1023) / 1024 * 1024;
  nchunks = (N + rows - 1) / rows;
  if (nchunks < 1) nchunks = 1;
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nchunks * 2 * M * sizeof(T)));
  T* part = (T*)h->ws;
  hipLaunchKernelGGL((cluster_stats_kernel<T>), dim3((unsigned)nblk, (unsigned)nchunks), dim3(NT), 0, h->stream, idx,
                     y, N, M, rows, part);
  MGP_LAUNCH_CHECK(h);
  hipLaunchKernelGGL((cluster_reduce_kernel<T>), dim3((unsigned)nblk), dim3(NT), 0, h->stream, (const T*)part,
                     (int)nchunks, M, sums, counts);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

// Segmented column sums of rows already grouped by cluster (order[] = stable sort of the rows by
// cluster, offsets[m] .. offsets[m+1] = the rows of cluster m): one wave per (cluster, column
// block), lanes stride the segment in order, fixed butterfly at the end -- deterministic, and
// N C work instead of the N M C of the transpose sweep.
template <typename T>
__global__ __launch_bounds__(256) void segment_sums_kernel(const long* __restrict__ order,
                                                           const long* __restrict__ offsets,
                                                           const T* __restrict__ Y, long C, long M,
                                                           T* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long lo = offsets[m], hi = offsets[m + 1];
  for (long c = 0; c < C; ++c) {
    T s = 0;
    for (long i = lo + lane; i < hi; i += 64) s += Y[order[i] * C + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) sums[m * C + c] = s;
  }
}

}  // namespace

extern "C" int mgp_segment_sums(mgp_handle* h, int dtype, const int64_t* order, const int64_t* offsets,
                                const void* Y, int64_t N, int64_t C, int64_t M, void* sums) {
  if (!h) return MGP_E_BADARG;
  if (dtype != MGP_F32 && dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", dtype);
  if (N < 0 || M <= 0 || C <= 0) return mgp_fail(h, MGP_E_SHAPE, "segment_sums: bad shape");
  if (!offsets || !sums || (N > 0 && (!order || !Y))) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  dim3 grid((unsigned)((M + 3) / 4));
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((segment_sums_kernel<double>), grid, dim3(256), 0, h->stream, (const long*)order,
                       (const long*)offsets, (const double*)Y, (long)C, (long)M, (double*)sums);
  else
    hipLaunchKernelGGL((segment_sums_kernel<float>), grid, dim3(256), 0, h->stream, (const long*)order,
                       (const long*)offsets, (const float*)Y, (long)C, (long)M, (float*)sums);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

extern "C" int mgp_nearest_center(mgp_handle* h, const mgp_kernel* k, int dist_type, const void* X, int64_t N,
                                  const void* Z, int64_t M, int64_t* idx, void* best) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (dist_type < 0 || dist_type > 3) return mgp_fail(h, MGP_E_BADARG, "bad dist_type %d", dist_type);
  if (N < 0 || M <= 0 || M > 2147483647L) return mgp_fail(h, MGP_E_SHAPE, "nearest_center needs N >= 0 and 0 < M < 2^31");
  if (N == 0) return MGP_OK;
  if (!X || !Z || !idx) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (k->D > MGP_FUSED_MAX_D) return mgp_nearest_generic(h, k, dist_type, X, N, Z, M, idx, best);  // generic.hip
  if (k->dtype == MGP_F64)
    return nearest_t<double>(h, k, dist_type, (const double*)X, N, (const double*)Z, M, (long*)idx, (double*)best);
  return nearest_t<float>(h, k, dist_type, (const float*)X, N, (const float*)Z, M, (long*)idx, (float*)best);
}

extern "C" int mgp_cluster_stats(mgp_handle* h, int dtype, const int64_t* idx, const void* y, int64_t N,
                                 int64_t M, void* sums, void* counts) {
  if (!h) return MGP_E_BADARG;
  if (dtype != MGP_F32 && dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", dtype);
  if (N < 0 || M <= 0 || M > 2147483647L) return mgp_fail(h, MGP_E_SHAPE, "cluster_stats: bad shape");
  if (!sums || !counts || (N > 0 && (!idx || !y))) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (N == 0) {
    MGP_HIP(h, hipMemsetAsync(sums, 0, (size_t)M * mgp_elem(dtype), h->stream));
    MGP_HIP(h, hipMemsetAsync(counts, 0, (size_t)M * mgp_elem(dtype), h->stream));
    return MGP_OK;
  }
  if (dtype == MGP_F64)
    return cluster_stats_t<double>(h, (const long*)idx, (const double*)y, N, M, (double*)sums, (double*)counts);
  return cluster_stats_t<float>(h, (const long*)idx, (const float*)y, N, M, (float*)sums, (float*)counts);
}
