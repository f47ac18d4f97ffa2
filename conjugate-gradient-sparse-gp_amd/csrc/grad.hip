// grad.hip -- next row F2: hyper-parameter gradients of a kernel block.
//
// Given G = dL/dK for K = k(A, B) [na, nb] (row-major, leading dimension ldg), the training step
// of the reference differentiates through gpflow's kernel (`cggp/models.py:125-134,293-354` under
// tf.GradientTape, `cggp/optimize.py:198-254`).  The vector-Jacobian product needs only
//     dL/dvariance   = sum_ij G_ij k_ij / variance
//     dL/dl_d        = sum_ij G_ij * variance * f'(r2_ij) * (-2 (a_id - b_jd)^2 / l_d^3)
// with f = k/variance as a function of the scaled squared distance r2 -- a fused N x M reduction,
// no dK/dtheta is materialised.  f'(r2): SE -f/2; Matern12 -e^{-r}/(2r) (0 below GPflow's 1e-36
// floor, where max() picks the constant); Matern32 -(3/2) e^{-sqrt3 r}; Matern52
// -(5/6)(1 + sqrt5 r) e^{-sqrt5 r}.  Direct differences are used for r2 (this is a gradient, the
// expansion's cancellation error would be amplified).  Per-block partial sums are written out and
// added on the host in block order: deterministic.
#include <vector>

#include "mgp_common.h"

namespace {

template <typename T, int DP, int KIND>
__global__ __launch_bounds__(256) void k_dense_vjp_kernel(const T* __restrict__ A, long na,
                                                          const T* __restrict__ B, long nb,
                                                          const T* __restrict__ G, long ldg, int D,
                                                          SweepParams prm, long rows_per_block,
                                                          double* __restrict__ part) {
  constexpr int TA = 32;
  __shared__ T tile[TA * DP];
  __shared__ double red[4][DP + 1];
  const int t = threadIdx.x;
  const long j = (long)blockIdx.x * 256 + t;
  const long ib = (long)blockIdx.y * rows_per_block;
  const long ie = ib + rows_per_block < na ? ib + rows_per_block : na;
  // inv_ls here is 1/l_d (no profile scale): the caller passes plain reciprocals
  T b[DP];
  {
    const long jc = j < nb ? j : nb - 1;
#pragma unroll
    for (int d = 0; d < DP; ++d) b[d] = d < D ? B[jc * D + d] * (T)prm.inv_ls[d] : (T)0;
  }
  double acc[DP + 1];
#pragma unroll
  for (int d = 0; d <= DP; ++d) acc[d] = 0.0;
  for (long i0 = ib; i0 < ie; i0 += TA) {
    __syncthreads();
    for (int e = t; e < TA * DP; e += 256) {
      const long i = i0 + e / DP;
      const int d = e % DP;
      tile[e] = (i < ie && d < D) ? A[i * D + d] * (T)prm.inv_ls[d] : (T)0;
    }
    __syncthreads();
    if (j < nb) {
      const int lim = (ie - i0) < TA ? (int)(ie - i0) : TA;
      for (int ii = 0; ii < lim; ++ii) {
        T d2[DP];
        T r2 = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          const T df = tile[ii * DP + d] - b[d];
          d2[d] = df * df;
          r2 += d2[d];
        }
        const T g = G[(i0 + ii) * ldg + j];
        T f, fp;  // f = k/variance, fp = df/dr2
        if (KIND == 0) {
          f = mgp_exp2((T)(-0.5 * MGP_LOG2E) * r2);
          fp = (T)-0.5 * f;
        } else {
          const bool floor_hit = !(r2 > (T)1e-36);
          const T r = mgp_sqrt(floor_hit ? (T)1e-36 : r2);
          if (KIND == 1) {
            f = mgp_exp2((T)(-MGP_LOG2E) * r);
            fp = floor_hit ? (T)0 : -f / ((T)2 * r);
          } else if (KIND == 2) {
            const T s3 = (T)1.7320508075688772935;
            const T e = mgp_exp2((T)(-MGP_LOG2E) * s3 * r);
            f = ((T)1 + s3 * r) * e;
            fp = floor_hit ? (T)0 : (T)-1.5 * e;
          } else {
            const T s5 = (T)2.2360679774997896964;
            const T e = mgp_exp2((T)(-MGP_LOG2E) * s5 * r);
            f = ((T)1 + s5 * r + (T)(5.0 / 3.0) * r2) * e;
            fp = floor_hit ? (T)0 : (T)(-5.0 / 6.0) * ((T)1 + s5 * r) * e;
          }
        }
        acc[DP] += (double)(g * f);
        const T gfp = g * fp;
#pragma unroll
        for (int d = 0; d < DP; ++d) acc[d] += (double)(gfp * d2[d]);
      }
    }
  }
  // block reduction of DP+1 sums
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int d = 0; d <= DP; ++d) {
    double v = acc[d];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][d] = v;
  }
  __syncthreads();
  if (t <= DP) {
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    part[blk * (DP + 1) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}

template <typename T, int KIND>
int vjp_dp(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* G, long ldg,
           double* dvar, double* dls) {
  SweepParams prm = mgp_make_params(k);
  for (int d = 0; d < MGP_FUSED_MAX_D; ++d) prm.inv_ls[d] = d < k->D ? 1.0 / k->lengthscales[d] : 0.0;
  const int D = k->D;
  const int DPv = D <= 2 ? 2 : (D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : 32)));
  const long nbx = (nb + 255) / 256;
  long nby = (4L * h->num_cus + nbx - 1) / nbx;
  const long max_y = (na + 31) / 32;
  if (nby > max_y) nby = max_y;
  if (nby < 1) nby = 1;
  long rows = (na + nby - 1) / nby;
  rows = (rows + 31) / 32 * 32;
  nby = (na + rows - 1) / rows;
  const long nblocks = nbx * nby;
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nblocks * (DPv + 1) * sizeof(double)));
  double* part = (double*)h->ws;
  dim3 grid((unsigned)nbx, (unsigned)nby);
#define MGP_VJP(DPV) \
  hipLaunchKernelGGL((k_dense_vjp_kernel<T, DPV, KIND>), grid, dim3(256), 0, h->stream, A, na, B, nb, G, ldg, D, prm, \
                     rows, part)
  switch (DPv) {
    case 2: MGP_VJP(2); break;
    case 4: MGP_VJP(4); break;
    case 8: MGP_VJP(8); break;
    case 16: MGP_VJP(16); break;
    default: MGP_VJP(32); break;
  }
#undef MGP_VJP
  MGP_LAUNCH_CHECK(h);
  std::vector<double> host((size_t)nblocks * (DPv + 1));
  MGP_HIP(h, hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> tot(DPv + 1, 0.0);
  for (long bI = 0; bI < nblocks; ++bI)
    for (int d = 0; d <= DPv; ++d) tot[d] += host[(size_t)bI * (DPv + 1) + d];
  *dvar = tot[DPv];  // sum G f  == sum G k / variance
  for (int d = 0; d < D; ++d) dls[d] = k->variance * (-2.0 / k->lengthscales[d]) * tot[d];
  return MGP_OK;
}

template <typename T>
int vjp_t(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* G, long ldg,
          double* dvar, double* dls) {
  switch (k->kind) {
    case MGP_SE: return vjp_dp<T, 0>(h, k, A, na, B, nb, G, ldg, dvar, dls);
    case MGP_MATERN12: return vjp_dp<T, 1>(h, k, A, na, B, nb, G, ldg, dvar, dls);
    case MGP_MATERN32: return vjp_dp<T, 2>(h, k, A, na, B, nb, G, ldg, dvar, dls);
    default: return vjp_dp<T, 3>(h, k, A, na, B, nb, G, ldg, dvar, dls);
  }
}

}  // namespace

extern "C" int mgp_k_dense_vjp(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B,
                               int64_t nb, const void* G, int64_t ldg, double* dvariance, double* dlengthscales) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (!dvariance || !dlengthscales) return mgp_fail(h, MGP_E_BADARG, "NULL output");
  *dvariance = 0.0;
  for (int d = 0; d < k->D; ++d) dlengthscales[d] = 0.0;
  if (na < 0 || nb < 0 || ldg < nb) return mgp_fail(h, MGP_E_SHAPE, "k_dense_vjp: bad shape");
  if (na == 0 || nb == 0) return MGP_OK;
  if (!A || !B || !G) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (k->D > MGP_FUSED_MAX_D)  // generic.hip: dimensions staged through LDS
    return mgp_k_dense_vjp_generic(h, k, A, na, B, nb, G, ldg, dvariance, dlengthscales);
  if (k->dtype == MGP_F64)
    return vjp_t<double>(h, k, (const double*)A, na, (const double*)B, nb, (const double*)G, ldg, dvariance,
                         dlengthscales);
  return vjp_t<float>(h, k, (const float*)A, na, (const float*)B, nb, (const float*)G, ldg, dvariance,
                      dlengthscales);
}
