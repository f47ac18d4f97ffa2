// generic.hip -- any input dimension D <= MGP_MAX_D (the fused sweeps keep a point's coordinates in
// registers and stop at D = 32).  This is the reference's dense form on the GPU: explicit kernel
// panels (`gpflow` Kuf per batch, cggp/models.py:334) and a GEMM against them, chunked so that no
// panel exceeds ~256 MB.  Used for high-dimensional inputs (UCI sets with D up to a few hundred);
// correctness path, not a tuned one.
#include "mgp_common.h"

namespace {

constexpr int GT = 64;  // output tile edge
constexpr int GK = 16;  // input dimensions per LDS stage

// out[i, j] = variance * f(|a_i - b_j|^2) (+ diag terms); direct differences, dims staged 16 at a time
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_dense_generic_kernel(const T* __restrict__ A, long na,
                                                              const T* __restrict__ B, long nb, T* __restrict__ out,
                                                              long ld, int D, const double* __restrict__ inv_ls,
                                                              T variance, T clamp, T jitter,
                                                              const T* __restrict__ diag_add,
                                                              const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  __shared__ T As[GT][GK + 1];
  __shared__ T Bs[GT][GK + 1];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const long i0 = (long)blockIdx.y * GT, j0 = (long)blockIdx.x * GT;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0;
  for (int d0 = 0; d0 < D; d0 += GK) {
    __syncthreads();
    for (int e = t; e < GT * GK; e += 256) {
      const int r = e / GK, d = e % GK;
      const T sc = d0 + d < D ? (T)inv_ls[d0 + d] : (T)0;
      As[r][d] = (i0 + r < na && d0 + d < D) ? A[(i0 + r) * D + d0 + d] * sc : (T)0;
      Bs[r][d] = (j0 + r < nb && d0 + d < D) ? B[(j0 + r) * D + d0 + d] * sc : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < GK; ++d) {
      T av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[ty * 4 + a][d];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[tx + 16 * b][d];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const T df = av[a] - bv[b];
          acc[a][b] = mgp_fma(df, df, acc[a][b]);
        }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long i = i0 + ty * 4 + a;
    if (i >= na) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long j = j0 + tx + 16 * b;
      if (j >= nb) continue;
      T v = variance * mgp_profile<KIND, T>(-acc[a][b], clamp);
      if (i == j) {
        v += jitter;
        if (diag_add != nullptr) v += diag_add[i];
      }
      out[i * ld + j] = v;
    }
  }
}

// Wt[r, j] = W(j, r)  (contiguous [R, nb] from a strided view)
template <typename T>
__global__ __launch_bounds__(256) void gather_view_kernel(const T* __restrict__ W, long sj, long sr, long nb, int R,
                                                          T* __restrict__ Wt, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nb * R) return;
  const long r = e / nb, j = e - r * nb;
  Wt[e] = W[j * sj + r * sr];
}

// out(i0 + i, r) = src[r, i] (+ alpha * addend)
template <typename T>
__global__ __launch_bounds__(256) void scatter_view_kernel(const T* __restrict__ src, long rc, int R, long i0,
                                                           T* __restrict__ out, long o_si, long o_sr, T alpha,
                                                           const T* __restrict__ addend, long ad_si, long ad_sr,
                                                           const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rc * R) return;
  const long r = e / rc, i = e - r * rc;
  T v = src[e];
  if (addend != nullptr) v = mgp_fma(alpha, addend[(i0 + i) * ad_si + r * ad_sr], v);
  out[(i0 + i) * o_si + r * o_sr] = v;
}

int upload_scales(mgp_handle* h, const mgp_kernel* k) {
  double host[MGP_MAX_D];
  const double c = mgp_profile_scale(k->kind);
  for (int d = 0; d < k->D; ++d) host[d] = c / k->lengthscales[d];
  MGP_HIP(h, hipMemcpyAsync(h->dparams, host, (size_t)k->D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));  // host[] is a stack temporary
  return MGP_OK;
}

template <typename T>
int k_dense_generic_t(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, T* out, long ld,
                      double jitter, const T* diag_add, const int* gate) {
  const double c = mgp_profile_scale(k->kind);
  dim3 grid((unsigned)((nb + GT - 1) / GT), (unsigned)((na + GT - 1) / GT));
#define MGP_KG(KV)                                                                                              \
  hipLaunchKernelGGL((k_dense_generic_kernel<T, KV>), grid, dim3(256), 0, h->stream, A, na, B, nb, out, ld, k->D, \
                     (const double*)h->dparams, (T)k->variance, (T)(c * c * 1e-36), (T)jitter, diag_add, gate)
  switch (k->kind) {
    case MGP_SE: MGP_KG(0); break;
    case MGP_MATERN12: MGP_KG(1); break;
    case MGP_MATERN32: MGP_KG(2); break;
    default: MGP_KG(3); break;
  }
#undef MGP_KG
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
int sweep_generic_t(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* W,
                    long w_sj, long w_sr, int R, T* out, long o_si, long o_sr, T alpha, const T* addend, long ad_si,
                    long ad_sr, const int* gate) {
  // owned chunks x streamed chunks; panel [rc, sc] <= 256 MB
  const long sc_max = nb < 16384 ? nb : 16384;
  long rc_max = (long)((256ull << 20) / ((size_t)sc_max * sizeof(T)));
  if (rc_max > na) rc_max = na;
  if (rc_max < 64) rc_max = 64;
  const size_t wt_elems = (size_t)R * nb, panel_elems = (size_t)rc_max * sc_max, oc_elems = (size_t)R * rc_max;
  MGP_TRY(mgp_reserve(h, &h->gen, &h->gen_bytes, (wt_elems + panel_elems + oc_elems) * sizeof(T) + 256));
  T* Wt = (T*)h->gen;
  T* panel = Wt + wt_elems;
  T* oc = panel + panel_elems;
  hipLaunchKernelGGL((gather_view_kernel<T>), dim3((unsigned)((wt_elems + 255) / 256)), dim3(256), 0, h->stream, W,
                     w_sj, w_sr, nb, R, Wt, gate);
  MGP_LAUNCH_CHECK(h);
  for (long i0 = 0; i0 < na; i0 += rc_max) {
    const long rc = na - i0 < rc_max ? na - i0 : rc_max;
    for (long j0 = 0; j0 < nb; j0 += sc_max) {
      const long sc = nb - j0 < sc_max ? nb - j0 : sc_max;
      MGP_TRY(k_dense_generic_t<T>(h, k, A + i0 * k->D, rc, B + j0 * k->D, sc, panel, sc, 0.0, nullptr, gate));
      // oc[R, rc] (+)= Wt[R, j0:j0+sc] . panel[rc, sc]^T
      MGP_TRY(mgp_gemm_nt(h, k->dtype, Wt + j0, nb, R, panel, sc, rc, sc, oc, rc, j0 > 0 ? 1 : 0, gate));
    }
    hipLaunchKernelGGL((scatter_view_kernel<T>), dim3((unsigned)((rc * R + 255) / 256)), dim3(256), 0, h->stream,
                       (const T*)oc, rc, R, i0, out, o_si, o_sr, alpha, addend, ad_si, ad_sr, gate);
    MGP_LAUNCH_CHECK(h);
  }
  return MGP_OK;
}

}  // namespace

int mgp_k_dense_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                        void* out, int64_t ld, double jitter, const void* diag_add, const int* gate) {
  MGP_TRY(upload_scales(h, k));
  if (k->dtype == MGP_F64)
    return k_dense_generic_t<double>(h, k, (const double*)A, na, (const double*)B, nb, (double*)out, ld, jitter,
                                     (const double*)diag_add, gate);
  return k_dense_generic_t<float>(h, k, (const float*)A, na, (const float*)B, nb, (float*)out, ld, jitter,
                                  (const float*)diag_add, gate);
}

int mgp_sweep_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                      VecView W, int32_t R, VecViewMut out, double alpha, VecView addend, const int* gate) {
  MGP_TRY(upload_scales(h, k));
  if (k->dtype == MGP_F64)
    return sweep_generic_t<double>(h, k, (const double*)A, na, (const double*)B, nb, (const double*)W.base, W.si,
                                   W.sr, R, (double*)out.base, out.si, out.sr, alpha, (const double*)addend.base,
                                   addend.si, addend.sr, gate);
  return sweep_generic_t<float>(h, k, (const float*)A, na, (const float*)B, nb, (const float*)W.base, W.si, W.sr, R,
                                (float*)out.base, out.si, out.sr, (float)alpha, (const float*)addend.base, addend.si,
                                addend.sr, gate);
}
