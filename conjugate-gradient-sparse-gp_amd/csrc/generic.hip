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


// ------------------------------------------------------------------ rows F1 / F2 at any D (<= MGP_MAX_D)
// The reference's assignment (`cggp/optimize.py:41-98`, `selection.py:14-32`) and training step
// (`optimize.py:198-254`) are dimension-free; its data sets reach D = 77 (buzz) and D = 90 (song)
// (`cggp/cli_utils.py:72-86`).  The fused kernels of cluster.hip / grad.hip keep a point's coordinates in
// registers and stop at D = 32; above that the same quantities are formed tile by tile with the input
// dimensions staged through LDS 16 at a time, as k_dense_generic_kernel does.

// nrm[i] = sum_d (x_id * sc_d)^2, d ascending
template <typename T>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const T* __restrict__ X, long n, int D,
                                                         const double* __restrict__ inv_ls, T* __restrict__ nrm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T s = 0;
  for (int d = 0; d < D; ++d) {
    const T v = X[i * D + d] * (T)inv_ls[d];
    s = mgp_fma(v, v, s);
  }
  nrm[i] = s;
}

// argmin_j of the (scaled) squared distance for the 64 rows of a workgroup, all M centres streamed in tiles
// of 64.  Thread (ty, tx) holds rows 4 ty .. 4 ty + 3 and, of every tile, columns tx, tx + 16, tx + 32,
// tx + 48: its columns ascend over the whole sweep, so a strict '<' keeps the FIRST index among its own
// candidates; the 16 threads of a row group are then merged by (value, index) -- first index on ties, as
// numpy / tf argmin (cggp/selection.py:28, optimize.py:51).  DIRECT (the reference's `euclid_distance`,
// distance.py:9-11): sum of squared differences; otherwise GPflow's expansion (-2 a.b + |b|^2) + |a|^2.
template <typename T, int KIND, bool DIRECT>
__global__ __launch_bounds__(256) void nearest_generic_kernel(const T* __restrict__ X, long N, const T* __restrict__ Z,
                                                              long M, int D, const double* __restrict__ inv_ls,
                                                              const T* __restrict__ xn, const T* __restrict__ zn,
                                                              T variance, T clamp, int dist_type,
                                                              long* __restrict__ idx, T* __restrict__ best) {
  __shared__ T As[GT][GK + 1];
  __shared__ T Bs[GT][GK + 1];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const long i0 = (long)blockIdx.x * GT;
  T bs[4], a2[4];
  long bj[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bs[a] = (T)INFINITY;
    bj[a] = 0;
    const long i = i0 + ty * 4 + a;
    a2[a] = (!DIRECT && i < N) ? xn[i] : (T)0;
  }
  for (long j0 = 0; j0 < M; j0 += GT) {
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
        As[r][d] = (i0 + r < N && d0 + d < D) ? X[(i0 + r) * D + d0 + d] * sc : (T)0;
        Bs[r][d] = (j0 + r < M && d0 + d < D) ? Z[(j0 + r) * D + d0 + d] * sc : (T)0;
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
            if (DIRECT) {
              const T df = av[a] - bv[b];
              acc[a][b] = mgp_fma(df, df, acc[a][b]);
            } else {
              acc[a][b] = mgp_fma(av[a], bv[b], acc[a][b]);
            }
          }
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long j = j0 + tx + 16 * b;
      if (j >= M) continue;
      const T b2 = DIRECT ? (T)0 : zn[j];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const T s = DIRECT ? acc[a][b] : (mgp_fma((T)-2, acc[a][b], b2) + a2[a]);
        if (s < bs[a]) {  // strict: first index on ties while j ascends
          bs[a] = s;
          bj[a] = j;
        }
      }
    }
  }
  // merge the 16 column groups of a row (lanes tx = 0..15 of the same ty are consecutive lanes of one wave)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const T os = __shfl_xor(bs[a], off, 64);
      const long oj = __shfl_xor(bj[a], off, 64);
      if (os < bs[a] || (os == bs[a] && oj < bj[a])) {
        bs[a] = os;
        bj[a] = oj;
      }
    }
  }
  if (tx == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long i = i0 + ty * 4 + a;
      if (i >= N) continue;
      idx[i] = bj[a];
      if (best != nullptr) {
        T o;
        if (dist_type == 0) {
          o = bs[a];
        } else if (dist_type == 1) {
          o = mgp_sqrt(bs[a] > 0 ? bs[a] : (T)0);
        } else {
          const T rho = mgp_profile<KIND, T>(-bs[a], clamp);  // k / variance
          o = dist_type == 2 ? (T)2 * variance * ((T)1 - rho) : (T)1 - rho;
        }
        best[i] = o;
      }
    }
  }
}

// f = k/variance and f' = df/dr2 of the profile as functions of the PLAIN scaled squared distance
// r2 = sum_d ((a_d - b_d) / l_d)^2 (grad.hip's formulas; GPflow's 1e-36 floor under the root)
template <typename T, int KIND>
__device__ __forceinline__ void profile_and_slope(T r2, T& f, T& fp) {
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
}

// Kernel-block VJP at any D: part[blk][d] = sum over the block's pairs of G_ij f'(r2_ij) ((a_id - b_jd)/l_d)^2,
// part[blk][D] = sum G_ij f_ij.  A workgroup takes one 64-column strip of a row range, tile by tile: first
// pass over the dimensions -> r2 of its 4 x 4 pairs per thread (direct differences, as grad.hip), then
// H = G f'; second pass over the dimensions -> the per-dimension sums, reduced over the workgroup stage by
// stage in a fixed order (wave butterflies, then waves 0..3) and accumulated in LDS.  Nothing of size
// na x nb x D is materialised; per-block partials are added on the host in block order.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_dense_vjp_generic_kernel(const T* __restrict__ A, long na,
                                                                  const T* __restrict__ B, long nb,
                                                                  const T* __restrict__ G, long ldg, int D,
                                                                  const double* __restrict__ inv_ls,
                                                                  long rows_per_block, double* __restrict__ part) {
  __shared__ T As[GT][GK + 1];
  __shared__ T Bs[GT][GK + 1];
  __shared__ double red[4][GK + 1];
  extern __shared__ double tot[];  // [D + 1]
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, lane = t & 63, wave = t >> 6;
  const long j0 = (long)blockIdx.x * GT;
  const long ib = (long)blockIdx.y * rows_per_block;
  const long ie = ib + rows_per_block < na ? ib + rows_per_block : na;
  for (int d = t; d <= D; d += 256) tot[d] = 0.0;
  double gf = 0.0;
  for (long i0 = ib; i0 < ie; i0 += GT) {
    T r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) r2[a][b] = 0;
    for (int d0 = 0; d0 < D; d0 += GK) {
      __syncthreads();
      for (int e = t; e < GT * GK; e += 256) {
        const int r = e / GK, d = e % GK;
        const T sc = d0 + d < D ? (T)inv_ls[d0 + d] : (T)0;
        As[r][d] = (i0 + r < ie && d0 + d < D) ? A[(i0 + r) * D + d0 + d] * sc : (T)0;
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
            r2[a][b] = mgp_fma(df, df, r2[a][b]);
          }
      }
    }
    T H[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long i = i0 + ty * 4 + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const long j = j0 + tx + 16 * b;
        T g = 0;
        if (i < ie && j < nb) g = G[i * ldg + j];
        T f, fp;
        profile_and_slope<T, KIND>(r2[a][b], f, fp);
        gf += (double)(g * f);
        H[a][b] = g * fp;  // 0 outside the block: padded pairs contribute nothing
      }
    }
    for (int d0 = 0; d0 < D; d0 += GK) {
      __syncthreads();
      for (int e = t; e < GT * GK; e += 256) {
        const int r = e / GK, d = e % GK;
        const T sc = d0 + d < D ? (T)inv_ls[d0 + d] : (T)0;
        As[r][d] = (i0 + r < ie && d0 + d < D) ? A[(i0 + r) * D + d0 + d] * sc : (T)0;
        Bs[r][d] = (j0 + r < nb && d0 + d < D) ? B[(j0 + r) * D + d0 + d] * sc : (T)0;
      }
      __syncthreads();
      double sd[GK];
#pragma unroll
      for (int d = 0; d < GK; ++d) {
        T av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = As[ty * 4 + a][d];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = Bs[tx + 16 * b][d];
        T s = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const T df = av[a] - bv[b];
            s = mgp_fma(H[a][b], df * df, s);
          }
        sd[d] = (double)s;
      }
#pragma unroll
      for (int d = 0; d < GK; ++d) {
        double v = sd[d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][d] = v;
      }
      __syncthreads();
      if (t < GK && d0 + t < D) tot[d0 + t] += (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
  }
  {
    double v = gf;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave][GK] = v;
    __syncthreads();
    if (t == 0) tot[D] = (red[0][GK] + red[1][GK]) + (red[2][GK] + red[3][GK]);
    __syncthreads();
  }
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  for (int d = t; d <= D; d += 256) part[blk * (D + 1) + d] = tot[d];
}

// k^2 column sums over explicit panels: part[c][m] = sum of panel[i][m]^2 over the rows of sub-chunk c
// (256 rows, i ascending), then acc[m] += part[0][m] + part[1][m] + ... in order -- deterministic
template <typename T>
__global__ __launch_bounds__(256) void colsq_partial_kernel(const T* __restrict__ panel, long rc, long M,
                                                            T* __restrict__ part) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const long ib = (long)blockIdx.y * 256, ie = ib + 256 < rc ? ib + 256 : rc;
  T s = 0;
  for (long i = ib; i < ie; ++i) {
    const T v = panel[i * M + m];
    s = mgp_fma(v, v, s);
  }
  part[(long)blockIdx.y * M + m] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void colsq_accumulate_kernel(const T* __restrict__ part, long nsub, long M,
                                                               T* __restrict__ out, int first) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  T s = first ? (T)0 : out[m];
  for (long c = 0; c < nsub; ++c) s += part[c * M + m];
  out[m] = s;
}

int upload_raw_scales(mgp_handle* h, int D, const double* host) {
  MGP_HIP(h, hipMemcpyAsync(h->dparams, host, (size_t)D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  return MGP_OK;
}

template <typename T>
int nearest_generic_t(mgp_handle* h, const mgp_kernel* k, int dist_type, const T* X, long N, const T* Z, long M,
                      long* idx, T* best) {
  const double c = mgp_profile_scale(k->kind);
  if (dist_type <= 1) {  // raw inputs: no lengthscale, no profile scale (as cluster.hip)
    double ones[MGP_MAX_D];
    for (int d = 0; d < k->D; ++d) ones[d] = 1.0;
    MGP_TRY(upload_raw_scales(h, k->D, ones));
  } else {
    MGP_TRY(upload_scales(h, k));
  }
  const bool direct = dist_type == 1;
  T *xn = nullptr, *zn = nullptr;
  if (!direct) {
    MGP_TRY(mgp_reserve(h, &h->gen, &h->gen_bytes, (size_t)(N + M) * sizeof(T) + 256));
    xn = (T*)h->gen;
    zn = xn + N;
    hipLaunchKernelGGL((row_sqnorm_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, X, N, k->D,
                       (const double*)h->dparams, xn);
    MGP_LAUNCH_CHECK(h);
    hipLaunchKernelGGL((row_sqnorm_kernel<T>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, Z, M, k->D,
                       (const double*)h->dparams, zn);
    MGP_LAUNCH_CHECK(h);
  }
  dim3 grid((unsigned)((N + GT - 1) / GT));
#define MGP_NG(KV, DV)                                                                                            \
  hipLaunchKernelGGL((nearest_generic_kernel<T, KV, DV>), grid, dim3(256), 0, h->stream, X, N, Z, M, k->D,          \
                     (const double*)h->dparams, (const T*)xn, (const T*)zn, (T)k->variance, (T)(c * c * 1e-36),   \
                     dist_type, idx, best)
#define MGP_NGK(KV)          \
  do {                       \
    if (direct) MGP_NG(KV, true); \
    else MGP_NG(KV, false);  \
  } while (0)
  switch (k->kind) {
    case MGP_SE: MGP_NGK(0); break;
    case MGP_MATERN12: MGP_NGK(1); break;
    case MGP_MATERN32: MGP_NGK(2); break;
    default: MGP_NGK(3); break;
  }
#undef MGP_NGK
#undef MGP_NG
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
int vjp_generic_t(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* G, long ldg,
                  double* dvar, double* dls) {
  const int D = k->D;
  double inv[MGP_MAX_D];
  for (int d = 0; d < D; ++d) inv[d] = 1.0 / k->lengthscales[d];  // plain reciprocals, as grad.hip
  MGP_TRY(upload_raw_scales(h, D, inv));
  const long nbx = (nb + GT - 1) / GT;
  long nby = (4L * h->num_cus + nbx - 1) / nbx;
  const long max_y = (na + GT - 1) / GT;
  if (nby > max_y) nby = max_y;
  if (nby < 1) nby = 1;
  long rows = (na + nby - 1) / nby;
  rows = (rows + GT - 1) / GT * GT;
  nby = (na + rows - 1) / rows;
  const long nblocks = nbx * nby;
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nblocks * (D + 1) * sizeof(double)));
  double* part = (double*)h->ws;
  dim3 grid((unsigned)nbx, (unsigned)nby);
  const size_t dyn = (size_t)(D + 1) * sizeof(double);
#define MGP_VG(KV)                                                                                                   \
  hipLaunchKernelGGL((k_dense_vjp_generic_kernel<T, KV>), grid, dim3(256), dyn, h->stream, A, na, B, nb, G, ldg, D, \
                     (const double*)h->dparams, rows, part)
  switch (k->kind) {
    case MGP_SE: MGP_VG(0); break;
    case MGP_MATERN12: MGP_VG(1); break;
    case MGP_MATERN32: MGP_VG(2); break;
    default: MGP_VG(3); break;
  }
#undef MGP_VG
  MGP_LAUNCH_CHECK(h);
  std::vector<double> host((size_t)nblocks * (D + 1));
  MGP_HIP(h, hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> tot((size_t)D + 1, 0.0);
  for (long bI = 0; bI < nblocks; ++bI)
    for (int d = 0; d <= D; ++d) tot[(size_t)d] += host[(size_t)bI * (D + 1) + d];
  *dvar = tot[(size_t)D];  // sum G f  == sum G k / variance
  for (int d = 0; d < D; ++d) dls[d] = k->variance * (-2.0 / k->lengthscales[d]) * tot[(size_t)d];
  return MGP_OK;
}

template <typename T>
int sq_colsum_generic_t(mgp_handle* h, const mgp_kernel* k, const T* X, long N, const T* Z, long M, T* out) {
  // row chunks of X: panel [rc, M] <= 256 MB
  long rc_max = (long)((256ull << 20) / ((size_t)M * sizeof(T)));
  if (rc_max > N) rc_max = N;
  if (rc_max < 256) rc_max = 256;
  rc_max = rc_max / 256 * 256;
  const long nsub_max = rc_max / 256;
  const size_t panel_elems = (size_t)rc_max * M, part_elems = (size_t)nsub_max * M;
  MGP_TRY(mgp_reserve(h, &h->gen, &h->gen_bytes, (panel_elems + part_elems) * sizeof(T) + 256));
  T* panel = (T*)h->gen;
  T* part = panel + panel_elems;
  const dim3 gm((unsigned)((M + 255) / 256));
  for (long i0 = 0; i0 < N; i0 += rc_max) {
    const long rc = N - i0 < rc_max ? N - i0 : rc_max;
    const long nsub = (rc + 255) / 256;
    MGP_TRY(k_dense_generic_t<T>(h, k, X + i0 * k->D, rc, Z, M, panel, M, 0.0, nullptr, nullptr));
    hipLaunchKernelGGL((colsq_partial_kernel<T>), dim3(gm.x, (unsigned)nsub), dim3(256), 0, h->stream, (const T*)panel,
                       rc, M, part);
    MGP_LAUNCH_CHECK(h);
    hipLaunchKernelGGL((colsq_accumulate_kernel<T>), gm, dim3(256), 0, h->stream, (const T*)part, nsub, M, out,
                       i0 == 0 ? 1 : 0);
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

int mgp_nearest_generic(mgp_handle* h, const mgp_kernel* k, int dist_type, const void* X, int64_t N, const void* Z,
                        int64_t M, int64_t* idx, void* best) {
  if (k->dtype == MGP_F64)
    return nearest_generic_t<double>(h, k, dist_type, (const double*)X, N, (const double*)Z, M, (long*)idx,
                                     (double*)best);
  return nearest_generic_t<float>(h, k, dist_type, (const float*)X, N, (const float*)Z, M, (long*)idx, (float*)best);
}

int mgp_k_dense_vjp_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                            const void* G, int64_t ldg, double* dvariance, double* dlengthscales) {
  if (k->dtype == MGP_F64)
    return vjp_generic_t<double>(h, k, (const double*)A, na, (const double*)B, nb, (const double*)G, ldg, dvariance,
                                 dlengthscales);
  return vjp_generic_t<float>(h, k, (const float*)A, na, (const float*)B, nb, (const float*)G, ldg, dvariance,
                              dlengthscales);
}

int mgp_kmn_sq_colsum_generic(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z, int64_t M,
                              void* out) {
  MGP_TRY(upload_scales(h, k));
  if (k->dtype == MGP_F64)
    return sq_colsum_generic_t<double>(h, k, (const double*)X, N, (const double*)Z, M, (double*)out);
  return sq_colsum_generic_t<float>(h, k, (const float*)X, N, (const float*)Z, M, (float*)out);
}
