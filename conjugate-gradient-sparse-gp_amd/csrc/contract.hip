// contract.hip -- explicit K_mn K_nm = k(Z,X) k(X,Z) on the fp64/fp32 matrix cores (row S1).
//
// GPflow's SGPR forms A A^T with A = L^-1 K_mn / sigma (an O(N M^2) GEMM on a materialised
// [M,N] matrix).  Here K is never materialised: each workgroup owns one 128x128 tile of the
// [M,M] output (upper-triangular tiles only, the result is symmetric), walks its share of the
// rows of X in steps of 16, evaluates the two 16x128 kernel panels straight into LDS (VALU) and
// contracts them with v_mfma_f64_16x16x4_f64.  Rows are split over blockIdx.z; the per-split
// partial tiles are summed in fixed order by a second kernel, which also mirrors the lower
// triangle (deterministic, no float atomics).  2 N M^2 flop on the matrix cores (half of it
// skipped by symmetry) next to N M (D + profile) VALU work per tile column.
#include <cstdlib>

#include "mgp_common.h"

namespace {

template <typename T>
struct MfmaT;
template <>
struct MfmaT<double> {
  using Acc = __attribute__((ext_vector_type(4))) double;
  static __device__ __forceinline__ Acc run(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct MfmaT<float> {
  using Acc = __attribute__((ext_vector_type(4))) float;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int CT = 128;  // output tile edge
constexpr int CK = 16;   // rows of X per step
constexpr int CS = CK + 2;

template <typename T, int DP, int KIND>
__global__ __launch_bounds__(256) void kmn_knm_kernel(const T* __restrict__ X, long N, const T* __restrict__ Z,
                                                      long M, T* __restrict__ part, long rows_per_split, int D,
                                                      SweepParams prm, const int* __restrict__ tile_ab) {
  __shared__ __attribute__((aligned(16))) T Ka[CT * CS];  // [a][k]
  __shared__ __attribute__((aligned(16))) T Kb[CT * CS];  // [b][k]
  __shared__ __attribute__((aligned(16))) T Xs[CK * (DP + 2)];
  using Acc = typename MfmaT<T>::Acc;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ta = tile_ab[2 * blockIdx.x], tb = tile_ab[2 * blockIdx.x + 1];
  const bool diag = ta == tb;
  const long a0 = (long)ta * CT, b0 = (long)tb * CT;
  const long i_begin = (long)blockIdx.z * rows_per_split;
  const long i_end = i_begin + rows_per_split < N ? i_begin + rows_per_split : N;

  // this thread's inducing points for the two panels (column c = t & 127, row half = t >> 7)
  const int c = t & 127, rh = t >> 7;
  T za[DP], zb[DP];
  T za2 = 0, zb2 = 0;
  {
    const long ja = a0 + c < M ? a0 + c : M - 1, jb = b0 + c < M ? b0 + c : M - 1;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      T va = d < D ? Z[ja * D + d] * (T)prm.inv_ls[d] : (T)0;
      T vb = d < D ? Z[jb * D + d] * (T)prm.inv_ls[d] : (T)0;
      za[d] = va;
      zb[d] = vb;
      za2 = mgp_fma(va, va, za2);
      zb2 = mgp_fma(vb, vb, zb2);
    }
  }
  const bool a_ok = a0 + c < M, b_ok = b0 + c < M;
  const T clamp = (T)prm.clamp;

  Acc acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[m][q] = Acc{0, 0, 0, 0};

  for (long i0 = i_begin; i0 < i_end; i0 += CK) {
    __syncthreads();
    if (t < CK) {  // stage 16 rows of X: doubled scaled coords + negative squared norm
      const long i = i0 + t;
      T* p = &Xs[t * (DP + 2)];
      T s = 0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        T v = (d < D && i < i_end) ? X[i * D + d] * (T)prm.inv_ls[d] : (T)0;
        s = mgp_fma(v, v, s);
        p[d] = v + v;
      }
      p[DP] = -s;
      p[DP + 1] = i < i_end ? (T)1 : (T)0;  // rows past the end contribute zero
    }
    __syncthreads();
#pragma unroll 2
    for (int kk = 0; kk < 8; ++kk) {
      const int k = rh * 8 + kk;
      const T* p = &Xs[k * (DP + 2)];
      T sa = p[DP] - za2, sb = p[DP] - zb2;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        sa = mgp_fma(za[d], p[d], sa);
        sb = mgp_fma(zb[d], p[d], sb);
      }
      const T live = p[DP + 1];
      Ka[c * CS + k] = a_ok ? live * mgp_profile<KIND, T>(sa, clamp) : (T)0;
      if (!diag) Kb[c * CS + k] = b_ok ? live * mgp_profile<KIND, T>(sb, clamp) : (T)0;
    }
    __syncthreads();
    const T* Kbp = diag ? Ka : Kb;
#pragma unroll
    for (int ks = 0; ks < CK; ks += 4) {
      T af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = Ka[(wm * 64 + m * 16 + (lane & 15)) * CS + ks + (lane >> 4)];
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[q] = Kbp[(wn * 64 + q * 16 + (lane & 15)) * CS + ks + (lane >> 4)];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = MfmaT<T>::run(af[m], bf[q], acc[m][q]);
    }
  }
  // partial[split][tile][128][128]
  T* o = part + ((long)blockIdx.z * gridDim.x + blockIdx.x) * (CT * CT);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = wm * 64 + m * 16 + MfmaT<T>::row(lane, r);
        const int b = wn * 64 + q * 16 + (lane & 15);
        o[a * CT + b] = acc[m][q][r];
      }
}

// out[a,b] = var^2 * sum_split partial ; mirrored into the lower triangle
template <typename T>
__global__ __launch_bounds__(256) void kmn_knm_reduce_kernel(const T* __restrict__ part, int nsplit, int ntiles,
                                                             const int* __restrict__ tile_ab, long M,
                                                             T* __restrict__ out, T var2) {
  const int tile = blockIdx.x;
  const int ta = tile_ab[2 * tile], tb = tile_ab[2 * tile + 1];
  for (int e = threadIdx.x; e < CT * CT; e += 256) {
    T s = 0;
    for (int z = 0; z < nsplit; ++z) s += part[((long)z * ntiles + tile) * (CT * CT) + e];
    s *= var2;
    const long a = (long)ta * CT + e / CT, b = (long)tb * CT + e % CT;
    if (a < M && b < M) {
      out[a * M + b] = s;
      if (ta != tb) out[b * M + a] = s;
    }
  }
}

template <typename T, int KIND>
int kmn_knm_dp(mgp_handle* h, const SweepParams& prm, int D, const T* X, long N, const T* Z, long M, T* out) {
  const int nt = (int)((M + CT - 1) / CT);
  const int ntiles = nt * (nt + 1) / 2;
  long nsplit = (4L * h->num_cus + ntiles - 1) / ntiles;
  if (nsplit < 1) nsplit = 1;
  const long max_split = (N + 4095) / 4096;  // at least 4096 rows per split
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  long rows = (N + nsplit - 1) / nsplit;
  rows = (rows + CK - 1) / CK * CK;
  nsplit = (N + rows - 1) / rows;
  if (nsplit < 1) nsplit = 1;
  const size_t tab_bytes = (size_t)ntiles * 2 * sizeof(int);
  const size_t part_bytes = (size_t)nsplit * ntiles * CT * CT * sizeof(T);
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, part_bytes + tab_bytes + 256));
  T* part = (T*)h->ws;
  int* tab_dev = (int*)((char*)h->ws + ((part_bytes + 255) & ~(size_t)255));
  std::string tab;
  tab.resize(tab_bytes);
  int* tp = (int*)&tab[0];
  int e = 0;
  for (int a = 0; a < nt; ++a)
    for (int b = a; b < nt; ++b) {
      tp[2 * e] = a;
      tp[2 * e + 1] = b;
      ++e;
    }
  MGP_HIP(h, hipMemcpyAsync(tab_dev, tp, tab_bytes, hipMemcpyHostToDevice, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));  // tab is a host temporary
  dim3 grid((unsigned)ntiles, 1, (unsigned)nsplit);
#define MGP_CT(DPV)                                                                                          \
  hipLaunchKernelGGL((kmn_knm_kernel<T, DPV, KIND>), grid, dim3(256), 0, h->stream, X, N, Z, M, part, rows, D, \
                     prm, (const int*)tab_dev)
  if (D <= 2) MGP_CT(2);
  else if (D <= 4) MGP_CT(4);
  else if (D <= 8) MGP_CT(8);
  else if (D <= 16) MGP_CT(16);
  else MGP_CT(32);
#undef MGP_CT
  MGP_LAUNCH_CHECK(h);
  hipLaunchKernelGGL((kmn_knm_reduce_kernel<T>), dim3((unsigned)ntiles), dim3(256), 0, h->stream, (const T*)part,
                     (int)nsplit, ntiles, (const int*)tab_dev, M, out, (T)(prm.variance * prm.variance));
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
int kmn_knm_t(mgp_handle* h, const mgp_kernel* k, const T* X, long N, const T* Z, long M, T* out) {
  const SweepParams prm = mgp_make_params(k);
  switch (k->kind) {
    case MGP_SE: return kmn_knm_dp<T, 0>(h, prm, k->D, X, N, Z, M, out);
    case MGP_MATERN12: return kmn_knm_dp<T, 1>(h, prm, k->D, X, N, Z, M, out);
    case MGP_MATERN32: return kmn_knm_dp<T, 2>(h, prm, k->D, X, N, Z, M, out);
    default: return kmn_knm_dp<T, 3>(h, prm, k->D, X, N, Z, M, out);
  }
}

}  // namespace

// Two-stage form (default): per chunk of rows, materialise K^T = k(Z, X_chunk) [M, rows] once
// (N*M kernel evaluations in total instead of ~33x that in the fused tiles) and accumulate
// K^T K^T^T on upper-triangular 128x128 tiles with the NT MFMA GEMM of dense.hip; chunks are
// accumulated in order (deterministic), the lower triangle is mirrored at the end.
static int kmn_knm_two_stage(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                             int64_t M, void* out) {
  const size_t es = mgp_elem(k->dtype);
  // 528 upper tiles on 512 resident workgroups would leave a full-length tail round, so every
  // launch contracts NZ independent column slices of the panel (grid.z) into NZ accumulators.
  const int NZ = h->contract_nz;
  int64_t rows = (int64_t)((h->contract_panel_mb << 20) / ((size_t)M * es));  // panel columns per launch
  rows = rows / (16 * NZ) * (16 * NZ);
  if (rows < 16 * NZ) rows = 16 * NZ;
  const size_t panel = (size_t)M * rows * es, slices = (size_t)NZ * M * M * es;
  const int nt = (int)((M + 127) / 128), ntiles = nt * (nt + 1) / 2;
  const size_t tab_bytes = (size_t)ntiles * 2 * sizeof(int);
  MGP_TRY(mgp_reserve(h, &h->opws, &h->opws_bytes, panel + slices + tab_bytes + 256));
  char* Kt = (char*)h->opws;
  char* acc = Kt + panel;
  int* tab_dev = (int*)(acc + ((slices + 255) & ~(size_t)255));
  {
    std::string tab;
    tab.resize(tab_bytes);
    int* tp = (int*)&tab[0];
    int e = 0;
    for (int a = 0; a < nt; ++a)
      for (int b = a; b < nt; ++b) {
        tp[2 * e] = a;
        tp[2 * e + 1] = b;
        ++e;
      }
    MGP_HIP(h, hipMemcpyAsync(tab_dev, tp, tab_bytes, hipMemcpyHostToDevice, h->stream));
    MGP_HIP(h, hipStreamSynchronize(h->stream));  // tab is a host temporary
  }
  MGP_HIP(h, hipMemsetAsync(acc, 0, slices, h->stream));
  for (int64_t i0 = 0; i0 < N; i0 += rows) {
    const int64_t rc = (N - i0 < rows) ? N - i0 : rows;
    const char* Xc = (const char*)X + (size_t)i0 * k->D * es;
    if (rc == rows) {
      MGP_TRY(mgp_k_dense(h, k, Z, M, Xc, rc, Kt, rc, 0.0, nullptr));
      MGP_TRY(mgp_syrk_nt_upper(h, k->dtype, Kt, M, rc / NZ, rc, acc, 1, NZ, tab_dev, ntiles));
    } else {  // ragged tail: one slice
      MGP_TRY(mgp_k_dense(h, k, Z, M, Xc, rc, Kt, rc, 0.0, nullptr));
      MGP_TRY(mgp_syrk_nt_upper(h, k->dtype, Kt, M, rc, rc, acc, 1, 1, tab_dev, ntiles));
    }
  }
  return mgp_mirror_upper(h, k->dtype, out, acc, NZ, M, 1.0);
}

extern "C" int mgp_kmn_knm(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                           int64_t M, void* out) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (N < 0 || M < 0) return mgp_fail(h, MGP_E_SHAPE, "negative size");
  if (M == 0) return MGP_OK;
  if (!Z || !out || (N > 0 && !X)) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (N == 0) {
    MGP_HIP(h, hipMemsetAsync(out, 0, (size_t)M * M * mgp_elem(k->dtype), h->stream));
    return MGP_OK;
  }
  {
    const char* mode = getenv("MGP_CONTRACT");  // "fused" keeps the single-kernel form for A/B runs
    if (k->D > MGP_FUSED_MAX_D || !(mode && strcmp(mode, "fused") == 0)) return kmn_knm_two_stage(h, k, X, N, Z, M, out);
  }
  if (k->dtype == MGP_F64) return kmn_knm_t<double>(h, k, (const double*)X, N, (const double*)Z, M, (double*)out);
  return kmn_knm_t<float>(h, k, (const float*)X, N, (const float*)Z, M, (float*)out);
}
