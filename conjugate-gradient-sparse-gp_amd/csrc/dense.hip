// dense.hip -- explicit kernel blocks and the dense symmetric product of the CG step.
//
//   mgp_k_dense      out[na, nb] = k(A, B) (+ jitter, + diag_add)       rows K3, K4
//   mgp_symm_matmul  out[Bt, n] = P[Bt, n] @ A[n, n], A symmetric       row M2 (`p @ A`)
//
// The product has three regimes: Bt = 1 is a GEMV (HBM-bound: A is read once, s(n^2 + 2n)
// bytes; coalesced 16-byte row reads, wavefront reductions); 2 <= Bt <= 128 keeps that single
// pass over A and does the Bt-wide contraction on the matrix cores (symm_skinny_kernel); larger
// Bt is an LDS-tiled NT GEMM on v_mfma_f64_16x16x4_f64 (v_mfma_f32_16x16x4_f32 for fp32), which
// also serves the K_mn K_nm contraction (contract.hip) and the generic-D products (generic.hip).
#include <algorithm>
#include <cstdlib>

#include "mgp_common.h"

namespace {

// ------------------------------------------------------------------ k_dense
// TA rows of A per block (staged in LDS, read back as wave-uniform broadcasts), one column of B per thread (held in
// registers).  A thread loads its B point once per block, so TA sets how often B is re-read from L2: at D = 32 a
// point is 32 loads for TA pairs -- with TA = 16 the C5 contraction's panel kernel read 4.3 GB of B through L2 to
// write 2.1 GB (2.13 ms per panel); TA = 64 there.
template <typename T, int DP, int KIND, int TA>
__global__ __launch_bounds__(256) void k_dense_kernel(const T* __restrict__ A, long na,
                                                      const T* __restrict__ B, long nb, T* __restrict__ out,
                                                      long ld, int D, SweepParams prm, T jitter,
                                                      const T* __restrict__ diag_add) {
  constexpr int PS = (DP + 1 + 1) & ~1;
  __shared__ __attribute__((aligned(16))) T tile[TA * PS];
  const int t = threadIdx.x;
  const long j = (long)blockIdx.x * 256 + t;
  const long i0 = (long)blockIdx.y * TA;
  if (t < TA) {
    const long i = i0 + t;
    T* p = &tile[t * PS];
    T s = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      T v = (d < D && i < na) ? A[i * D + d] * (T)prm.inv_ls[d] : (T)0;
      s = mgp_fma(v, v, s);
      p[d] = v + v;
    }
    p[DP] = -s;
  }
  T b[DP];
  T b2 = 0;
  {
    const long jc = j < nb ? j : nb - 1;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      T v = d < D ? B[jc * D + d] * (T)prm.inv_ls[d] : (T)0;
      b[d] = v;
      b2 = mgp_fma(v, v, b2);
    }
  }
  __syncthreads();
  if (j >= nb) return;
  const T var = (T)prm.variance;
  const T clamp = (T)prm.clamp;
#pragma unroll 4
  for (int ii = 0; ii < TA; ++ii) {
    const long i = i0 + ii;
    if (i >= na) break;
    const T* p = &tile[ii * PS];
    T s = p[DP] - b2;
#pragma unroll
    for (int d = 0; d < DP; ++d) s = mgp_fma(b[d], p[d], s);
    T v = var * mgp_profile<KIND, T>(s, clamp);
    if (i == j) {
      v += jitter;
      if (diag_add != nullptr) v += diag_add[i];
    }
    out[i * ld + j] = v;
  }
}

template <typename T, int KIND>
int k_dense_dp(mgp_handle* h, const SweepParams& prm, int D, const T* A, long na, const T* B, long nb, T* out,
               long ld, T jitter, const T* diag_add) {
  // rows of A per block: 64 when a B point is expensive to load (D > 8) or there are enough blocks anyway
  const int ta = (h->kdense_ta > 0) ? h->kdense_ta : ((D > 8 && na >= 64) ? 64 : 16);
  dim3 grid((unsigned)((nb + 255) / 256), (unsigned)((na + ta - 1) / ta));
#define MGP_KD(DPV)                                                                                                  \
  do {                                                                                                               \
    if (ta == 64)                                                                                                    \
      hipLaunchKernelGGL((k_dense_kernel<T, DPV, KIND, 64>), grid, dim3(256), 0, h->stream, A, na, B, nb, out, ld, D, \
                         prm, jitter, diag_add);                                                                     \
    else                                                                                                             \
      hipLaunchKernelGGL((k_dense_kernel<T, DPV, KIND, 16>), grid, dim3(256), 0, h->stream, A, na, B, nb, out, ld, D, \
                         prm, jitter, diag_add);                                                                     \
  } while (0)
  if (D <= 2) MGP_KD(2);
  else if (D <= 4) MGP_KD(4);
  else if (D <= 8) MGP_KD(8);
  else if (D <= 16) MGP_KD(16);
  else MGP_KD(32);
#undef MGP_KD
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
int k_dense_t(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, T* out, long ld,
              double jitter, const T* diag_add) {
  const SweepParams prm = mgp_make_params(k);
  switch (k->kind) {
    case MGP_SE: return k_dense_dp<T, 0>(h, prm, k->D, A, na, B, nb, out, ld, (T)jitter, diag_add);
    case MGP_MATERN12: return k_dense_dp<T, 1>(h, prm, k->D, A, na, B, nb, out, ld, (T)jitter, diag_add);
    case MGP_MATERN32: return k_dense_dp<T, 2>(h, prm, k->D, A, na, B, nb, out, ld, (T)jitter, diag_add);
    default: return k_dense_dp<T, 3>(h, prm, k->D, A, na, B, nb, out, ld, (T)jitter, diag_add);
  }
}

// ------------------------------------------------------------------ GEMV (one right-hand side)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wave per RW rows of A; lanes stride the row in VEC-element pieces (RW*VEC*sizeof(T)*64 bytes
// of A in flight per wave and loop trip, unrolled by 2)
template <typename T, int BT, int VEC>
__global__ __launch_bounds__(256) void symm_gemv_kernel(const T* __restrict__ A, long n,
                                                        const T* __restrict__ P, int bt, T* __restrict__ out,
                                                        const int* __restrict__ gate, long row_begin, long row_end,
                                                        T alpha, int accumulate, T* __restrict__ word = nullptr) {
  // `word` (multi-rank SGPR operator): this rank's agreement word behind the partial, 1 iff it computed this
  // application -- written here instead of by a launch of its own (put_gate_word_kernel, 4.6 us per step)
  const bool open = gate == nullptr || *gate != 0;
  if (word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *word = open ? (T)1 : (T)0;
  if (!open) return;
  constexpr int RW = 2;  // 4 rows per wave measured the same at n=4096 and 3% slower at n=8192
  const int lane = threadIdx.x & 63;
  const long row0 = row_begin + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= row_end) return;
  T acc[RW][BT];
#pragma unroll
  for (int q = 0; q < RW; ++q)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[q][b] = 0;
  const T* ar[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) ar[q] = A + (row0 + q < row_end ? row0 + q : row0) * n;
#pragma unroll 2
  for (long i = (long)lane * VEC; i < n; i += 64 * VEC) {
    T x[RW][VEC];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      if (VEC == 1) {
        x[q][0] = ar[q][i];
      } else {
        // VEC*sizeof(T) == 16 bytes, rows are 16-B aligned because n % VEC == 0
        using V = __attribute__((ext_vector_type(VEC))) T;
        const V v = *reinterpret_cast<const V*>(ar[q] + i);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[q][e] = v[e];
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b < bt) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const T pv = P[(long)b * n + i + e];
#pragma unroll
          for (int q = 0; q < RW; ++q) acc[q][b] = mgp_fma(x[q][e], pv, acc[q][b]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RW; ++q)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const T s = wave_sum(acc[q][b]);
      if (lane == 0 && b < bt && row0 + q < row_end) {
        T* o = &out[(long)b * n + row0 + q];
        *o = accumulate ? mgp_fma(alpha, s, *o) : s;
      }
    }
}

// A rank's slab of the replicated s2 Kmm.p term (few rows: 512 of 4096 at 8 ranks), accumulated into its partial:
// out[row] += alpha * A[row, :] . p.  One workgroup per 2 rows with the COLUMNS split over its four waves, so that
// 256 workgroups x 4 waves x 4 KB are in flight on the chip (one wave per 2 rows left 1 MB in flight: 12.5 us for
// 16.8 MB whether the waves sat in 64 or in 256 workgroups); the four column sums of a row are added in wave order.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void symm_gemv_slab_kernel(const T* __restrict__ A, long n, const T* __restrict__ p,
                                                             T* __restrict__ out, const int* __restrict__ gate,
                                                             long row_begin, long row_end, T alpha,
                                                             T* __restrict__ word) {
  const bool open = gate == nullptr || *gate != 0;
  if (word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *word = open ? (T)1 : (T)0;  // see symm_gemv_kernel
  if (!open) return;
  __shared__ T part[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = row_begin + (long)blockIdx.x * 2;
  const T* a0 = A + row0 * n;
  const T* a1 = A + (row0 + 1 < row_end ? row0 + 1 : row0) * n;
  // columns [c0, c1) of this wave: quarters rounded to whole vector groups of the wave
  const long per = ((n + 3) / 4 + 64 * VEC - 1) / (64 * VEC) * (64 * VEC);
  const long c0 = (long)wave * per, c1 = c0 + per < n ? c0 + per : n;
  T s0 = 0, s1 = 0;
#pragma unroll 4
  for (long i = c0 + (long)lane * VEC; i < c1; i += 64 * VEC) {
    if (VEC == 1) {
      const T pv = p[i];
      s0 = mgp_fma(a0[i], pv, s0);
      s1 = mgp_fma(a1[i], pv, s1);
    } else {
      using V = __attribute__((ext_vector_type(VEC))) T;
      const V x0 = *reinterpret_cast<const V*>(a0 + i), x1 = *reinterpret_cast<const V*>(a1 + i);
      const V pv = *reinterpret_cast<const V*>(p + i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s0 = mgp_fma(x0[e], pv[e], s0);
        s1 = mgp_fma(x1[e], pv[e], s1);
      }
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    part[wave][0] = s0;
    part[wave][1] = s1;
  }
  __syncthreads();
  if (threadIdx.x < 2 && row0 + threadIdx.x < row_end) {
    const int q = threadIdx.x;
    const T s = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
    T* o = &out[row0 + q];
    *o = mgp_fma(alpha, s, *o);
  }
}

// ------------------------------------------------------------------ one-RHS product on the upper triangle
// out = A p for symmetric A reading every off-diagonal 64x64 tile once: the workgroup of tile (I,J),
// I <= J, forms both A_IJ p_J (a contribution to rows of chunk I) and A_IJ^T p_I (to chunk J), so a
// CG step streams n^2/2 elements instead of n^2.  Contributions go to Q[other chunk][i] (each slot
// has exactly one writer) and a second small kernel adds the nt slots of every row in index order:
// deterministic, no atomics.  (A single-launch form with an arrival ticket per chunk was measured
// slower: 2080 workgroups each publishing 1 KB write-through cost more than the second launch.)
template <typename T>
__global__ __launch_bounds__(256) void symm_gemv_tri_kernel(const T* __restrict__ A, long n,
                                                            const T* __restrict__ p, T* __restrict__ Q, int nt,
                                                            const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  constexpr int TS = 64;
  __shared__ T rowp[TS][TS + 1];
  __shared__ T colp[4][TS];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // block -> tile of the row-major enumeration of the upper triangle: row I starts at I nt - I (I-1)/2
  const long b = blockIdx.x;
  const double q2 = 2.0 * nt + 1.0;
  int I = (int)((q2 - sqrt(q2 * q2 - 8.0 * (double)b)) * 0.5);
  if (I < 0) I = 0;
  if (I > nt - 1) I = nt - 1;
  while (I < nt - 1 && (long)(I + 1) * nt - (long)(I + 1) * I / 2 <= b) ++I;
  while (I > 0 && (long)I * nt - (long)I * (I - 1) / 2 > b) --I;
  const int J = I + (int)(b - ((long)I * nt - (long)I * (I - 1) / 2));
  const long r0 = (long)I * TS + 16 * w, c = (long)J * TS + l;
  T a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = (r0 + r < n && c < n) ? A[(r0 + r) * n + c] : (T)0;
  const T pj = c < n ? p[c] : (T)0;
  T cs = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    rowp[16 * w + r][l] = a[r] * pj;
    const T pi = r0 + r < n ? p[r0 + r] : (T)0;  // wave-uniform address
    cs = mgp_fma(a[r], pi, cs);
  }
  colp[w][l] = cs;
  __syncthreads();
  {  // rows of chunk I: thread (row, quarter) adds 16 columns, the quad finishes the row
    const int row = t >> 2, qd = t & 3;
    T s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += rowp[row][qd * 16 + k];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const long i = (long)I * TS + row;
    if (qd == 0 && i < n) Q[(long)J * n + i] = s;
  }
  if (I != J && t < TS) {
    const T s = (colp[0][t] + colp[1][t]) + (colp[2][t] + colp[3][t]);
    const long i = (long)J * TS + t;
    if (i < n) Q[(long)I * n + i] = s;
  }
}

// One step of the reduce-scatter over lanes: lanes whose BIT is set keep the upper HALF of the values they carry,
// the others the lower; each adds its partner's copy of what it keeps.  (Template constants: with run-time loop
// bounds the compiler indexed x[] dynamically -- 859 v_cndmask in the first version of this kernel.)
template <typename T, int HALF, int BIT>
__device__ __forceinline__ void lane_reduce_scatter_step(T (&x)[16], int l) {
  const bool hi = (l & BIT) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const T keep = hi ? x[k + HALF] : x[k];
    const T send = hi ? x[k] : x[k + HALF];
    x[k] = keep + __shfl_xor(send, BIT, 64);
  }
}

// (I, J) of every upper-triangle tile in launch order, built once per nt (the sqrt + correction loops that round 1s
// kernel runs per workgroup are ~300 scalar instructions in front of its first load)
__global__ void tri_tile_table_kernel(int2* __restrict__ tab, int nt) {
  const int I = blockIdx.x;
  const long start = (long)I * nt - (long)I * (I - 1) / 2;
  for (int J = I + threadIdx.x; J < nt; J += blockDim.x) tab[start + (J - I)] = make_int2(I, J);
}

// Round 3 form of the tile kernel.  Same tiles, same slots, but the 64 row sums of a tile no longer go through a
// 33 KB LDS staging array: a wave holds 16 rows x 64 columns of products, one row-value per lane and row, and a
// reduce-scatter over the lanes (xor 32, 16, 8, 4: each step halves the rows a lane still carries; then xor 2, 1)
// leaves lane l with the total of row l >> 2 -- 17 64-bit cross-lane moves per wave instead of 16 LDS stores, a
// barrier and 16 LDS loads per lane.  One code path for whole and ragged tiles: addresses are clamped to the last
// row / column and the strays zeroed by selects (a guarded load is a branch; at the join of a guarded and an
// unguarded path the compiler's wait counts are the pessimistic merge of both).  n = 4096: 14.0 -> 12.6 us.
// A two-tiles-per-workgroup variant (1 KB per row and request) measured slower (17.2 us) and was removed.
template <typename T>
__global__ __launch_bounds__(256) void symm_gemv_tri_bfly_kernel(const T* __restrict__ A, long n,
                                                                 const T* __restrict__ p, T* __restrict__ Q,
                                                                 const int2* __restrict__ tab,
                                                                 const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  constexpr int TS = 64;
  __shared__ T colp[4][TS];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int2 ij = tab[blockIdx.x];  // uniform: a scalar load
  const int I = ij.x, J = ij.y;
  const long r0 = (long)I * TS + 16 * w, c = (long)J * TS + l, ci = (long)I * TS + l;
  const long cj = c < n ? c : n - 1, cic = ci < n ? ci : n - 1;
  const T pj_raw = p[cj], pi_raw = p[cic];
  T a[16];
  {
    const T* row = A + (r0 < n ? r0 : n - 1) * n + cj;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      a[q] = *row;
      row += (r0 + q + 1 < n) ? n : 0;
    }
  }
  const T pj = c < n ? pj_raw : (T)0, pi = ci < n ? pi_raw : (T)0;
  T x[16];
  T cs = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const T aq = (r0 + q < n && c < n) ? a[q] : (T)0;
    x[q] = aq * pj;
    cs = mgp_fma(aq, mgp_read_lane(pi, 16 * w + q), cs);
  }
  colp[w][l] = cs;
  lane_reduce_scatter_step<T, 8, 32>(x, l);
  lane_reduce_scatter_step<T, 4, 16>(x, l);
  lane_reduce_scatter_step<T, 2, 8>(x, l);
  lane_reduce_scatter_step<T, 1, 4>(x, l);
  T s = x[0];
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  const long i = r0 + (l >> 2);
  if ((l & 3) == 0 && i < n) Q[(long)J * n + i] = s;
  __syncthreads();
  if (t < TS && I != J) {
    const T sc = (colp[0][t] + colp[1][t]) + (colp[2][t] + colp[3][t]);
    const long i2 = (long)J * TS + t;
    if (i2 < n) Q[(long)I * n + i2] = sc;
  }
}

// out[i] = sum_k Q[k][i], k ascending in four fixed runs per row
template <typename T>
__global__ __launch_bounds__(256) void symm_gemv_tri_reduce_kernel(const T* __restrict__ Q, long n, int nt,
                                                                   T* __restrict__ out,
                                                                   const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  __shared__ T part[4][64];
  const int t = threadIdx.x, l = t & 63, pt = t >> 6;
  const int per = (nt + 3) / 4;
  const int kb = pt * per, ke = (kb + per < nt) ? kb + per : nt;
  const long i = (long)blockIdx.x * 64 + l;
  T s = 0;
  if (i < n) {
#pragma unroll 8
    for (int k = kb; k < ke; ++k) s += Q[(long)k * n + i];
  }
  part[pt][l] = s;
  __syncthreads();
  if (t < 64 && i < n) out[i] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// ------------------------------------------------------------------ MFMA GEMM
template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using Acc = __attribute__((ext_vector_type(4))) double;
  static __device__ __forceinline__ Acc run(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mfma<float> {
  using Acc = __attribute__((ext_vector_type(4))) float;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + reg
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// out[m, n] (+)= P[m, K] . A[n, K]^T ("NT": both operands contiguous along the contraction index).
// Block tile (32 WT) x (32 WT), BK = 16, 4 waves as 2x2, each wave WT x WT MFMA tiles of 16x16
// (WT = 4: 128x128 tile, the throughput shape; WT = 2: 64x64, used when the 128-tile grid would
// leave CUs idle).  Operand tiles are register-prefetched one step ahead and staged through LDS
// with a conflict-free stride.  `upper_only` skips tiles strictly below the diagonal.
// UA (with VEC = false): any K and leading dimensions -- whole k steps use 16-byte loads that carry element
// alignment only, the partial last step is loaded element by element with zeros past K.
template <typename T, bool VEC, int WT, bool UA = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const T* __restrict__ P, long ldp, long m,
                                                      const T* __restrict__ A, long lda, long n, long K,
                                                      T* __restrict__ out, long ldo, int accumulate, int upper_only,
                                                      const int* __restrict__ gate, long zstride_k, long zstride_out,
                                                      const int* __restrict__ tile_tab, int ntiles) {
  if (gate != nullptr && *gate == 0) return;
  // 2-D grid (tile_tab == nullptr): blockIdx.{y,x} = output tile, blockIdx.z = contraction slice.
  // 1-D grid: blockIdx.x = slice * ntiles + entry of tile_tab (the upper-triangular tiles of a
  // symmetric result only, so no workgroup is launched for the skipped half).
  int by, bx, bz;
  if (tile_tab != nullptr) {
    bz = blockIdx.x / ntiles;
    const int e = blockIdx.x - bz * ntiles;
    by = tile_tab[2 * e];
    bx = tile_tab[2 * e + 1];
  } else {
    by = blockIdx.y;
    bx = blockIdx.x;
    bz = blockIdx.z;
    if (upper_only && bx < by) return;
  }
  // each slice of the contraction index has its own output buffer
  P += (long)bz * zstride_k;
  A += (long)bz * zstride_k;
  out += (long)bz * zstride_out;
  if (UA && zstride_k > 0) {  // UA slices: K is the TOTAL length, a slice is zstride_k long and the last one what is left
    const long left = K - (long)bz * zstride_k;
    K = left < zstride_k ? left : zstride_k;
  }
  constexpr int BM = 32 * WT, BK = 16, LDS_S = BK + 2;  // stride 18: conflict-free b64 reads
  constexpr int EPT = BM * BK / 256;                      // elements per thread and operand per step
  __shared__ __attribute__((aligned(16))) T Ps[BM * LDS_S];
  __shared__ __attribute__((aligned(16))) T As[BM * LDS_S];
  using Acc = typename Mfma<T>::Acc;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long b0 = (long)by * BM, j0 = (long)bx * BM;

  Acc acc[WT][WT];
#pragma unroll
  for (int mi = 0; mi < WT; ++mi)
#pragma unroll
    for (int q = 0; q < WT; ++q) acc[mi][q] = Acc{0, 0, 0, 0};

  // staging: a tile is BM rows x 16 k; a thread loads EPT consecutive k of one row of each operand
  const int srow = t / (BK / EPT), skk = (t % (BK / EPT)) * EPT;
  const long pb = b0 + srow, aj = j0 + srow;
  const bool p_ok = pb < m, a_ok = aj < n;
  const T* prow = P + (p_ok ? pb : 0) * ldp;
  const T* arow = A + (a_ok ? aj : 0) * lda;
  T pre_p[EPT], pre_a[EPT];
  auto load_tiles = [&](long k0) {
    const long k = k0 + skk;
    if (VEC) {  // host guarantees K % 16 == 0, 16-byte aligned rows
      constexpr int VW = 16 / sizeof(T);
      using V = __attribute__((ext_vector_type(VW))) T;
#pragma unroll
      for (int e = 0; e < EPT; e += VW) {
        const V vp = p_ok ? *reinterpret_cast<const V*>(prow + k + e) : V{};
        const V va = a_ok ? *reinterpret_cast<const V*>(arow + k + e) : V{};
#pragma unroll
        for (int x = 0; x < VW; ++x) {
          pre_p[e + x] = vp[x];
          pre_a[e + x] = va[x];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        pre_p[e] = (p_ok && k + e < K) ? prow[k + e] : (T)0;
        pre_a[e] = (a_ok && k + e < K) ? arow[k + e] : (T)0;
      }
    }
  };
  load_tiles(0);
  if constexpr ((VEC || UA) && WT == 4) {
    // Throughput shape, whole 16-element k steps: the next step's operand loads carry no predicate (rows beyond
    // m / n are clamped to row 0 and their results never written; the step after the last re-reads the last) and
    // go out in four pairs, one behind each group of 16 MFMAs -- not as a burst of eight in front of them, during
    // which the in-order wave issues no MFMA (the skinny product's timeline).
    constexpr int VW = 16 / sizeof(T);
    constexpr int NL = EPT / VW;  // 16-byte loads per operand, thread and step
    typedef T Vr __attribute__((ext_vector_type(VW)));
    typedef Vr V __attribute__((aligned(UA ? sizeof(T) : 16)));
    Vr np[NL], na_[NL];  // staged at the top of the next step straight from these registers: no copy of a load in flight
#pragma unroll
    for (int u = 0; u < NL; ++u)
#pragma unroll
      for (int x = 0; x < VW; ++x) {
        np[u][x] = pre_p[u * VW + x];
        na_[u][x] = pre_a[u * VW + x];
      }
    for (long k0 = 0; k0 < K; k0 += BK) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NL; ++u)
#pragma unroll
        for (int x = 0; x < VW; ++x) {
          Ps[srow * LDS_S + skk + u * VW + x] = np[u][x];
          As[srow * LDS_S + skk + u * VW + x] = na_[u][x];
        }
      __syncthreads();
      // past the last step the request is a dummy: the current step again, or (UA: it may be the partial one) step 0
      const long kn = (k0 + BK < K ? k0 + BK : (UA ? 0 : k0)) + skk;
      const bool part_next = UA && (k0 + BK < K) && (k0 + 2 * BK > K);  // uniform: the step being requested is the partial one
      // operand fragments: lane group g takes k = 4g .. 4g+3 of the 16-wide step (MFMA j contracts k = 4g + j over the
      // four groups) -- four consecutive elements per tile, read as 16-byte LDS loads, instead of k = ks + g, one
      // element per MFMA (half the LDS instructions, no 2-way conflicts of the compiler's ds_read2_b64 pairing)
      T afa[WT][4], bfa[WT][4];
#pragma unroll
      for (int mi = 0; mi < WT; ++mi)
#pragma unroll
        for (int x = 0; x < 4; ++x)
          afa[mi][x] = Ps[(wm * 16 * WT + mi * 16 + (lane & 15)) * LDS_S + 4 * (lane >> 4) + x];
#pragma unroll
      for (int q = 0; q < WT; ++q)
#pragma unroll
        for (int x = 0; x < 4; ++x)
          bfa[q][x] = As[(wn * 16 * WT + q * 16 + (lane & 15)) * LDS_S + 4 * (lane >> 4) + x];
#pragma unroll
      for (int ks = 0; ks < BK; ks += 4) {
#pragma unroll
        for (int mi = 0; mi < WT; ++mi)
#pragma unroll
          for (int q = 0; q < WT; ++q) acc[mi][q] = Mfma<T>::run(afa[mi][ks / 4], bfa[q][ks / 4], acc[mi][q]);
        constexpr int G = (NL + 3) / 4;  // loads per operand behind this group
#pragma unroll
        for (int u = (ks / 4) * G; u < (ks / 4 + 1) * G && u < NL; ++u) {
          if (UA && part_next) {
#pragma unroll
            for (int x = 0; x < VW; ++x) {
              const long kk = kn + u * VW + x;
              np[u][x] = kk < K ? prow[kk] : (T)0;
              na_[u][x] = kk < K ? arow[kk] : (T)0;
            }
          } else {
            np[u] = *reinterpret_cast<const V*>(prow + kn + u * VW);
            na_[u] = *reinterpret_cast<const V*>(arow + kn + u * VW);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else
  for (long k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      Ps[srow * LDS_S + skk + e] = pre_p[e];
      As[srow * LDS_S + skk + e] = pre_a[e];
    }
    __syncthreads();
    if (k0 + BK < K) load_tiles(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      T af[WT], bf[WT];
#pragma unroll
      for (int mi = 0; mi < WT; ++mi)
        af[mi] = Ps[(wm * 16 * WT + mi * 16 + (lane & 15)) * LDS_S + ks + (lane >> 4)];
#pragma unroll
      for (int q = 0; q < WT; ++q) bf[q] = As[(wn * 16 * WT + q * 16 + (lane & 15)) * LDS_S + ks + (lane >> 4)];
#pragma unroll
      for (int mi = 0; mi < WT; ++mi)
#pragma unroll
        for (int q = 0; q < WT; ++q) acc[mi][q] = Mfma<T>::run(af[mi], bf[q], acc[mi][q]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < WT; ++mi)
#pragma unroll
    for (int q = 0; q < WT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long b = b0 + wm * 16 * WT + mi * 16 + Mfma<T>::row(lane, r);
        const long j = j0 + wn * 16 * WT + q * 16 + (lane & 15);
        if (b < m && j < n) {
          T* o = &out[b * ldo + j];
          *o = accumulate ? *o + acc[mi][q][r] : acc[mi][q][r];
        }
      }
}

template <typename T>
inline bool gemm_vec_ok(const T* P, long ldp, const T* A, long lda, long K) {
  return (K % 16) == 0 && ((ldp | lda) % (16 / (long)sizeof(T))) == 0 && ((((uintptr_t)P) | ((uintptr_t)A)) % 16) == 0;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_slices_sum_kernel(const T* __restrict__ part, long tot, int nz,
                                                              T* __restrict__ out, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= tot) return;
  T s = part[e];
  for (int z = 1; z < nz; ++z) s += part[(long)z * tot + e];  // slice order: deterministic
  out[e] = s;
}

// out[m, n] (+)= P[m,K] . A[n,K]^T.  Shape policy: 128x128 tiles when they fill the CUs twice over;
// otherwise, when the contraction is long enough, 128x128 tiles x nz slices of K (each slice its own
// partial output, summed in slice order) so that the throughput tile still sees >= 2 workgroups per
// CU; otherwise 64x64 tiles.
template <typename T>
int gemm_nt_launch(mgp_handle* h, const T* P, long ldp, long m, const T* A, long lda, long n, long K, T* out,
                   long ldo, int accumulate, const int* gate) {
  const bool vec = gemm_vec_ok<T>(P, ldp, A, lda, K);
  const long big = ((n + 127) / 128) * ((m + 127) / 128);
#define MGP_GEMM(VV, WTV)                                                                                        \
  do {                                                                                                           \
    dim3 grid((unsigned)((n + 32 * WTV - 1) / (32 * WTV)), (unsigned)((m + 32 * WTV - 1) / (32 * WTV)));         \
    hipLaunchKernelGGL((gemm_nt_kernel<T, VV, WTV>), grid, dim3(256), 0, h->stream, P, ldp, m, A, lda, n, K, out, \
                       ldo, accumulate, 0, gate, 0L, 0L, (const int*)nullptr, 0);                                \
  } while (0)
  if (big >= 2L * h->num_cus) {
    if (vec) {
      MGP_GEMM(true, 4);
    } else if (K >= 32) {  // ragged K / unaligned rows: element-aligned vector loads, element-wise partial step
      dim3 grid((unsigned)((n + 127) / 128), (unsigned)((m + 127) / 128));
      hipLaunchKernelGGL((gemm_nt_kernel<T, false, 4, true>), grid, dim3(256), 0, h->stream, P, ldp, m, A, lda, n, K,
                         out, ldo, accumulate, 0, gate, 0L, 0L, (const int*)nullptr, 0);
    } else {
      MGP_GEMM(false, 4);
    }
  } else {
    long nz = (2L * h->num_cus + big - 1) / big;
    if (nz > 8) nz = 8;
    while (nz > 1 && (K % (16 * nz) != 0 || K / nz < 256)) --nz;  // slices stay 16-aligned and worth a launch
    if (h->gemm_ksplit && nz > 1 && vec && !accumulate && ldo == n) {
      const long tot = m * n;
      MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nz * tot * sizeof(T)));
      T* part = (T*)h->ws;
      dim3 grid((unsigned)((n + 127) / 128), (unsigned)((m + 127) / 128), (unsigned)nz);
      hipLaunchKernelGGL((gemm_nt_kernel<T, true, 4>), grid, dim3(256), 0, h->stream, P, ldp, m, A, lda, n, K / nz,
                         part, n, 0, 0, gate, K / nz, tot, (const int*)nullptr, 0);
      MGP_LAUNCH_CHECK(h);
      hipLaunchKernelGGL((gemm_slices_sum_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                         (const T*)part, tot, (int)nz, out, gate);
    } else if (h->gemm_ksplit && !vec && !accumulate && ldo == n && K >= 512) {
      // ragged K / unaligned rows, mid-size shape: the same 128x128 tile over slices, through the UA loads
      long nzu = (2L * h->num_cus + big - 1) / big;
      if (nzu > 8) nzu = 8;
      while (nzu > 1 && K / nzu < 256) --nzu;
      const long ksl = ((K + nzu - 1) / nzu + 15) / 16 * 16;
      nzu = (K + ksl - 1) / ksl;
      const long tot = m * n;
      T* part = out;
      if (nzu > 1) {
        MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nzu * tot * sizeof(T)));
        part = (T*)h->ws;
      }
      dim3 grid((unsigned)((n + 127) / 128), (unsigned)((m + 127) / 128), (unsigned)nzu);
      hipLaunchKernelGGL((gemm_nt_kernel<T, false, 4, true>), grid, dim3(256), 0, h->stream, P, ldp, m, A, lda, n, K,
                         part, n, 0, 0, gate, nzu > 1 ? ksl : 0L, nzu > 1 ? tot : 0L, (const int*)nullptr, 0);
      if (nzu > 1) {
        MGP_LAUNCH_CHECK(h);
        hipLaunchKernelGGL((gemm_slices_sum_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                           (const T*)part, tot, (int)nzu, out, gate);
      }
    } else {
      if (vec) MGP_GEMM(true, 2);
      else MGP_GEMM(false, 2);
    }
  }
#undef MGP_GEMM
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

// out[j, i] = out[i, j] for i < j (fills the lower triangle after an upper_only accumulation)
template <typename T>
__global__ __launch_bounds__(256) void mirror_upper_kernel(T* __restrict__ out, const T* __restrict__ slices, int nz,
                                                           long n, T scale) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * n) return;
  const long i = e / n, j = e - i * n;
  if (j >= i) {
    T v = 0;
    for (int z = 0; z < nz; ++z) v += slices[(long)z * n * n + e];  // slices summed in index order
    v *= scale;
    out[e] = v;
    if (j > i) out[j * n + i] = v;
  }
}

// ------------------------------------------------------------------ skinny product (2 <= Bt <= 128)
// out[Bt, n] = P[Bt, n] . A^T with A streamed from HBM exactly once (bytes s(n^2 + 2 n Bt), the
// GEMV roofline) and the Bt-wide contraction on the matrix cores.  A workgroup of 8 waves owns 16
// columns j (= 16 rows of A); wave w takes every 8th 16-wide k slab.  Per slab a lane reads 4
// consecutive k of its row of A (32 B, whole 128-B lines per 4 lanes) and the matching rows of
// P^T (P is transposed and zero-padded to 16*NBT columns once per call, so those reads are
// 128-B segments too); the MFMA K order is permuted consistently on both operands.  The 8 partial
// tiles are summed through LDS in wave order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ P, long Bt, long n,
                                                            T* __restrict__ Pt, int BP,
                                                            const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * BP) return;
  const long k = e / BP;
  const int b = (int)(e - k * BP);
  Pt[e] = b < Bt ? P[(long)b * n + k] : (T)0;
}

template <typename T, int NBT>
__global__ __launch_bounds__(512) void symm_skinny_kernel(const T* __restrict__ A, long n,
                                                          const T* __restrict__ Pt, long Bt,
                                                          T* __restrict__ out, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  constexpr int BP = 16 * NBT;
  using Acc = typename Mfma<T>::Acc;
  __shared__ T red[8 * NBT * 4 * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int jj = lane & 15, g = lane >> 4;
  const long j0 = (long)blockIdx.x * 16;
  const long j = j0 + jj < n ? j0 + jj : n - 1;
  const T* arow = A + j * n;
  Acc acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = Acc{0, 0, 0, 0};
  const bool vec = (n & 3) == 0;
  // software pipeline: the operands of slab k+1 are in flight while slab k runs on the matrix core
  T a[4], an[4];
  T p[NBT][4], pn[NBT][4];
  auto load_slab = [&](long k0, T (&av)[4], T (&pv)[NBT][4]) {
    const long kb = k0 + 4 * g;
    if (vec && kb + 3 < n) {
      using V4 = __attribute__((ext_vector_type(4))) T;
      const V4 v = *reinterpret_cast<const V4*>(arow + kb);
      av[0] = v[0];
      av[1] = v[1];
      av[2] = v[2];
      av[3] = v[3];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = kb + s < n ? arow[kb + s] : (T)0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long k = kb + s;
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) pv[bt][s] = k < n ? Pt[k * BP + bt * 16 + jj] : (T)0;
    }
  };
  long k0 = (long)wave * 16;
  if (k0 < n) load_slab(k0, a, p);
  for (; k0 < n; k0 += 8 * 16) {
    const long kn = k0 + 8 * 16;
    if (kn < n) load_slab(kn, an, pn);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) acc[bt] = Mfma<T>::run(p[bt][s], a[s], acc[bt]);
    if (kn < n) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = an[s];
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) p[bt][s] = pn[bt][s];
      }
    }
  }
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NBT + bt) * 4 + r) * 64 + lane] = acc[bt][r];
  __syncthreads();
  for (int e = t; e < NBT * 4 * 64; e += 512) {
    T s = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * (NBT * 4 * 64) + e];
    const int l = e & 63, r = (e >> 6) & 3, bt = e >> 8;
    const long b = bt * 16 + Mfma<T>::row(l, r);
    const long jo = j0 + (l & 15);
    if (b < Bt && jo < n) out[b * n + jo] = s;
  }
}

// 2 <= Bt <= 128, second form: the P panel is staged through LDS and shared by the four row tiles of a
// workgroup.  The register-operand form above fetches one P value per lane and MFMA from L2 (at
// Bt = 64: 537 MB of L2->L1 traffic against 134 MB of A, the measured limiter); here a workgroup
// covers 64 rows of A x one slice of the contraction index, 8 waves = 4 row tiles x 2 halves of a
// KW-wide k step, P for the step is written to LDS once, already in MFMA operand order
// ([half][b tile][element][lane group][row]: a wave's operand read is 512 contiguous bytes, a
// staging store 128 contiguous bytes per 16 lanes -- both conflict-free), A and the next P step are
// register-prefetched one step ahead.  Slices of the contraction index (grid.y) write their own
// [Bt,n] partial, summed in slice order by skinny_reduce_kernel -- deterministic.  P is read in its
// original [Bt,n] layout (no transpose pass).
constexpr int TLW = 128;  // timeline words per workgroup (ABL 5/6)

// VEC: n % 4 == 0 and 32-byte aligned bases (naturally aligned 32-byte loads); otherwise the same loads carry
// element alignment only and a group of four that straddles n is read element by element.
template <typename T, int NBT, int KW, bool VEC, int ABL = 0>
__global__ __launch_bounds__(512) void symm_skinny_lds_kernel(const T* __restrict__ A, long n,
                                                              const T* __restrict__ P, long Bt,
                                                              T* __restrict__ dst, long kr_len,
                                                              const int* __restrict__ gate, int stagger) {
  if (gate != nullptr && *gate == 0) return;
  // experiment: de-phase the workgroups (all of them otherwise request their tiles at the same instants)
  if (stagger > 0 && stagger < 100) {
    const int ph = (blockIdx.x + blockIdx.y) & 3;
    for (int i = 0; i < ph * stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  constexpr int EH = KW / 2;                     // k per half step
  constexpr int EPL = EH / 4;                    // consecutive k per lane and half step (4 or 8)
  constexpr int PPB = KW / 4;                    // 4-element pieces per row of P and step
  constexpr int STEP = NBT * 16 * KW;            // elements of P per step
  constexpr int NPIECE = NBT * 16 * PPB;         // pieces per step
  constexpr int NV = (NPIECE + 511) / 512;       // pieces staged per thread and step
  constexpr int REDN = NBT * 1024;               // epilogue exchange: 4 tiles x NBT x 4 x 64
  constexpr int LDSN = 2 * STEP > REDN ? 2 * STEP : REDN;
  using Acc = typename Mfma<T>::Acc;
  __shared__ __attribute__((aligned(32))) T Pl[LDSN];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wj = wave & 3, wk = wave >> 2;
  const int jj = lane & 15, g = lane >> 4;
  const long j0 = (long)blockIdx.x * 64 + wj * 16;
  const long j = j0 + jj < n ? j0 + jj : n - 1;
  const T* arow = A + j * n;
  const long k_begin = (long)blockIdx.y * kr_len;
  const long k_end = k_begin + kr_len < n ? k_begin + kr_len : n;
  Acc acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = Acc{0, 0, 0, 0};

  T a[EPL], an[EPL], ps[NV][4];
  auto load4 = [&](const T* base, long k, bool ok, T* v) {
    typedef T V4r __attribute__((ext_vector_type(4)));
    typedef V4r V4 __attribute__((aligned(VEC ? 4 * sizeof(T) : sizeof(T))));
    if (ok && k + 3 < n) {
      const V4r x = *reinterpret_cast<const V4*>(base + k);
      v[0] = x[0];
      v[1] = x[1];
      v[2] = x[2];
      v[3] = x[3];
    } else if (VEC) {
      v[0] = v[1] = v[2] = v[3] = (T)0;  // n % 4 == 0: a group is whole or beyond n
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (ok && k + e < n) ? base[k + e] : (T)0;
    }
  };
  auto load_step = [&](long kr, T* av) {
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) load4(arow, kr + wk * EH + EPL * g + 4 * q, true, av + 4 * q);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = t + 512 * i;
      const int kq = (v >> 4) % PPB, b = ((v >> 4) / PPB) * 16 + (v & 15);
      load4(P + (long)b * n, kr + 4 * kq, b < Bt && v < NPIECE, ps[i]);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = t + 512 * i;
      if (v < NPIECE) {
        const int kq = (v >> 4) % PPB, bt = (v >> 4) / PPB;
        const int kh = kq / (EH / 4), kk0 = 4 * (kq % (EH / 4));
        const int gg = kk0 / EPL, e0 = kk0 % EPL;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          Pl[buf * STEP + ((((kh * NBT + bt) * EPL + e0 + e) * 4 + gg) * 16 + (v & 15))] = ps[i][e];
      }
    }
  };
  // ABL 5: timeline of each workgroup (thread 0: s_memtime at entry, after the prologue, after every step's
  // barrier, at exit; s_memrealtime at entry and exit) into the 48 words per workgroup that the launch reserved
  // behind the partials
  unsigned long long* tl = nullptr;
  int tli = 0;
  if ((ABL == 5 || ABL == 6)) {
    tl = (unsigned long long*)(dst + (long)gridDim.y * Bt * n) + (long)(blockIdx.y * gridDim.x + blockIdx.x) * TLW;
    if (t == 0) {
      tl[0] = __builtin_amdgcn_s_memrealtime();
      tl[1] = __builtin_amdgcn_s_memtime();
    }
    tli = 2;
  }
  dst += (long)blockIdx.y * Bt * n;
  if (k_begin < k_end) {
    load_step(k_begin, a);
    stage(0);
  }
  __syncthreads();
  if ((ABL == 5 || ABL == 6) && t == 0) tl[tli++] = __builtin_amdgcn_s_memtime();
  int buf = 0;
  for (long kr = k_begin; kr < k_end; kr += KW, buf ^= 1) {
    const bool more = kr + KW < k_end;
    // ABL selects ablations for diagnosis (tools/run_skinny.py, MGP_SKINNY_STAGGER=101..104): 1 = no MFMAs,
    // 2 = no A/P loads after the first step, 3 = 2 + no barrier, 4 = 3 + no LDS operand reads
    if (more && (ABL < 2 || (ABL == 5 || ABL == 6) || kr == k_begin)) load_step(kr + KW, an);
    if (ABL == 6 && t == 0) tl[tli++] = __builtin_amdgcn_s_memtime();  // loads issued
    if (ABL != 1) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      T pf[NBT];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) pf[bt] = ABL == 4 ? a[(e + bt) % EPL] : Pl[buf * STEP + ((((wk * NBT + bt) * EPL + e) * 4 + g) * 16 + jj)];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) acc[bt] = Mfma<T>::run(pf[bt], a[e], acc[bt]);
    }
    }
    if (ABL == 6 && t == 0) tl[tli++] = __builtin_amdgcn_s_memtime();  // MFMAs issued
    if (more) {
      stage(buf ^ 1);
#pragma unroll
      for (int e = 0; e < EPL; ++e) a[e] = an[e];
    }
    if (ABL == 6 && t == 0) tl[tli++] = __builtin_amdgcn_s_memtime();  // loads landed, next panel staged
    if (ABL < 3 || (ABL == 5 || ABL == 6)) __syncthreads();
    if ((ABL == 5 || ABL == 6) && t == 0 && tli < TLW - 4) tl[tli++] = __builtin_amdgcn_s_memtime();
  }
  // the two k halves of a row tile meet in LDS (upper half stores, lower half adds and writes)
  T* red = Pl;
  if (wk == 1) {
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wj * NBT + bt) * 4 + r) * 64 + lane] = acc[bt][r];
  }
  __syncthreads();
  if (wk == 0) {
    const long jo = j0 + jj;
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long b = bt * 16 + Mfma<T>::row(lane, r);
        if (b < Bt && jo < n) dst[b * n + jo] = acc[bt][r] + red[((wj * NBT + bt) * 4 + r) * 64 + lane];
      }
  }
  if ((ABL == 5 || ABL == 6) && t == 0) {
    tl[tli++] = __builtin_amdgcn_s_memtime();
    tl[TLW - 2] = (unsigned long long)tli;
    tl[TLW - 1] = __builtin_amdgcn_s_memrealtime();
  }
}

// The same product, software-pipelined (any n >= 256, NBT = 2 or 4).  The timeline of the kernel above
// (MGP_SKINNY_STAGGER=106, profiles/r02_skinny_timeline.txt) showed where its step went: all 8 loads of a thread
// were issued in one burst at the top of the step -- 64 KB per CU, 1024+ cycles of the CU's 64 B/clk vector
// memory path, during which the in-order waves could not issue MFMAs (1500 cycles) -- and came back 5000
// cycles later, after the step's MFMAs had drained (2300 cycles at s_waitcnt vmcnt).  Here
//   * the A fragments are requested TWO steps ahead (HBM under load: ~2.5 us), the P panel one step ahead (L2);
//   * one 32-byte load is issued after every second group of 4 MFMAs instead of 8 in a burst;
//   * no bounds branches in the loop: rows of P beyond Bt and rows of A beyond n are clamped to the last valid
//     row (their results are never written); a partial last k step (n % 64 != 0) is requested element by element
//     with zeros past n, by a second instantiation of the step that only the last three steps of the last slice run;
//     rows are read with element-aligned 32-byte vector loads, so n need not be a multiple of 4;
//   * operand fragments of MFMA group e+1 are read from LDS before group e is issued (explicitly: the
//     scheduling barriers that fix the load positions also stop the compiler from doing it).
// Same lane/element mapping and the same order of accumulation as the kernel above: bit-identical results.
// KW: k per step -- 64 (two staging buffers of NBT <= 4 panels fit 64 KB) or 32 (NBT = 8: Bt up to 128)
template <typename T, int NBT, bool AL, bool RAG, int ABL = 0, int KW = 64>
__global__ __launch_bounds__(512) void symm_skinny_pipe_kernel(const T* __restrict__ A, long n,
                                                               const T* __restrict__ P, long Bt,
                                                               T* __restrict__ dst, long kr_len,
                                                               const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  constexpr int EH = KW / 2, EPL = EH / 4, PPB = KW / 4, NA = EPL / 4;  // NA: 32-byte units of the A fragment
  static_assert(KW == 64 || KW == 32, "k per step");
  constexpr int STEP = NBT * 16 * KW;
  constexpr int NPIECE = NBT * 16 * PPB;
  constexpr int NV = (NPIECE + 511) / 512;
  constexpr int REDN = NBT * 1024;
  constexpr int LDSN = 2 * STEP > REDN ? 2 * STEP : REDN;
  constexpr int NU = NA + NV;  // 32-byte load units per thread and step: NA of A, NV of P
  constexpr int ISTEP = EPL == 8 ? 2 : 1;     // a request behind every ISTEP-th MFMA group
  constexpr int STAGE_AT = EPL == 8 ? 5 : 2;  // the group behind which the next panel is staged
  static_assert(NU * ISTEP <= EPL - 1 + ISTEP && NU <= 4, "requests fit the groups in front of the barrier");
  using Acc = typename Mfma<T>::Acc;
  typedef T V4r __attribute__((ext_vector_type(4)));
  // AL: n % 4 == 0 and 32-byte aligned bases -> naturally aligned 32-byte loads; otherwise rows start at arbitrary
  // multiples of the element size and the loads carry element alignment only
  typedef V4r V4 __attribute__((aligned(AL ? 4 * sizeof(T) : sizeof(T))));
  __shared__ __attribute__((aligned(32))) T Pl[LDSN];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wj = wave & 3, wk = wave >> 2;
  const int jj = lane & 15, g = lane >> 4;
  const long j0 = (long)blockIdx.x * 64 + wj * 16;
  const long j = j0 + jj < n ? j0 + jj : n - 1;
  const long k_begin = (long)blockIdx.y * kr_len;
  const long k_end = k_begin + kr_len < n ? k_begin + kr_len : n;
  const int nsteps = (int)((k_end - k_begin + KW - 1) / KW);
  // n % 64 != 0: the last step of the last slice is partial.  Its operands are loaded element by element with
  // zeros past n (both of them), by the steps that request it; every other step keeps the vector loads.
  // (RAG = false: n % 64 == 0, no such step, and none of its code)
  const int tail = (RAG && ((k_end - k_begin) % KW) != 0) ? nsteps - 1 : -1;
  unsigned long long* tl = nullptr;
  int tli = 0;
  if (ABL == 5) {
    tl = (unsigned long long*)(dst + (long)gridDim.y * Bt * n) + (long)(blockIdx.y * gridDim.x + blockIdx.x) * TLW;
    if (t == 0) {
      tl[0] = __builtin_amdgcn_s_memrealtime();
      tl[1] = __builtin_amdgcn_s_memtime();
    }
    tli = 2;
  }
  dst += (long)blockIdx.y * Bt * n;
  // per-thread source pointers at k_begin; a step advances them by KW elements
  const long ka = k_begin + wk * EH + EPL * g;  // first k of this lane's A fragment in step 0
  const T* ap = A + j * n + ka;
  const T* pp[NV];
  long kp[NV];  // first k of the piece in step 0
  int pdst[NV];  // LDS element offset of the piece's first element within a staging buffer
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    // no predicate anywhere on the staging path (a load whose only use sits in a conditional block is sunk into
    // it by the compiler, next to its use, and its whole latency is exposed): with fewer pieces than threads
    // the surplus threads stage a twin's piece again -- the same values to the same words
    const int vv = (t + 512 * i) % NPIECE;
    const int kq = (vv >> 4) % PPB, bt = (vv >> 4) / PPB;
    long b = bt * 16 + (vv & 15);
    b = b < Bt ? b : Bt - 1;
    kp[i] = k_begin + 4 * kq;
    pp[i] = P + b * n + kp[i];
    const int kh = kq / (EH / 4), kk0 = 4 * (kq % (EH / 4));
    const int gg = kk0 / EPL, e0 = kk0 % EPL;
    pdst[i] = (((kh * NBT + bt) * EPL + e0) * 4 + gg) * 16 + (vv & 15);
  }
  Acc acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = Acc{0, 0, 0, 0};
  V4 a0[NA], a1[NA], a2[NA], ps[NV];
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Pl[buf * STEP + pdst[i] + e * 64] = ps[i][e];
    }
  };
  // unit u of step `st`: u < NV -> piece u of the P panel, else half (u - NV) of the A fragment
  auto load_vec = [&](int u, int st, V4 (&av)[NA]) {
    const long o = (long)st * KW;
    if (u < NV) ps[u < NV ? u : 0] = *reinterpret_cast<const V4*>(pp[u < NV ? u : 0] + o);
    else av[u - NV] = *reinterpret_cast<const V4*>(ap + o + 4 * (u - NV));
  };
  auto load_gen = [&](int u, int st, V4 (&av)[NA]) {
    if (!RAG || st != tail) {
      load_vec(u, st, av);
    } else {
      const long o = (long)st * KW;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (u < NV) {
          const int i = u < NV ? u : 0;
          ps[i][e] = kp[i] + o + e < n ? pp[i][o + e] : (T)0;
        } else {
          const int q = u - NV;
          av[q][e] = ka + o + 4 * q + e < n ? ap[o + 4 * q + e] : (T)0;
        }
      }
    }
  };
  if (nsteps > 0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) load_gen(u, 0, a0);
#pragma unroll
    for (int q = 0; q < NA; ++q) load_gen(NV + q, 0, a0);
    const int s1 = nsteps > 1 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < NA; ++q) load_gen(NV + q, s1, a1);
    stage(0);
  }
  __syncthreads();
  if (ABL == 5 && t == 0) tl[tli++] = __builtin_amdgcn_s_memtime();
  const T* lds_rd = Pl + (wk * NBT * EPL * 4 + g) * 16 + jj;  // + (bt * EPL + e) * 64 per operand
  // one k step: MFMAs on `ac` (the A fragments of step s) against LDS buffer s & 1; requests the P panel of step
  // s+1 (staged at the end of this step) and the A fragments of step s+2 into `ain` (clamped to the last step:
  // the few repeated requests at the end hit in cache).  The three fragment sets rotate by NAME over a loop
  // unrolled by three -- a register copy of a set still in flight would wait for it.
  // The step's barrier sits BEFORE its last MFMA group: that group's operands are read ahead of the barrier,
  // the first group of the next step is read right after it, and the four MFMAs in between cover the LDS latency
  // (a barrier after the last MFMA left the matrix pipe idle for staging + barrier + read latency, ~700 of a
  // step's 4800 cycles).  The panel of the next step is staged after group 5, by when its loads (requested after
  // groups 0 and 2) have landed.
  T pf[2][NBT];
  auto do_step = [&](auto gen_tag, int s, const V4 (&ac)[NA], V4 (&ain)[NA]) {
    constexpr bool GEN = decltype(gen_tag)::value;  // this step may request the partial step: checked per unit
    const int buf = s & 1;
    const int sp = s + 1 < nsteps ? s + 1 : nsteps - 1;
    const int sa = s + 2 < nsteps ? s + 2 : nsteps - 1;
    auto issue = [&](int u) {
      if (u >= NU) return;
      if (GEN) load_gen(u, u < NV ? sp : sa, ain);
      else load_vec(u, u < NV ? sp : sa, ain);
    };
    const T* rd = lds_rd + buf * STEP;
#pragma unroll
    for (int e = 0; e < EPL - 1; ++e) {
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) pf[(e + 1) & 1][bt] = rd[(bt * EPL + e + 1) * 64];
      const T av = ac[e >> 2][e & 3];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) acc[bt] = Mfma<T>::run(pf[e & 1][bt], av, acc[bt]);
      // NU <= 4 units over the groups in front of the barrier, the last groups free of requests
      if (e % ISTEP == 0) issue(e / ISTEP);
      if (e == STAGE_AT) stage(buf ^ 1);  // unconditional (see above); after the last step that buffer is not read again
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (ABL == 5 && t == 0 && tli < TLW - 4) tl[tli++] = __builtin_amdgcn_s_memtime();
    const T* rn = lds_rd + (buf ^ 1) * STEP;
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) pf[0][bt] = rn[(bt * EPL) * 64];
    {
      const T av = ac[(EPL - 1) >> 2][(EPL - 1) & 3];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) acc[bt] = Mfma<T>::run(pf[1][bt], av, acc[bt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) pf[0][bt] = lds_rd[(bt * EPL) * 64];  // group 0 of step 0 (buffer 0)
  // steps from gen_from on request the partial step (or, clamped, re-request it)
  const int gen_from = tail < 0 ? nsteps : (tail - 2 > 0 ? tail - 2 : 0);
  auto step_any = [&](int s, const V4 (&ac)[NA], V4 (&ain)[NA]) {
    if (!RAG || s < gen_from) do_step(std::false_type{}, s, ac, ain);
    else do_step(std::true_type{}, s, ac, ain);
  };
  for (int s = 0; s < nsteps; s += 3) {
    step_any(s, a0, a2);
    if (s + 1 < nsteps) step_any(s + 1, a1, a0);
    if (s + 2 < nsteps) step_any(s + 2, a2, a1);
  }
  T* red = Pl;
  if (wk == 1) {
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wj * NBT + bt) * 4 + r) * 64 + lane] = acc[bt][r];
  }
  __syncthreads();
  if (wk == 0) {
    const long jo = j0 + jj;
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long b = bt * 16 + Mfma<T>::row(lane, r);
        if (b < Bt && jo < n) dst[b * n + jo] = acc[bt][r] + red[((wj * NBT + bt) * 4 + r) * 64 + lane];
      }
  }
  if (ABL == 5 && t == 0) {
    tl[tli++] = __builtin_amdgcn_s_memtime();
    tl[TLW - 2] = (unsigned long long)tli;
    tl[TLW - 1] = __builtin_amdgcn_s_memrealtime();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const T* __restrict__ part, long tot, int ks,
                                                            T* __restrict__ out, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= tot) return;
  T s = part[e];
  for (int z = 1; z < ks; ++z) s += part[(long)z * tot + e];
  out[e] = s;
}

template <typename T, int NBT>
int symm_skinny_lds_launch(mgp_handle* h, const T* A, long n, const T* P, long Bt, T* out, const int* gate) {
  constexpr int KW = (NBT * 16 * 64 * 2 * sizeof(T) <= 65536) ? 64 : 32;  // two staging buffers within 64 KB
  const long jg = (n + 63) / 64;
  // workgroups per CU: two for the narrow panels (Bt <= 32: 28-30 us instead of 33-35 at n = 4096), one for
  // the wide ones, where a second resident workgroup only adds slice partials (measured, MGP_SKINNY_BPC)
  // (the pipelined kernel, 16 < Bt <= 64, prefers one as well since round 2: Bt = 32, n = 4096: 33.6 vs 36.9 us)
  const long bpc = h->skinny_blocks_per_cu > 0 ? h->skinny_blocks_per_cu : (NBT <= 1 ? 2 : 1);
  // ONE resident round: the slice count is rounded down (n = 4032, Bt = 64: 63 row blocks x 4 slices = 252
  // workgroups, not 5 slices = 315 in two rounds: 71.8 -> 50.6 us; n = 4001, Bt = 8: 9 -> 8 slices)
  long ks = bpc * h->num_cus / jg;
  if (ks > 16) ks = 16;
  if (ks < 1) ks = 1;
  long kr_len = ((n + ks - 1) / ks + KW - 1) / KW * KW;
  ks = (n + kr_len - 1) / kr_len;
  T* dst = out;
  if (ks > 1) {
    MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)ks * Bt * n * sizeof(T) + (size_t)jg * ks * TLW * 8));
    dst = (T*)h->ws;
  }
  const bool vec = (n % 4) == 0 && (((uintptr_t)A) % 32) == 0 && (((uintptr_t)P) % 32) == 0;
  dim3 grid((unsigned)jg, (unsigned)ks);
  bool piped = false;
  if constexpr (NBT == 8) {  // 64 < Bt <= 128: the same pipeline over 32-wide k steps (two panels of 8 tiles in 64 KB)
    if (n >= 256 && h->skinny_pipe && h->skinny_stagger <= 100) {
      piped = true;
      const bool whole = (n % 32) == 0;
      if (vec && whole)
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, true, false, 0, 32>), grid, dim3(512), 0, h->stream, A, n, P,
                           Bt, dst, kr_len, gate);
      else if (vec)
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, true, true, 0, 32>), grid, dim3(512), 0, h->stream, A, n, P,
                           Bt, dst, kr_len, gate);
      else
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, false, true, 0, 32>), grid, dim3(512), 0, h->stream, A, n, P,
                           Bt, dst, kr_len, gate);
    }
  }
  if constexpr (NBT == 2 || NBT == 4) {  // Bt <= 16 is bound by the A stream and the round-1 form already runs at it (29.7 vs 34.2 us)
    if (n >= 256 && h->skinny_pipe && (h->skinny_stagger <= 100 || h->skinny_stagger == 107)) {
      piped = true;
      const bool whole = (n % 64) == 0;
      if (h->skinny_stagger == 107 && vec && whole)
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, true, false, 5>), grid, dim3(512), 0, h->stream, A, n, P, Bt,
                           dst, kr_len, gate);
      else if (vec && whole)
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, true, false>), grid, dim3(512), 0, h->stream, A, n, P, Bt,
                           dst, kr_len, gate);
      else if (vec)
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, true, true>), grid, dim3(512), 0, h->stream, A, n, P, Bt,
                           dst, kr_len, gate);
      else
        hipLaunchKernelGGL((symm_skinny_pipe_kernel<T, NBT, false, true>), grid, dim3(512), 0, h->stream, A, n, P, Bt,
                           dst, kr_len, gate);
    }
  }
  if (piped) {
  } else if (vec && h->skinny_stagger > 100 && NBT == 4 && std::is_same<T, double>::value) {  // diagnosis only
#define MGP_SK_ABL(V)                                                                                              \
  hipLaunchKernelGGL((symm_skinny_lds_kernel<T, NBT, KW, true, V>), grid, dim3(512), 0, h->stream, A, n, P, Bt, dst, \
                     kr_len, gate, 0)
    if (h->skinny_stagger == 101) MGP_SK_ABL(1);
    else if (h->skinny_stagger == 102) MGP_SK_ABL(2);
    else if (h->skinny_stagger == 103) MGP_SK_ABL(3);
    else if (h->skinny_stagger == 104) MGP_SK_ABL(4);
    else if (ks > 1) {
      if (h->skinny_stagger == 105) MGP_SK_ABL(5);
      else MGP_SK_ABL(6);
    }
#undef MGP_SK_ABL
  } else if (vec)
    hipLaunchKernelGGL((symm_skinny_lds_kernel<T, NBT, KW, true>), grid, dim3(512), 0, h->stream, A, n, P, Bt, dst,
                       kr_len, gate, h->skinny_stagger);
  else
    hipLaunchKernelGGL((symm_skinny_lds_kernel<T, NBT, KW, false>), grid, dim3(512), 0, h->stream, A, n, P, Bt, dst,
                       kr_len, gate, h->skinny_stagger);
  MGP_LAUNCH_CHECK(h);
  if (h->skinny_stagger >= 105 && h->skinny_stagger <= 107 && ks > 1 && NBT == 4) {
    {
      static int dumps = 0;
      if (dumps++ == 4) {  // the fifth call (warm): per-phase s_memtime deltas, median / max over workgroups
        MGP_HIP(h, hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> tl((size_t)jg * ks * TLW);
        MGP_HIP(h, hipMemcpy(tl.data(), (const char*)dst + (size_t)ks * Bt * n * sizeof(T), tl.size() * 8,
                             hipMemcpyDeviceToHost));
        const size_t nw = (size_t)jg * ks;
        unsigned long long r0 = ~0ULL, r1 = 0;
        for (size_t w = 0; w < nw; ++w) {
          r0 = tl[w * TLW] < r0 ? tl[w * TLW] : r0;
          r1 = tl[w * TLW + TLW - 1] > r1 ? tl[w * TLW + TLW - 1] : r1;
        }
        fprintf(stderr, "skinny timeline: %zu workgroups, first entry -> last exit %.2f us (s_memrealtime, 100 MHz)\n", nw,
                (double)(r1 - r0) / 100.0);
        const int cnt = (int)tl[TLW - 2];
        for (int i = 1; i < cnt; ++i) {
          std::vector<double> d;
          for (size_t w = 0; w < nw; ++w) d.push_back((double)(tl[w * TLW + i] - tl[w * TLW + (i == 1 ? 0 : i - 1)]));
          std::sort(d.begin(), d.end());
          if (i == 1) continue;
          fprintf(stderr, "  phase %2d: median %8.0f  min %8.0f  max %8.0f ticks\n", i - 2, d[nw / 2], d[0], d[nw - 1]);
        }
        std::vector<double> e, x;
        for (size_t w = 0; w < nw; ++w) {
          e.push_back((double)(tl[w * TLW] - r0) / 100.0);
          x.push_back((double)(tl[w * TLW + TLW - 1] - tl[w * TLW]) / 100.0);
        }
        std::sort(e.begin(), e.end());
        std::sort(x.begin(), x.end());
        fprintf(stderr, "  entry offset us: median %.2f max %.2f; workgroup lifetime us: median %.2f min %.2f max %.2f\n",
                e[nw / 2], e[nw - 1], x[nw / 2], x[0], x[nw - 1]);
      }
    }
  }
  if (ks > 1 && ks <= 8 && h->defer_slices) {  // the caller (the fused CG update) sums the slices as it reads them
    h->deferred_part = dst;
    h->deferred_ks = (int)ks;
    h->deferred_stride = Bt * n;
    return MGP_OK;
  }
  if (ks > 1) {
    const long tot = Bt * n;
    hipLaunchKernelGGL((skinny_reduce_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const T*)dst, tot, (int)ks, out, gate);
    MGP_LAUNCH_CHECK(h);
  }
  return MGP_OK;
}

// slots Q[nt][n] in h->ws and the tile table of the one-RHS upper-triangle product
int tri_prepare(mgp_handle* h, size_t esize, long n, int form) {
  const int nt = (int)((n + 63) / 64);
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nt * n * esize));
  if (form != 0 && h->tri_tab_nt != nt) {
    const long ntiles = (long)nt * (nt + 1) / 2;
    MGP_TRY(mgp_reserve(h, &h->tri_tab, &h->tri_tab_bytes, (size_t)ntiles * sizeof(int2)));
    hipLaunchKernelGGL(tri_tile_table_kernel, dim3((unsigned)nt), dim3(64), 0, h->stream, (int2*)h->tri_tab, nt);
    MGP_LAUNCH_CHECK(h);
    h->tri_tab_nt = nt;
  }
  return MGP_OK;
}

// the tile kernel of the one-RHS upper-triangle product: slots Q[nt][n] left in h->ws
template <typename T>
int symm_gemv_tri_slots_t(mgp_handle* h, const T* A, long n, const T* P, const int* gate, int* nt_out) {
  const int nt = (int)((n + 63) / 64);
  const long ntiles = (long)nt * (nt + 1) / 2;
  const int form = nt <= 512 ? h->tri_form : 0;  // the tile table is kept for n <= 32768 (1 MB)
  MGP_TRY(tri_prepare(h, sizeof(T), n, form));
  if (form != 0)
    hipLaunchKernelGGL((symm_gemv_tri_bfly_kernel<T>), dim3((unsigned)ntiles), dim3(256), 0, h->stream, A, n, P,
                       (T*)h->ws, (const int2*)h->tri_tab, gate);
  else
    hipLaunchKernelGGL((symm_gemv_tri_kernel<T>), dim3((unsigned)ntiles), dim3(256), 0, h->stream, A, n, P,
                       (T*)h->ws, nt, gate);
  MGP_LAUNCH_CHECK(h);
  *nt_out = nt;
  return MGP_OK;
}

template <typename T>
int symm_matmul_t(mgp_handle* h, const T* A, long n, const T* P, long Bt, T* out, const int* gate) {
  if (Bt >= 2 && Bt <= 128) {
    const int nbt = Bt <= 16 ? 1 : (Bt <= 32 ? 2 : (Bt <= 64 ? 4 : 8));
    if (h->skinny_mode == 1) {  // P staged through LDS (default); MGP_SKINNY=reg selects the form below
      if (nbt == 1) return symm_skinny_lds_launch<T, 1>(h, A, n, P, Bt, out, gate);
      if (nbt == 2) return symm_skinny_lds_launch<T, 2>(h, A, n, P, Bt, out, gate);
      if (nbt == 4) return symm_skinny_lds_launch<T, 4>(h, A, n, P, Bt, out, gate);
      return symm_skinny_lds_launch<T, 8>(h, A, n, P, Bt, out, gate);
    }
    const int BP = 16 * nbt;
    MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)n * BP * sizeof(T)));
    T* Pt = (T*)h->ws;
    const long tot = n * BP;
    hipLaunchKernelGGL((transpose_pad_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, P,
                       Bt, n, Pt, BP, gate);
    MGP_LAUNCH_CHECK(h);
    dim3 grid((unsigned)((n + 15) / 16));
#define MGP_SK(NBTV) \
  hipLaunchKernelGGL((symm_skinny_kernel<T, NBTV>), grid, dim3(512), 0, h->stream, A, n, (const T*)Pt, Bt, out, gate)
    if (nbt == 1) MGP_SK(1);
    else if (nbt == 2) MGP_SK(2);
    else if (nbt == 4) MGP_SK(4);
    else MGP_SK(8);
#undef MGP_SK
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  if (Bt > 128) {
    return gemm_nt_launch<T>(h, P, n, Bt, A, n, n, n, out, n, 0, gate);
  }
  if (n >= h->tri_min_n && n <= 64L * 32768) {
    int nt = 0;
    MGP_TRY(symm_gemv_tri_slots_t<T>(h, A, n, P, gate, &nt));
    hipLaunchKernelGGL((symm_gemv_tri_reduce_kernel<T>), dim3((unsigned)nt), dim3(256), 0, h->stream,
                       (const T*)h->ws, n, nt, out, gate);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  constexpr int VECW = 16 / sizeof(T);
  const bool vec = (n % VECW) == 0 && (((uintptr_t)A) % 16) == 0;
  dim3 grid((unsigned)((n + 7) / 8));
#define MGP_GV(BTV)                                                                                             \
  do {                                                                                                          \
    if (vec)                                                                                                    \
      hipLaunchKernelGGL((symm_gemv_kernel<T, BTV, VECW>), grid, dim3(256), 0, h->stream, A, n, P, (int)Bt, out, \
                         gate, 0L, n, (T)1, 0);                                                                 \
    else                                                                                                        \
      hipLaunchKernelGGL((symm_gemv_kernel<T, BTV, 1>), grid, dim3(256), 0, h->stream, A, n, P, (int)Bt, out,    \
                         gate, 0L, n, (T)1, 0);                                                                 \
  } while (0)
  MGP_GV(1);  // Bt == 1 here: 2..128 took the skinny path, larger the GEMM
#undef MGP_GV
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

}  // namespace

// out[row] += alpha * A[row,:].p for rows [row_begin,row_end) (one RHS): a rank's slab of the
// replicated s2*Kmm.p term of the SGPR operator, added into its partial before the all-reduce
template <typename T>
int symm_gemv_rows_t(mgp_handle* h, const T* A, long n, const T* p, long rb, long re, T alpha, T* out,
                     const int* gate, T* word) {
  if (re <= rb) return MGP_OK;
  constexpr int VECW = 16 / sizeof(T);
  const bool vec = (n % VECW) == 0 && (((uintptr_t)A) % 16) == 0;
  if ((re - rb) <= 2L * h->num_cus * 4 && n >= 1024) {  // a rank's slab: columns split over the waves of a workgroup
    dim3 grid((unsigned)((re - rb + 1) / 2));
    if (vec && (n % (64 * VECW)) == 0)
      hipLaunchKernelGGL((symm_gemv_slab_kernel<T, VECW>), grid, dim3(256), 0, h->stream, A, n, p, out, gate, rb, re,
                         alpha, word);
    else
      hipLaunchKernelGGL((symm_gemv_slab_kernel<T, 1>), grid, dim3(256), 0, h->stream, A, n, p, out, gate, rb, re,
                         alpha, word);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  dim3 grid((unsigned)((re - rb + 7) / 8));
  if (vec)
    hipLaunchKernelGGL((symm_gemv_kernel<T, 1, VECW>), grid, dim3(256), 0, h->stream, A, n, p, 1, out, gate, rb, re,
                       alpha, 1, word);
  else
    hipLaunchKernelGGL((symm_gemv_kernel<T, 1, 1>), grid, dim3(256), 0, h->stream, A, n, p, 1, out, gate, rb, re,
                       alpha, 1, word);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

// out = A p, every row, one right-hand side, nothing accumulated (the SGPR operator's Kmm.p beside the sweep when this
// rank owns every row: no memset of the target before it)
int mgp_symm_gemv_assign(mgp_handle* h, int dtype, const void* A, int64_t n, const void* p, void* out, const int* gate) {
  dim3 grid((unsigned)((n + 7) / 8));
  const bool vec16 = (((uintptr_t)A) % 16) == 0;
  if (dtype == MGP_F64) {
    if (vec16 && n % 2 == 0)
      hipLaunchKernelGGL((symm_gemv_kernel<double, 1, 2>), grid, dim3(256), 0, h->stream, (const double*)A, (long)n,
                         (const double*)p, 1, (double*)out, gate, 0L, (long)n, 1.0, 0, (double*)nullptr);
    else
      hipLaunchKernelGGL((symm_gemv_kernel<double, 1, 1>), grid, dim3(256), 0, h->stream, (const double*)A, (long)n,
                         (const double*)p, 1, (double*)out, gate, 0L, (long)n, 1.0, 0, (double*)nullptr);
  } else {
    if (vec16 && n % 4 == 0)
      hipLaunchKernelGGL((symm_gemv_kernel<float, 1, 4>), grid, dim3(256), 0, h->stream, (const float*)A, (long)n,
                         (const float*)p, 1, (float*)out, gate, 0L, (long)n, 1.0f, 0, (float*)nullptr);
    else
      hipLaunchKernelGGL((symm_gemv_kernel<float, 1, 1>), grid, dim3(256), 0, h->stream, (const float*)A, (long)n,
                         (const float*)p, 1, (float*)out, gate, 0L, (long)n, 1.0f, 0, (float*)nullptr);
  }
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

int mgp_symm_gemv_tri_prepare(mgp_handle* h, int dtype, int64_t n, void** Q, const void** tab) {
  MGP_TRY(tri_prepare(h, mgp_elem(dtype), n, 1));
  *Q = h->ws;
  *tab = h->tri_tab;
  return MGP_OK;
}

int mgp_symm_gemv_rows_acc(mgp_handle* h, int dtype, const void* A, int64_t n, const void* p, int64_t rb, int64_t re,
                           double alpha, void* out, const int* gate, void* word) {
  if (dtype == MGP_F64)
    return symm_gemv_rows_t<double>(h, (const double*)A, n, (const double*)p, rb, re, alpha, (double*)out, gate,
                                    (double*)word);
  return symm_gemv_rows_t<float>(h, (const float*)A, n, (const float*)p, rb, re, (float)alpha, (float*)out, gate,
                                 (float*)word);
}

// out[n, n] (+)= Kt[n, K] . Kt[n, K]^T on upper-triangular tiles (contract.hip accumulates row chunks)
int mgp_syrk_nt_upper(mgp_handle* h, int dtype, const void* Kt, int64_t n, int64_t K, int64_t ld, void* out,
                      int accumulate, int nz, const int* tile_tab, int ntiles) {
  // nz slices of K columns each (Kt row = nz*K columns wide, ld), slice z accumulates into out + z*n*n;
  // tile_tab lists the upper-triangular 128x128 tiles (device, [ntiles][2])
  dim3 grid((unsigned)(ntiles * nz));
  const long zk = K, zo = n * n;
  const int upper = 1;
#define MGP_SYRK(TT, VV)                                                                                      \
  hipLaunchKernelGGL((gemm_nt_kernel<TT, VV, 4>), grid, dim3(256), 0, h->stream, (const TT*)Kt, ld, n, (const TT*)Kt, \
                     ld, n, K, (TT*)out, n, accumulate, upper, (const int*)nullptr, zk, zo, tile_tab, ntiles)
  if (dtype == MGP_F64) {
    if (gemm_vec_ok<double>((const double*)Kt, ld, (const double*)Kt, ld, K)) MGP_SYRK(double, true);
    else MGP_SYRK(double, false);
  } else {
    if (gemm_vec_ok<float>((const float*)Kt, ld, (const float*)Kt, ld, K)) MGP_SYRK(float, true);
    else MGP_SYRK(float, false);
  }
#undef MGP_SYRK
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

int mgp_gemm_nt(mgp_handle* h, int dtype, const void* P, int64_t ldp, int64_t m, const void* A, int64_t lda, int64_t n,
                int64_t K, void* out, int64_t ldo, int accumulate, const int* gate) {
  if (dtype == MGP_F64)
    return gemm_nt_launch<double>(h, (const double*)P, ldp, m, (const double*)A, lda, n, K, (double*)out, ldo,
                                  accumulate, gate);
  return gemm_nt_launch<float>(h, (const float*)P, ldp, m, (const float*)A, lda, n, K, (float*)out, ldo, accumulate,
                               gate);
}

int mgp_mirror_upper(mgp_handle* h, int dtype, void* out, const void* slices, int nz, int64_t n, double scale) {
  const long tot = n * n;
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((mirror_upper_kernel<double>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (double*)out, (const double*)slices, nz, n, scale);
  else
    hipLaunchKernelGGL((mirror_upper_kernel<float>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (float*)out, (const float*)slices, nz, n, (float)scale);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

int mgp_symm_matmul_gated(mgp_handle* h, int dtype, const void* A, int64_t n, const void* P, int64_t Bt,
                          void* out, const int* gate) {
  if (!h) return MGP_E_BADARG;
  if (dtype != MGP_F32 && dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", dtype);
  if (n < 0 || Bt < 0) return mgp_fail(h, MGP_E_SHAPE, "negative size");
  if (n == 0 || Bt == 0) return MGP_OK;
  if (!A || !P || !out) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (dtype == MGP_F64) return symm_matmul_t<double>(h, (const double*)A, n, (const double*)P, Bt, (double*)out, gate);
  return symm_matmul_t<float>(h, (const float*)A, n, (const float*)P, Bt, (float*)out, gate);
}

extern "C" int mgp_symm_matmul(mgp_handle* h, int dtype, const void* A, int64_t n, const void* P, int64_t Bt,
                               void* out) {
  return mgp_symm_matmul_gated(h, dtype, A, n, P, Bt, out, nullptr);
}

extern "C" int mgp_k_dense(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B,
                           int64_t nb, void* out, int64_t ld, double jitter, const void* diag_add) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (na < 0 || nb < 0 || ld < nb) return mgp_fail(h, MGP_E_SHAPE, "k_dense: bad shape na=%ld nb=%ld ld=%ld",
                                                   (long)na, (long)nb, (long)ld);
  if (na == 0 || nb == 0) return MGP_OK;
  if (!A || !B || !out) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (k->D > MGP_FUSED_MAX_D) return mgp_k_dense_generic(h, k, A, na, B, nb, out, ld, jitter, diag_add, nullptr);
  if (k->dtype == MGP_F64)
    return k_dense_t<double>(h, k, (const double*)A, na, (const double*)B, nb, (double*)out, ld, jitter,
                             (const double*)diag_add);
  return k_dense_t<float>(h, k, (const float*)A, na, (const float*)B, nb, (float*)out, ld, jitter,
                          (const float*)diag_add);
}
