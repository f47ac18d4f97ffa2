// sweep_mfma.hip -- fp64 fused kernel products with the distance cross-term on the matrix cores.
//
// Same contract as sweep.hip:  out(i,r) = variance * sum_j k(a_i, b_j) w(j,r) [+ alpha*addend].
// The VALU version spends D+1 of its ~27 fp64 instructions per pair on the squared distance.
// Here a 16x16 tile of  -s_ij = 2 a_i.b_j - |a_i|^2 - |b_j|^2  comes out of
// v_mfma_f64_16x16x4_f64 directly, by augmenting the K dimension:
//     A row (owned point)    = [ a_1 .. a_D,  |a|^2,  1,      0.. ]
//     B col (streamed point) = [2b_1 .. 2b_D, -1,    -|b|^2,  0.. ]      K = 4*KS >= D + 2
// The matrix pipe runs beside the VALU, which is left with only the profile (exp2 polynomial)
// and the RC accumulate fmas.  C/D layout of the f64 MFMA: lane l holds column (l & 15) and rows
// (l >> 4) + 4 r, r = 0..3 -- rows are owned points (4 accumulators per lane and tile), the
// column is the streamed point, whose multiplier w_j is one LDS word per lane.  The 16 lanes
// that share a row are summed once, at the very end, by xor-shuffles.
//
// LDS holds the streamed tile already in MFMA fragment order ([j-tile][k-step][lane]), so a
// fragment is one conflict-free ds_read_b64 per lane.
#include "mgp_common.h"

namespace {

using Acc4 = __attribute__((ext_vector_type(4))) double;

constexpr int kThreadsM = 256;
constexpr int kTBJ = 256;  // streamed points per LDS tile (16 j-tiles)

template <int RC>
struct MfmaCfg {
  static constexpr int TR = RC <= 2 ? 4 : 2;  // 16-row tiles owned by one wave
};

template <int KIND, int KS, int RC>
__global__ __launch_bounds__(kThreadsM) void sweep_mfma_kernel(
    const double* __restrict__ A, long na, const double* __restrict__ B, long nb, long b_chunk,
    const double* __restrict__ W, long w_sj, long w_sr, double* __restrict__ out, long o_si, long o_sr,
    long o_chunk, int D, SweepParams prm, double alpha, const double* __restrict__ addend, long ad_si,
    long ad_sr, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  constexpr int TR = MfmaCfg<RC>::TR;
  constexpr int K4 = 4 * KS;
  constexpr int OWN = 64 * TR;  // owned points per block
  // streamed tile: fragments [16 j-tiles][KS][64 lanes] + multipliers [256][RC]
  __shared__ __attribute__((aligned(16))) double frag[16 * KS * 64];
  __shared__ __attribute__((aligned(16))) double wl[kTBJ * RC];
  __shared__ double e2tab[MGP_EXP2_TAB_SIZE];  // 2^(i/2048) for mgp_exp2_tab
  // owned points are staged through the same fragment buffer in the prologue
  static_assert(16 * KS * 64 >= OWN * K4, "prologue staging must fit in the fragment buffer");

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long base = (long)blockIdx.x * OWN;
  for (int e = t; e < MGP_EXP2_TAB_SIZE; e += kThreadsM) e2tab[e] = mgp_exp2_tab_entry(e);
  const E2Tab<true> e2{e2tab};

  // ---- prologue: owned points -> augmented vectors in LDS -> A fragments in registers
  // (OWN * K4 doubles; OWN = 64*TR <= 256, K4 <= 36: fits in frag[] because 16*64 >= OWN)
  for (int p = t; p < OWN; p += kThreadsM) {
    long i = base + p;
    if (i >= na) i = na - 1;
    double s = 0;
    double* dst = &frag[p * K4];
    for (int d = 0; d < D; ++d) {
      const double v = A[i * D + d] * prm.inv_ls[d];
      dst[d] = v;
      s = mgp_fma(v, v, s);
    }
    dst[D] = s;
    dst[D + 1] = 1.0;
    for (int d = D + 2; d < K4; ++d) dst[d] = 0.0;
  }
  __syncthreads();
  double af[TR][KS];
#pragma unroll
  for (int tr = 0; tr < TR; ++tr)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      af[tr][ks] = frag[((wave * TR + tr) * 16 + (lane & 15)) * K4 + 4 * ks + (lane >> 4)];

  double acc[TR][4][RC];
#pragma unroll
  for (int tr = 0; tr < TR; ++tr)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
      for (int r = 0; r < RC; ++r) acc[tr][r4][r] = 0.0;

  const long jb = (long)blockIdx.y * b_chunk;
  const long je = (jb + b_chunk < nb) ? jb + b_chunk : nb;
  const double clamp = prm.clamp;

  for (long j0 = jb; j0 < je; j0 += kTBJ) {
    __syncthreads();  // previous tile (or the prologue) fully consumed
    {
      const long j = j0 + t;
      const int jt = t >> 4, jj = t & 15;
      double* fb = &frag[jt * KS * 64 + jj];
      if (j < je) {
        double s = 0;
        for (int d = 0; d < D; ++d) {
          const double v = B[j * D + d] * prm.inv_ls[d];
          s = mgp_fma(v, v, s);
          fb[(d >> 2) * 64 + (d & 3) * 16] = v + v;
        }
        fb[(D >> 2) * 64 + (D & 3) * 16] = -1.0;
        fb[((D + 1) >> 2) * 64 + ((D + 1) & 3) * 16] = -s;
        for (int d = D + 2; d < K4; ++d) fb[(d >> 2) * 64 + (d & 3) * 16] = 0.0;
#pragma unroll
        for (int r = 0; r < RC; ++r) wl[t * RC + r] = W[j * w_sj + r * w_sr];
      } else {
        for (int d = 0; d < K4; ++d) fb[(d >> 2) * 64 + (d & 3) * 16] = 0.0;
#pragma unroll
        for (int r = 0; r < RC; ++r) wl[t * RC + r] = 0.0;
      }
    }
    __syncthreads();

#pragma unroll 2
    for (int jt = 0; jt < 16; ++jt) {
      double bf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bf[ks] = frag[(jt * KS + ks) * 64 + lane];
      double w[RC];
#pragma unroll
      for (int r = 0; r < RC; ++r) w[r] = wl[(jt * 16 + (lane & 15)) * RC + r];
      Acc4 c[TR];
#pragma unroll
      for (int tr = 0; tr < TR; ++tr) c[tr] = Acc4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tr = 0; tr < TR; ++tr)
          c[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[tr][ks], bf[ks], c[tr], 0, 0, 0);
#pragma unroll
      for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const double kv = mgp_profile<KIND, double, E2Tab<true>>(c[tr][r4], clamp, e2);
#pragma unroll
          for (int r = 0; r < RC; ++r) acc[tr][r4][r] = mgp_fma(kv, w[r], acc[tr][r4][r]);
        }
    }
  }

  // ---- epilogue: sum the 16 lanes (columns) of every owned row, lane (l&15)==0 stores
  const double var = prm.variance;
  double* o = out + (long)blockIdx.y * o_chunk;
#pragma unroll
  for (int tr = 0; tr < TR; ++tr)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        double v = acc[tr][r4][r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        const long i = base + (wave * TR + tr) * 16 + (lane >> 4) + 4 * r4;
        if ((lane & 15) == 0 && i < na) {
          v *= var;
          if (addend != nullptr) v = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], v);
          o[i * o_si + r * o_sr] = v;
        }
      }
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_m_kernel(const T* __restrict__ part, long na, int R,
                                                                int nchunks, T* __restrict__ out, long o_si,
                                                                long o_sr, T alpha, const T* __restrict__ addend,
                                                                long ad_si, long ad_sr,
                                                                const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= na * R) return;
  const long r = idx / na, i = idx - r * na;
  T s = 0;
  const long stride = na * R;
  for (int c = 0; c < nchunks; ++c) s += part[(long)c * stride + idx];
  if (addend != nullptr) s = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], s);
  out[i * o_si + r * o_sr] = s;
}

template <int KIND, int KS, int RC>
int launch_mfma(mgp_handle* h, const SweepParams& prm, int D, const double* A, long na, const double* B, long nb,
                const double* W, long w_sj, long w_sr, double* out, long o_si, long o_sr, double alpha,
                const double* addend, long ad_si, long ad_sr, const int* gate) {
  constexpr int OWN = 64 * MfmaCfg<RC>::TR;
  const long nblk = (na + OWN - 1) / OWN;
  const long target = 8L * h->num_cus;
  long nchunks = nblk >= 4L * h->num_cus ? 1 : (target + nblk - 1) / nblk;
  const long max_chunks = (nb + kTBJ - 1) / kTBJ;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  long b_chunk = (nb + nchunks - 1) / nchunks;
  b_chunk = (b_chunk + kTBJ - 1) / kTBJ * kTBJ;
  nchunks = (nb + b_chunk - 1) / b_chunk;
  if (nchunks > 65535) return mgp_fail(h, MGP_E_SHAPE, "sweep: too many chunks");
  dim3 grid((unsigned)nblk, (unsigned)nchunks);
  if (nchunks == 1) {
    hipEvent_t stop = mgp_prof_begin(h);
    hipLaunchKernelGGL((sweep_mfma_kernel<KIND, KS, RC>), grid, dim3(kThreadsM), 0, h->stream, A, na, B, nb, b_chunk,
                       W, w_sj, w_sr, out, o_si, o_sr, 0L, D, prm, alpha, addend, ad_si, ad_sr, gate);
    mgp_prof_end(h, stop);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  const size_t need = (size_t)nchunks * na * RC * sizeof(double);
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, need));
  double* part = (double*)h->ws;
  hipEvent_t stop = mgp_prof_begin(h);
  hipLaunchKernelGGL((sweep_mfma_kernel<KIND, KS, RC>), grid, dim3(kThreadsM), 0, h->stream, A, na, B, nb, b_chunk, W,
                     w_sj, w_sr, part, 1L, na, na * (long)RC, D, prm, 0.0, (const double*)nullptr, 0L, 0L, gate);
  mgp_prof_end(h, stop);
  MGP_LAUNCH_CHECK(h);
  const long tot = na * RC;
  hipLaunchKernelGGL((reduce_partials_m_kernel<double>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     h->stream, (const double*)part, na, RC, (int)nchunks, out, o_si, o_sr, alpha, addend, ad_si,
                     ad_sr, gate);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <int KIND, int KS>
int mfma_rc(mgp_handle* h, const SweepParams& prm, int D, const double* A, long na, const double* B, long nb,
            const double* W, long w_sj, long w_sr, int R, double* out, long o_si, long o_sr, double alpha,
            const double* addend, long ad_si, long ad_sr, const int* gate) {
  int r0 = 0;
  while (r0 < R) {
    const int left = R - r0;
    const double* Wr = W + (long)r0 * w_sr;
    double* outr = out + (long)r0 * o_sr;
    const double* adr = addend ? addend + (long)r0 * ad_sr : nullptr;
    int rc;
    if (left >= 4) {
      rc = 4;
      MGP_TRY((launch_mfma<KIND, KS, 4>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr, ad_si,
                                         ad_sr, gate)));
    } else if (left >= 2) {
      rc = 2;
      MGP_TRY((launch_mfma<KIND, KS, 2>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr, ad_si,
                                         ad_sr, gate)));
    } else {
      rc = 1;
      MGP_TRY((launch_mfma<KIND, KS, 1>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr, ad_si,
                                         ad_sr, gate)));
    }
    r0 += rc;
  }
  return MGP_OK;
}

template <int KIND>
int mfma_ks(mgp_handle* h, const SweepParams& prm, int D, const double* A, long na, const double* B, long nb,
            const double* W, long w_sj, long w_sr, int R, double* out, long o_si, long o_sr, double alpha,
            const double* addend, long ad_si, long ad_sr, const int* gate) {
  const int ks = (D + 2 + 3) / 4;
#define MGP_KS_CASE(V)                                                                                         \
  case V:                                                                                                      \
    return mfma_rc<KIND, V>(h, prm, D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si, \
                            ad_sr, gate)
  switch (ks) {
    MGP_KS_CASE(1);
    MGP_KS_CASE(2);
    MGP_KS_CASE(3);
    MGP_KS_CASE(4);
    MGP_KS_CASE(5);
    MGP_KS_CASE(6);
    MGP_KS_CASE(7);
    MGP_KS_CASE(8);
    MGP_KS_CASE(9);
    default:
      return mgp_fail(h, MGP_E_SHAPE, "sweep_mfma: D=%d unsupported", D);
  }
#undef MGP_KS_CASE
}

}  // namespace

// fp64 only; same semantics as the VALU path in sweep.hip
int mgp_sweep_mfma_f64(mgp_handle* h, const mgp_kernel* k, const double* A, long na, const double* B, long nb,
                       const double* W, long w_sj, long w_sr, int R, double* out, long o_si, long o_sr,
                       double alpha, const double* addend, long ad_si, long ad_sr, const int* gate) {
  const SweepParams prm = mgp_make_params(k);
  switch (k->kind) {
    case MGP_SE:
      return mfma_ks<0>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si, ad_sr,
                        gate);
    case MGP_MATERN12:
      return mfma_ks<1>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si, ad_sr,
                        gate);
    case MGP_MATERN32:
      return mfma_ks<2>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si, ad_sr,
                        gate);
    default:
      return mfma_ks<3>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si, ad_sr,
                        gate);
  }
}
