// sweep.hip -- fused matrix-free kernel products on the VALU (rows M1, K1-K3 of SURVEY §8a).
//
//   out(i, r) = variance * sum_j k(a_i, b_j) * w(j, r)    [+ alpha * addend(i, r)]
//
// One kernel serves both directions of the path:
//   K_nm V : owned points a = rows of X (one or more per lane), broadcast points b = Z
//   K_mn W : owned points a = Z, broadcast points b = rows of X, split into chunks over
//            blockIdx.y; per-chunk partials are summed by a second kernel in fixed order
//            (deterministic, no float atomics).
//
// Data movement: every lane reads its own point(s) once (coalesced row reads, scaled by
// c/lengthscale in registers); broadcast points are staged tile by tile into LDS already
// scaled, doubled and with their negative squared norm, so the inner loop per pair is
//   D fma (2 a.b - |a|^2 - |b|^2, GPflow's square_distance expansion)  +  profile  +  RC fma
// with all LDS reads being wave-uniform broadcasts (conflict free).  The kernel is bound by
// fp64 VALU issue, not HBM: algorithmic bytes are s(ND + MD + MR + NR).
#include <type_traits>

#include "mgp_common.h"

namespace {

template <int DP>
struct TileCfg {
  static constexpr int TB = DP <= 8 ? 256 : (DP <= 16 ? 128 : 64);  // broadcast points per LDS tile
  static constexpr int RPT = DP <= 8 ? 4 : 2;                        // owned points per lane
  static constexpr int UJ = (RPT >= 4 || DP >= 32) ? 1 : 2;          // streamed points per loop trip
};

constexpr int kThreads = 256;

// SQ = true accumulates k^2 instead of k (diag of K_mn K_nm for the Jacobi preconditioner)
template <typename T, int DP, int KIND, int RC, bool SQ = false>
__global__ __launch_bounds__(kThreads, (DP <= 8 && RC == 1) ? 4 : 1) void sweep_kernel(
    const T* __restrict__ A, long na, const T* __restrict__ B, long nb, long b_chunk,
    const T* __restrict__ W, long w_sj, long w_sr, T* __restrict__ out, long o_si, long o_sr,
    long o_chunk, int D, SweepParams prm, T alpha, const T* __restrict__ addend, long ad_si,
    long ad_sr, const int* __restrict__ gate, int nblk, int nchunks, unsigned long long* __restrict__ clk) {
  if (gate != nullptr && *gate == 0) return;
  mgp_prof_stamp(clk, 0);
  // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so the
  // nblk workgroups that stream the SAME chunk of broadcast points get linear ids that differ by
  // multiples of 8 (same XCD, dispatched together) and share the chunk through that XCD's L2.
  // Placement only changes speed/traffic, never the result.
  // Only when the chunk count is a multiple of 8 (otherwise the plain decode keeps all XCDs busy).
  const int lin = blockIdx.x;
  int bx, by;
  if ((nchunks & 7) == 0) {
    const int grp = lin / (8 * nblk), rem = lin - grp * (8 * nblk);
    bx = rem >> 3;
    by = grp * 8 + (rem & 7);
  } else {
    by = lin / nblk;
    bx = lin - by * nblk;
  }
  constexpr int TB = TileCfg<DP>::TB;
  constexpr int RPT = TileCfg<DP>::RPT;
  // fp32 bodies are a handful of instructions per pair: unroll so LDS latency hides behind them
  constexpr int UJ = (sizeof(T) == 4 && DP <= 8) ? 4 : TileCfg<DP>::UJ;
  constexpr int PS = (DP + 1 + RC + 1) & ~1;  // per-point LDS stride, even => 16-B aligned rows
  __shared__ __attribute__((aligned(16))) T tile[TB * PS];
  // fp64: 2^(i/2048) table for mgp_exp2_tab (16 KB); fp32 uses v_exp_f32 and no table
  __shared__ double e2tab[sizeof(T) == 8 ? MGP_EXP2_TAB_SIZE : 1];

  const int t = threadIdx.x;
  const long base = (long)bx * (kThreads * RPT);
  if (sizeof(T) == 8) {
    for (int e = t; e < MGP_EXP2_TAB_SIZE; e += kThreads) e2tab[e] = mgp_exp2_tab_entry(e);
  }
  __shared__ T bmax_w[kThreads / 64];  // per-wave max |b|^2 of the staged tile
  __shared__ T amax_w[kThreads / 64];

  // SE / fp64 fast path: a2 is folded into the exp2 magic constant (mgp_exp2_tab_shifted)
  constexpr bool FAST = KIND == 0 && sizeof(T) == 8 && !SQ;
  // SE / fp32: while every owned point of the workgroup has |a|^2 < 64 (log2 units; fp32 reaches 2^127, and the sums carry up to nb terms times |w| on top of 2^64)
  // the sums are kept scaled by 2^(|a_i|^2): the pair's exponent is 2 a.b - |b|^2 <= |a|^2 -- no overflow, and it
  // underflows only where the kernel value does -- so the subtraction of |a|^2 leaves the loop (4 instead of 5
  // instructions per pair at D = 2) and returns as one factor 2^(-|a_i|^2) per output.  Same rounding behaviour:
  // the expansion form already carries terms of size |a|^2 + |b|^2 in the exponent.
  constexpr bool FOLD32 = KIND == 0 && sizeof(T) == 4 && !SQ;

  // ---- owned points: scaled coordinates and squared norm in registers
  // (FAST keeps only cq per owned point live across the sweep; a2 and the 2^rho scales are
  // recomputed from the coordinates where the rare paths need them -- register pressure decides
  // between 3 and 4 waves per SIMD here)
  T a[RPT][DP];
  T a2[FAST ? 1 : RPT];
  double cq[FAST ? RPT : 1];
  T acc[RPT][RC];
  auto norm2 = [&](int q) {
    T s = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) s = mgp_fma(a[q][d], a[q][d], s);
    return s;
  };
  T amax = 0;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    long i = base + q * kThreads + t;
    if (i >= na) i = na - 1;  // clamp: computed but never stored
    T s = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      T v = d < D ? A[i * D + d] * (T)prm.inv_ls[d] : (T)0;
      a[q][d] = v;
      s = mgp_fma(v, v, s);
    }
    amax = s > amax ? s : amax;
    if (FAST)
      cq[q] = (double)MGP_EXP2_MAGIC - (double)s;  // rounded to a multiple of 2^-11
    else
      a2[q] = s;
#pragma unroll
    for (int r = 0; r < RC; ++r) acc[q][r] = 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_xor(amax, off, 64);
    amax = o > amax ? o : amax;
  }
  if ((t & 63) == 0) amax_w[t >> 6] = amax;

  const long jb = (long)by * b_chunk;
  const long je = (jb + b_chunk < nb) ? jb + b_chunk : nb;
  const T clamp = (T)prm.clamp;

  for (long j0 = jb; j0 < je; j0 += TB) {
    __syncthreads();  // previous tile fully consumed
    T bs = 0;
    if (t < TB) {
      const long j = j0 + t;
      T* p = &tile[t * PS];
      if (j < je) {
        T s = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          T v = d < D ? B[j * D + d] * (T)prm.inv_ls[d] : (T)0;
          s = mgp_fma(v, v, s);
          p[d] = v + v;
        }
        p[DP] = -s;
        bs = s;
#pragma unroll
        for (int r = 0; r < RC; ++r) p[DP + 1 + r] = W[j * w_sj + r * w_sr];
      } else {
#pragma unroll
        for (int d = 0; d < DP + 1 + RC; ++d) p[d] = 0;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const T o = __shfl_xor(bs, off, 64);
      bs = o > bs ? o : bs;
    }
    if ((t & 63) == 0) bmax_w[t >> 6] = bs;
    __syncthreads();
    // |t| = |a-b|^2 <= 2(|a|^2 + |b|^2): below 2^19 the exp2 table form needs no clamp
    T aa = 0, bb = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      aa = amax_w[w] > aa ? amax_w[w] : aa;
      bb = bmax_w[w] > bb ? bmax_w[w] : bb;
    }
    const bool safe = (T)2 * (aa + bb) < (T)524288;  // NaN inputs compare false -> clamped loop
    const bool fold = FOLD32 && aa < (T)64;          // the same for every tile of this workgroup (aa: owned points only)

    auto body = [&](auto e2) {
#pragma unroll UJ
      for (int jj = 0; jj < TB; ++jj) {
        const T* p = &tile[jj * PS];
        T b[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) b[d] = p[d];
        const T nb2 = p[DP];
        T w[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) w[r] = p[DP + 1 + r];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          // FAST kernels reach this loop only for tiles that fail the distance bound (rare): a2 and
          // the 2^(-rho) factor that keeps the common 2^rho epilogue valid are recomputed per pair
          // rather than held in registers across the whole sweep
          const T a2q = FAST ? norm2(q) : a2[FAST ? 0 : q];
          T s = nb2 - a2q;
#pragma unroll
          for (int d = 0; d < DP; ++d) s = mgp_fma(a[q][d], b[d], s);
          T kv = mgp_profile<KIND, T, decltype(e2)>(s, clamp, e2);
          if (SQ) kv = kv * kv;
          if (FOLD32) kv = fold ? kv * mgp_exp2(a2q) : kv;  // rare path of a folding workgroup: match the scale of its sums
          if (FAST) kv = (T)((double)kv * mgp_exp2(-(((double)MGP_EXP2_MAGIC - cq[FAST ? q : 0]) - (double)a2q)));
#pragma unroll
          for (int r = 0; r < RC; ++r) acc[q][r] = mgp_fma(kv, w[r], acc[q][r]);
        }
      }
    };
    auto body_fast = [&]() {
#pragma unroll UJ
      for (int jj = 0; jj < TB; ++jj) {
        const T* p = &tile[jj * PS];
        T b[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) b[d] = p[d];
        const T nb2 = p[DP];
        T w[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) w[r] = p[DP + 1 + r];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          T s = nb2;
#pragma unroll
          for (int d = 0; d < DP; ++d) s = mgp_fma(a[q][d], b[d], s);
          const T kv = (T)mgp_exp2_tab_shifted((double)s, cq[FAST ? q : 0], e2tab);
#pragma unroll
          for (int r = 0; r < RC; ++r) acc[q][r] = mgp_fma(kv, w[r], acc[q][r]);
        }
      }
    };
    auto body_fold = [&]() {
#pragma unroll UJ
      for (int jj = 0; jj < TB; ++jj) {
        const T* p = &tile[jj * PS];
        T b[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) b[d] = p[d];
        const T nb2 = p[DP];
        T w[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) w[r] = p[DP + 1 + r];
        if constexpr (sizeof(T) == 4) {
          // two owned points per instruction: v_pk_fma_f32 for the distance chains and the accumulation (the
          // streamed operand is the same value in both halves), v_exp_f32 once per pair
          typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int q = 0; q < RPT; q += 2) {
            v2f sx = {nb2, nb2};
#pragma unroll
            for (int d = 0; d < DP; ++d)
              sx = __builtin_elementwise_fma(v2f{a[q][d], a[q + 1][d]}, v2f{b[d], b[d]}, sx);
            const v2f kv = {mgp_exp2(sx.x), mgp_exp2(sx.y)};
#pragma unroll
            for (int r = 0; r < RC; ++r) {
              const v2f ac = __builtin_elementwise_fma(kv, v2f{w[r], w[r]}, v2f{acc[q][r], acc[q + 1][r]});
              acc[q][r] = ac.x;
              acc[q + 1][r] = ac.y;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < RPT; ++q) {
            T sx = nb2;
#pragma unroll
            for (int d = 0; d < DP; ++d) sx = mgp_fma(a[q][d], b[d], sx);
            const T kv = mgp_exp2(sx);
#pragma unroll
            for (int r = 0; r < RC; ++r) acc[q][r] = mgp_fma(kv, w[r], acc[q][r]);
          }
        }
      }
    };
    if (safe) {
      if (FAST)
        body_fast();
      else if (FOLD32 && fold)
        body_fold();
      else
        body(E2Tab<false>{e2tab});
    } else {
      body(E2Tab<true>{e2tab});
    }
  }

  mgp_prof_stamp(clk, 1);
  const T var = SQ ? (T)(prm.variance * prm.variance) : (T)prm.variance;
  bool folded = false;
  if (FOLD32 && jb < je) {  // the loop ran, so amax_w is published (its barrier precedes the first tile)
    T aa = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) aa = amax_w[w] > aa ? amax_w[w] : aa;
    folded = aa < (T)64;
  }
  T* o = out + (long)by * o_chunk;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long i = base + q * kThreads + t;
    if (i < na) {
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        T v = var * acc[q][r];
        if (FAST) v = (T)((double)v * mgp_exp2(((double)MGP_EXP2_MAGIC - cq[q]) - (double)norm2(q)));  // 2^rho
        if (FOLD32) v = folded ? v * mgp_exp2(-a2[FAST ? 0 : q]) : v;
        if (addend != nullptr) v = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], v);
        o[i * o_si + r * o_sr] = v;
      }
    }
  }
}


// ---- fp64 fast path (every profile, D <= 32, 1 / 2 / 4 / 8 right-hand sides per launch) ---------------
// (written first for SE, D <= 8, one right-hand side -- the CG case -- and described for it:)
// Same arithmetic as the FAST branch of sweep_kernel, re-laid-out for the issue ports of a CDNA4
// CU (profiles/r02_valu_issue_probe.txt: a VALU-bound loop pays ~0.7-0.8 of an fp64 slot for every
// 32-bit integer op and for every per-lane LDS gather, and nothing for scalar loads):
//   * streamed points are packed once (pack_points_kernel: 2 c b/l and -|c b/l|^2, DP+1 doubles)
//     and read by SCALAR loads, double-buffered in SGPRs: no LDS tile, no barrier, no broadcast
//     ds_read, no VGPRs spent on them -- the fma takes the SGPR pair as its operand;
//   * 2^n is applied by ONE integer add on the high word of the gathered table entry: the table
//     holds T'[i] = T[i] with (i << 9) subtracted from its high word, so hi + (m << 9) =
//     hi(T[i]) + (n << 20) for m = 2048 n + i (replaces v_ldexp_f64 + v_ashrrev_i32);
//   * LDS holds only the 16 KB table.
// Per pair: D fma + 3 add (magic) + 2 fma + mul + fma (polynomial, table) + fma (accumulate)
// = D + 8 fp64 and 3 integer instructions (and, lshl, lshl_add), one ds_read_b64 gather.
// Valid while the exponent stays normal: t = -|a-b|^2 >= -2(|a|^2+|b|^2) > -1000 (scaled units),
// checked per workgroup against the packed set's maximum norm; otherwise, and for NaN inputs, the
// clamped variant of the same loop runs (2^-1000 stands for 0; NaN stays NaN).

// Rows are formed one per thread and leave through LDS, so that the block's (DP+1)*NTP doubles are written
// as one contiguous run (a 72-byte row per lane written in place cost 220 us for 2^20 rows; this form ~50).
template <int DP>
struct PackCfg {
  static constexpr int NTP = DP > 16 ? 128 : 256;  // (DP+1) * NTP * 8 bytes of LDS <= 64 KB
};

template <int DP>
__global__ __launch_bounds__(PackCfg<DP>::NTP) void pack_points_kernel(const double* __restrict__ B, long nb, int D,
                                                                       SweepParams prm, double* __restrict__ P,
                                                                       unsigned long long* __restrict__ bmax_bits) {
  constexpr int NTP = PackCfg<DP>::NTP;
  __shared__ double rows[NTP * (DP + 1)];  // odd stride in doubles: conflict-free row writes
  const int t = threadIdx.x;
  const long base = (long)blockIdx.x * NTP;
  const long j = base + t;
  double s = 0;
  double* p = rows + t * (DP + 1);
  if (j < nb) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const double v = d < D ? B[j * D + d] * prm.inv_ls[d] : 0.0;
      s = mgp_fma(v, v, s);
      p[d] = v + v;
    }
    p[DP] = -s;
  } else {  // row nb is the pad row (the origin): the multi-right-hand-side sweep streams an even count
#pragma unroll
    for (int d = 0; d <= DP; ++d) p[d] = 0.0;
  }
  __syncthreads();
  long live = nb + 1 - base;  // rows of this block that exist (incl. the pad row)
  live = live > NTP ? NTP : live;
  const int cnt = (int)live * (DP + 1);
  double* dst = P + base * (DP + 1);
  for (int e = t; e < cnt; e += NTP) dst[e] = rows[e];
  // max |b|^2 of the whole set (bit pattern of a non-negative double orders like the value; a NaN
  // pattern is larger than every number, so a NaN point makes the set "unsafe")
  unsigned long long bits = __builtin_bit_cast(unsigned long long, s) & 0x7fffffffffffffffULL;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(bits, off, 64);
    bits = o > bits ? o : bits;
  }
  // one atomic per block at most, and none when the word already holds as much: 16 K same-address atomics
  // (one per wave of 2^20 rows) serialised in L2 and took 160 of this kernel's 200 us
  __shared__ unsigned long long wmax[NTP / 64];
  if ((t & 63) == 0) wmax[t >> 6] = bits;
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int w = 1; w < NTP / 64; ++w) bits = wmax[w] > bits ? wmax[w] : bits;
    if (bits > __hip_atomic_load(bmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bmax_bits, bits);
  }
}

// wT[j][r] = W[j * w_sj + r * w_sr]: the weights of a streamed point side by side, for the scalar loads of
// the multi-right-hand-side fast sweep (a launch-time copy of nb * RC doubles: 0.1 % of the sweep's time)
template <int RC>
__global__ __launch_bounds__(256) void transpose_weights_kernel(const double* __restrict__ W, long w_sj, long w_sr,
                                                                long nb, double* __restrict__ wT,
                                                                const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  __shared__ double rows[256 * (RC + 1)];
  const int t = threadIdx.x;
  const long base = (long)blockIdx.x * 256;
  const long j = base + t;
#pragma unroll
  for (int r = 0; r < RC; ++r) rows[t * (RC + 1) + r] = j < nb ? W[j * w_sj + r * w_sr] : 0.0;  // row nb: the pad row
  __syncthreads();
  long live = nb + 1 - base;
  live = live > 256 ? 256 : live;
  const int cnt = (int)live * RC;
  double* dst = wT + base * RC;
  for (int e = t; e < cnt; e += 256) dst[e] = rows[(e / RC) * (RC + 1) + (e % RC)];  // RC is a power of two
}

// T'[e] = 2^(e / 2^TBITS) with e << (20 - TBITS) subtracted from its high word (see below); one copy per handle
template <int TBITS>
__global__ __launch_bounds__(256) void build_e2tab_kernel(double* __restrict__ tab) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= (1 << TBITS)) return;
  unsigned long long bits = __builtin_bit_cast(unsigned long long, mgp_exp2((double)e * (1.0 / (1 << TBITS))));
  bits -= (unsigned long long)e << (20 - TBITS + 32);  // high word -= e << (20 - TBITS)
  tab[e] = __builtin_bit_cast(double, bits);
}

// NT threads per workgroup share one table of 2^TBITS entries (t = m / 2^TBITS + g, |g| <= 2^-(TBITS+1)):
//   TBITS 11, NT 256: 16 KB, several workgroups per CU; byte offset of the entry by shift + and;
//   TBITS 13, NT 512: 64 KB, two workgroups per CU; the offset is ONE instruction -- an SDWA shift that
//   keeps the low 16 bits of (m << 3), i.e. (m & 8191) * 8 -- so an SE pair costs D + 8 fp64 + 2 integer.
// KIND 0 (SE) folds |a|^2 into the magic constant; the Matern kinds take t = -q, q = sqrt(max(r^2, floor)),
// through the same table and multiply by their polynomial in q (mgp_math.h, mgp_profile).
// DBUF: two SGPR copies of the streamed row (the next point's scalar loads fly during the current point);
// at DP = 32 a row is 66 SGPRs, so there is one copy and the next row is requested as soon as the distance
// phase of the current point has consumed it.
// RC right-hand sides: the weights of a streamed point are RC consecutive doubles (w_sj = RC: the caller hands
// the transposed copy made by transpose_weights_kernel), read by one scalar load into RC SGPR pairs; each
// pair's kernel value feeds RC accumulators.
#ifndef MGP_D32_WAVES
#define MGP_D32_WAVES 3  // waves per SIMD the D = 32, two-points-per-lane instantiations are compiled for: 167 VGPRs (one under
                         // the three-wave limit; the compiler took 171 when allowed 256) -- C5 sweep 7.86 -> 7.64 ms in an in-run A/B
#endif
template <int DP, int KIND, int RC, int RPT, int NT, int TBITS, bool DBUF>
__global__ __launch_bounds__(NT, DP > 16 ? (RPT == 1 ? 4 : MGP_D32_WAVES) : (RPT >= 4 || NT == 512 || DP > 8 ? 4 : (RPT == 3 ? 5 : 8))) void sweep_fast_kernel(
    const double* __restrict__ A, long na, const double* __restrict__ Pk, long nb, long b_chunk,
    const double* __restrict__ W, long w_sj, double* __restrict__ out, long o_si, long o_sr, long o_chunk, int D,
    SweepParams prm, double alpha, const double* __restrict__ addend, long ad_si, long ad_sr,
    const int* __restrict__ gate, int nblk, int nchunks, const unsigned long long* __restrict__ bmax_bits,
    int pf_mask, int pf_ahead, const double* __restrict__ gtab, unsigned long long* __restrict__ clk) {
  if (gate != nullptr && *gate == 0) return;
  mgp_prof_stamp(clk, 0);
  const int lin = blockIdx.x;
  int bx, by;
  if ((nchunks & 7) == 0) {  // XCD-aware decode, as sweep_kernel: the workgroups that stream one chunk share an XCD
    const int grp = lin / (8 * nblk), rem = lin - grp * (8 * nblk);
    bx = rem >> 3;
    by = grp * 8 + (rem & 7);
  } else if (nchunks <= 4) {
    // few chunks (C3's K_nm.v: two): the workgroups that OWN the same rows -- one per chunk -- get linear ids
    // that differ by 8, i.e. the same XCD a moment apart, so the rows are fetched from HBM once and re-read
    // from that XCD's L2 (grid padded to a multiple of 8 row blocks; the padding exits here)
    const int grp = lin / (8 * nchunks), rem = lin - grp * (8 * nchunks);
    by = rem >> 3;
    bx = grp * 8 + (rem & 7);
    if (bx >= nblk) return;
  } else {
    by = lin / nblk;
    bx = lin - by * nblk;
  }
  constexpr int TSIZE = 1 << TBITS;
  constexpr double MAGIC = TBITS == 11 ? 0x1.8p+41 : 0x1.8p+39;  // ulp = 2^-TBITS: low word of t + MAGIC = round(2^TBITS t)
  static_assert(TBITS == 11 || TBITS == 13, "table sizes: 2048 or 8192 entries");
  // the exponent must stay normal: t > -1000.  SE: t = -r^2 (scaled); Matern: t = -q, q^2 = r^2 (scaled)
  constexpr double kTLimit = 1000.0, kNormLimit = KIND == 0 ? kTLimit : kTLimit * kTLimit;
  __shared__ double e2tab[TSIZE];
  __shared__ unsigned long long amax_w[NT / 64];
  const int t = threadIdx.x;
  // the table is copied from the handle's device copy (build_e2tab_kernel, once per handle): computing its 16
  // entries per thread was 700 instructions in front of every workgroup -- 5-8 % of a workgroup that streams
  // 256 points (a rank's share of C3), <1 % of one that streams 2048
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int e = 2 * t; e < TSIZE; e += 2 * NT) *reinterpret_cast<d2*>(&e2tab[e]) = *reinterpret_cast<const d2*>(gtab + e);
  const long base = (long)bx * (NT * RPT);
  double a[RPT][DP], cq[RPT], acc[RPT][RC];  // cq: SE -> MAGIC - |a|^2 ; Matern -> |a|^2
  unsigned long long amax = 0;            // bit pattern of max |a|^2: orders like the value, NaN above everything
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    long i = base + q * NT + t;
    if (i >= na) i = na - 1;  // clamp: computed but never stored
    double s = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const double v = d < D ? A[i * D + d] * prm.inv_ls[d] : 0.0;
      a[q][d] = v;
      s = mgp_fma(v, v, s);
    }
    const unsigned long long sb = __builtin_bit_cast(unsigned long long, s) & 0x7fffffffffffffffULL;
    amax = sb > amax ? sb : amax;
    cq[q] = KIND == 0 ? MAGIC - s : s;  // SE: rounded to a multiple of 2^-TBITS
#pragma unroll
    for (int r = 0; r < RC; ++r) acc[q][r] = 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(amax, off, 64);
    amax = o > amax ? o : amax;
  }
  if ((t & 63) == 0) amax_w[t >> 6] = amax;
  __syncthreads();  // table + amax_w
  unsigned long long ab = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) ab = amax_w[w] > ab ? amax_w[w] : ab;
  const double aa = __builtin_bit_cast(double, ab), bb = __builtin_bit_cast(double, *bmax_bits);
  const bool safe = 2.0 * (aa + bb) < kNormLimit;  // NaN compares false -> clamped loop

  const long jb = (long)by * b_chunk;
  const long je = (jb + b_chunk < nb) ? jb + b_chunk : nb;
  const double C1 = 0x1.62e42fefa39efp-1, C2 = 0x1.ebfbdff82c58fp-3, C3 = 0x1.c6b08d704a0c0p-5;  // ln2^k / k!
  const double floor_r2 = prm.clamp;  // GPflow's 1e-36 under the sqrt, in scaled units

  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const char* tab_bytes = (const char*)e2tab;
  // One streamed point against the RPT owned points, in phases, so that the table gathers of all RPT pairs
  // are in flight while the polynomial is evaluated (one LDS wait per point, not per pair).
  // dist: the distance chains (the only consumers of the row held in SGPRs)
  auto dist = [&](const double (&b)[DP], double nb2, double (&sv)[RPT]) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      double s = KIND == 0 ? nb2 : nb2 - cq[q];
#pragma unroll
      for (int d = 0; d < DP; ++d) s = mgp_fma(a[q][d], b[d], s);
      sv[q] = s;  // -(scaled r^2) (SE: + |a|^2, which the magic constant carries)
    }
  };
  auto finish = [&](auto clamp_tag, const double (&sv)[RPT], const double (&w)[RC]) {
    constexpr bool CLAMP = decltype(clamp_tag)::value;
    double g[RPT], tq[RPT], qv[KIND == 0 ? 1 : RPT];
    unsigned ex[RPT];
    bool under[RPT];  // CLAMP only: t below the limit -> the pair contributes an exact 0, as exp's underflow does
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      double u, gg;
      bool low = false;
      if (KIND == 0) {
        const double s = sv[q];
        u = s + cq[q];
        if (CLAMP) {
          const double cmin = MAGIC - kTLimit;
          low = u < cmin;  // false for NaN: NaN flows on
          u = low ? cmin : u;
        }
        gg = s - (u - cq[q]);
      } else {
        double r2 = -sv[q];
        // the floor keeps sqrt off zero; the comparison form keeps a NaN (tf.maximum does), fmax would not --
        // NaN inputs are caught by the norm bound and take the CLAMP variant
        r2 = CLAMP ? (r2 < floor_r2 ? floor_r2 : r2) : __builtin_fmax(r2, floor_r2);
        double qq = mgp_sqrt_pos(r2);
        if (CLAMP) {
          low = qq > kTLimit;
          qq = low ? kTLimit : qq;
        }
        qv[q] = qq;
        u = MAGIC - qq;
        gg = -qq - (u - MAGIC);
      }
      under[q] = low;
      const unsigned m = __builtin_bit_cast(u32x2, u).x;  // round(2^TBITS t) in two's complement
      g[q] = (KIND == 0 && CLAMP) ? (low ? 0.0 : gg) : gg;
      // byte offset of table entry m & (2^TBITS - 1) and the exponent increment (on the high word): pinned
      // as 32-bit instructions (left to itself the compiler packs pairs of indices with v_perm and widens
      // the exponent add to 64 bits: ~7 integer instructions per pair instead of 2-3)
      unsigned off;
      if (TBITS == 11)
        asm("v_lshlrev_b32 %0, 3, %1\n\tv_and_b32 %0, 0x3ff8, %0" : "=&v"(off) : "v"(m));
      else  // low 16 bits of (m << 3), zero padded: (m & 8191) * 8
        asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
            : "=v"(off)
            : "v"(m));
      ex[q] = m;
      tq[q] = *(const double*)(tab_bytes + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    double pq[RPT];  // stage-major: RPT independent chains back to back, no dependent-issue stall
#pragma unroll
    for (int q = 0; q < RPT; ++q) pq[q] = mgp_fma(g[q], C3, C2);
#pragma unroll
    for (int q = 0; q < RPT; ++q) pq[q] = mgp_fma(pq[q], g[q], C1);
#pragma unroll
    for (int q = 0; q < RPT; ++q) pq[q] = pq[q] * g[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      u32x2 tb = __builtin_bit_cast(u32x2, tq[q]);
      // hi(T') + (m << (20 - TBITS)) = hi(T) + (n << 20)
      if (TBITS == 11)
        asm("v_lshl_add_u32 %0, %1, 9, %0" : "+v"(tb.y) : "v"(ex[q]));
      else
        asm("v_lshl_add_u32 %0, %1, 7, %0" : "+v"(tb.y) : "v"(ex[q]));
      const double T2 = __builtin_bit_cast(double, tb);
      double kv = mgp_fma(T2, pq[q], T2);  // 2^t
      if (KIND == 2) kv *= mgp_fma(qv[KIND == 0 ? 0 : q], MGP_LN2, 1.0);
      if (KIND == 3) {
        const double qq = qv[KIND == 0 ? 0 : q];
        kv *= mgp_fma(mgp_fma(qq, MGP_LN2 * MGP_LN2 / 3.0, MGP_LN2), qq, 1.0);
      }
      if (CLAMP) kv = under[q] ? 0.0 : kv;
#pragma unroll
      for (int r = 0; r < RC; ++r) acc[q][r] = mgp_fma(kv, w[r], acc[q][r]);
    }
  };
  auto sweep_loop = [&](auto clamp_tag) {
    // scalar buffering over the chunk [jb, je): rows are requested ahead of their use; pointers are bumped (no
    // 64-bit index multiplies), the count is 32-bit
    const long wst = RC > 1 ? (long)RC : w_sj;  // RC > 1: the transposed copy, RC doubles per point
    const double* rp = Pk + jb * (DP + 1);
    const double* wp = W + jb * wst;
    int rem = (int)(je - jb);
    auto load = [&](const double* r, const double* wq, bool live, double (&b)[DP], double& nb2, double (&w)[RC]) {
#pragma unroll
      for (int d = 0; d < DP; ++d) b[d] = r[d];
      nb2 = r[DP];
#pragma unroll
      for (int c = 0; c < RC; ++c) {
        const double wv = wq[c];
        w[c] = live ? wv : 0.0;
      }
    };
    // The scalar loads run only about one point ahead: enough for rows that sit in L2 (Z: 300 KB), not for rows
    // that come from HBM (the K_mn direction streams X).  So every pf_mask+1 trips each lane touches one
    // 128-byte line of the packed rows pf_ahead bytes further on (a vector load whose result is never
    // used: it only pulls the lines into this XCD's L2 before the scalar loads ask for them).
    const char* pf_end = (const char*)(Pk + (je - 1) * (DP + 1));
    // RC > 1: the transposed weights stream like the rows; both buffers are the library's own and end in
    // pf_ahead + 8 KB of slack, so the prefetch address needs no clamp there (SGPRs are short)
    int trip = 0;
    unsigned pf_sink = 0, pw_sink = 0;
    auto prefetch = [&]() {
      // RC > 1: a countdown (SGPRs are short there; the mask form re-read pf_mask from the kernel arguments
      // every trip and waited for every scalar load in flight)
      bool now;
      if (RC > 1) {
        now = trip == 0;
        trip = now ? pf_mask : trip - 1;
      } else {
        now = (trip++ & pf_mask) == 0;
      }
      if (now) {
        // lane id recomputed here (two v_mbcnt) rather than kept live across the loop: the loop is at the
        // register limit of its occupancy
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const char* pa = (const char*)rp + pf_ahead + lane * 128;
        if (RC == 1) pa = pa < pf_end ? pa : pf_end;
        // the value is looked at only when the NEXT prefetch is issued (many trips later, long after it
        // has landed), so the compiler's vmcnt wait for it costs nothing and the loop body never stalls on it
        asm volatile("" ::"v"(pf_sink));
        pf_sink = *(const unsigned*)pa;
        if (RC > 1) {
          const char* pb = (const char*)wp + pf_ahead + lane * 128;
          asm volatile("" ::"v"(pw_sink));
          pw_sink = *(const unsigned*)pb;
        }
      }
    };
    double sv[RPT];
    if constexpr (RC > 1) {
      // Several right-hand sides: ONE copy of the weights, requested when its point's distance phase starts
      // and used when that point's finish ends (the wait for the table gathers in between covers it) -- a
      // second copy a point ahead, as the rows have, does not fit the SGPR file next to them.
      auto load_row = [&](const double* r, double (&b)[DP], double& nb2) {
#pragma unroll
        for (int d = 0; d < DP; ++d) b[d] = r[d];
        nb2 = r[DP];
      };
      auto load_w = [&](const double* wq, double (&w)[RC]) {
#pragma unroll
        for (int c = 0; c < RC; ++c) w[c] = wq[c];
      };
      double w0[RC];
      if (DBUF) {  // an even count (pad row with zero weights): no tail mask
        double b0[DP], b1[DP], n0, n1;
        load_row(rp, b0, n0);
        while (rem > 0) {
          prefetch();
          const double* r1 = rp + (DP + 1);
          load_row(r1, b1, n1);
          load_w(wp, w0);
          dist(b0, n0, sv);
          finish(clamp_tag, sv, w0);
          const double* r2 = rem > 2 ? r1 + (DP + 1) : r1;
          load_row(r2, b0, n0);
          load_w(wp + RC, w0);
          dist(b1, n1, sv);
          finish(clamp_tag, sv, w0);
          rp = r2;
          wp += 2 * RC;
          rem -= 2;
        }
      } else {
        double b0[DP], n0;
        load_row(rp, b0, n0);
        while (rem > 0) {
          prefetch();
          load_w(wp, w0);
          dist(b0, n0, sv);
          rp = rem > 1 ? rp + (DP + 1) : rp;
          __builtin_amdgcn_sched_barrier(0);  // the row is consumed: request the next one now, not at its use
          load_row(rp, b0, n0);
          __builtin_amdgcn_sched_barrier(0);
          finish(clamp_tag, sv, w0);
          wp += RC;
          rem -= 1;
        }
      }
    } else if (DBUF) {
      double b0[DP], b1[DP], n0, n1, w0[RC], w1[RC];
      load(rp, wp, true, b0, n0, w0);
      while (rem > 0) {
        prefetch();
        const bool m1 = rem > 1, m2 = rem > 2;
        const double* r1 = m1 ? rp + (DP + 1) : rp;
        const double* q1 = m1 ? wp + wst : wp;
        load(r1, q1, m1, b1, n1, w1);
        dist(b0, n0, sv);
        finish(clamp_tag, sv, w0);
        const double* r2 = m2 ? r1 + (DP + 1) : r1;
        const double* q2 = m2 ? q1 + wst : q1;
        load(r2, q2, m2, b0, n0, w0);
        dist(b1, n1, sv);
        finish(clamp_tag, sv, w1);
        rp = r2;
        wp = q2;
        rem -= 2;
      }
    } else {
      double b0[DP], n0, w0[RC];
      load(rp, wp, true, b0, n0, w0);
      while (rem > 0) {
        prefetch();
        dist(b0, n0, sv);
        double wc[RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) wc[c] = w0[c];
        const bool m1 = rem > 1;
        rp = m1 ? rp + (DP + 1) : rp;
        wp = m1 ? wp + wst : wp;
        __builtin_amdgcn_sched_barrier(0);  // the row is consumed: request the next one now, not at its use
        load(rp, wp, m1, b0, n0, w0);
        __builtin_amdgcn_sched_barrier(0);
        finish(clamp_tag, sv, wc);
        rem -= 1;
      }
    }
    asm volatile("" ::"v"(pf_sink));  // the last prefetch is consumed here
    if (RC > 1) asm volatile("" ::"v"(pw_sink));
  };
  if (jb < je) {
    if (safe)
      sweep_loop(std::false_type{});
    else
      sweep_loop(std::true_type{});
  }
  mgp_prof_stamp(clk, 1);

  double* o = out + (long)by * o_chunk;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long i = base + q * NT + t;
    if (i < na) {
      double rho = 1.0;
      if (KIND == 0) {
        double a2 = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) a2 = mgp_fma(a[q][d], a[q][d], a2);
        rho = mgp_exp2((MAGIC - cq[q]) - a2);  // 2^rho: what rounding MAGIC - |a|^2 dropped
      }
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        double v = prm.variance * acc[q][r];
        if (KIND == 0) v *= rho;
        if (addend != nullptr) v = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], v);
        o[i * o_si + r * o_sr] = v;
      }
    }
  }
}

// out(i,r) = sum_c partial[c][r][i] (+ alpha*addend).  Block = 16 groups x 64 columns: group g
// sums chunks g, g+16, ... in index order, the 16 group sums are added in a fixed tree -- the
// result does not depend on scheduling (deterministic, no float atomics).
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const T* __restrict__ part, long na, int R,
                                                               int nchunks, T* __restrict__ out, long o_si,
                                                               long o_sr, T alpha, const T* __restrict__ addend,
                                                               long ad_si, long ad_sr,
                                                               const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  __shared__ T red[16][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + col;
  const long stride = na * R;
  T s = 0;
  if (idx < stride) {
#pragma unroll 4
    for (int c = grp; c < nchunks; c += 16) s += part[(long)c * stride + idx];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && idx < stride) {
    T t8[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) t8[g] = red[2 * g][col] + red[2 * g + 1][col];
    T v = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
    const long r = idx / na, i = idx - r * na;
    if (addend != nullptr) v = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], v);
    out[i * o_si + r * o_sr] = v;
  }
}

// The same sums for at most four chunks (C3's K_nm.p: two), one element per thread: in the kernel above 14 of
// the 16 groups idle then and 25 MB move at 0.8 TB/s (30 us per CG step).  Bit-identical: the same zero-seeded
// group sums through the same tree.
template <typename T>
__global__ __launch_bounds__(256) void reduce_few_partials_kernel(const T* __restrict__ part, long na, int R,
                                                                  int nchunks, T* __restrict__ out, long o_si,
                                                                  long o_sr, T alpha, const T* __restrict__ addend,
                                                                  long ad_si, long ad_sr,
                                                                  const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = na * R;
  if (idx >= stride) return;
  T c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = k < nchunks ? part[(long)k * stride + idx] : (T)0;
  T g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    g[k] = 0;
    if (k < nchunks) g[k] += c[k];
  }
  const T z = 0;
  const T t0 = g[0] + g[1], t1 = g[2] + g[3];
  T v = ((t0 + t1) + (z + z)) + ((z + z) + (z + z));
  const long r = idx / na, i = idx - r * na;
  if (addend != nullptr) v = mgp_fma(alpha, addend[i * ad_si + r * ad_sr], v);
  out[i * o_si + r * o_sr] = v;
}

template <typename T>
void launch_reduce_partials(mgp_handle* h, const T* part, long na, int R, int nchunks, T* out, long o_si, long o_sr,
                            T alpha, const T* addend, long ad_si, long ad_sr, const int* gate) {
  const long tot = na * R;
  if (nchunks <= 4)
    hipLaunchKernelGGL((reduce_few_partials_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       part, na, R, nchunks, out, o_si, o_sr, alpha, addend, ad_si, ad_sr, gate);
  else
    hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((unsigned)((tot + 63) / 64)), dim3(1024), 0, h->stream, part,
                       na, R, nchunks, out, o_si, o_sr, alpha, addend, ad_si, ad_sr, gate);
}

template <typename T, int DP, int KIND, int RC, bool SQ = false>
int launch_sweep(mgp_handle* h, const SweepParams& prm, int D, const T* A, long na, const T* B, long nb,
                 const T* W, long w_sj, long w_sr, T* out, long o_si, long o_sr, T alpha, const T* addend,
                 long ad_si, long ad_sr, const int* gate) {
  constexpr int TB = TileCfg<DP>::TB;
  constexpr bool kFastEligible = std::is_same<T, double>::value && !SQ;
  // fast-path geometry: D <= 8 -> 4 owned points per lane, 512 threads, 8192-entry table (or the 256-thread form);
  // D <= 16 -> 2 points, 512 threads; D <= 32 -> 2 points, 256 threads, 2048-entry table, one SGPR row copy
  // several right-hand sides: 2 owned points per lane (the RC accumulators per point take the registers), 512-thread form only
  const bool fast_on = kFastEligible && h->sweep_fast != 0 && (RC == 1 || h->sweep_fast == 2);
  const int frpt = !fast_on ? 0 : (RC > 1 ? (DP > 16 ? 1 : ((RC <= 4 && DP <= 8) ? h->sweep_fast_rpt_rc : 2)) : (DP <= 8 ? h->sweep_fast_rpt : (DP > 16 ? h->sweep_fast_rpt32 : 2)));  // owned points per lane
  const int RPT = frpt ? frpt : TileCfg<DP>::RPT;
  const int fnt = (frpt && h->sweep_fast == 2 && DP <= 16) ? 512 : kThreads;
  const long per_block = (long)fnt * RPT;
  const long nblk = (na + per_block - 1) / per_block;
  // enough workgroups to fill the chip: none of the streamed set is split when the owned side
  // already gives >= 4 workgroups per CU, else aim for ~8 per CU
  const long target = (long)h->sweep_target_per_cu * h->num_cus * kThreads / fnt;  // workgroups to aim for (256-thread units per CU)
  // The fast kernel splits the streamed set while the owned side gives fewer than 8 (256-thread-equivalent)
  // workgroups per CU: with exactly one resident round (C3's K_nm.v: 4 per CU) the slowest CU sets the time
  // (measured 2.41 -> 2.26 ms with two chunks); the LDS-tile kernels keep the round-1 threshold
  const long nosplit = frpt ? (h->nosplit_per_cu > 8 ? h->nosplit_per_cu : 8) : h->nosplit_per_cu;
  long nchunks = nblk * (fnt / kThreads) >= nosplit * h->num_cus ? 1 : (target + nblk - 1) / nblk;
  // chunk granularity: the LDS-tile kernels stream whole tiles of TB points; the fast kernel has no tile and takes
  // any even count: MGP_SWEEP_GRAN = 64 / 128 / 256 overrides it for A/B runs, the default (0) is TB -- 256 points at
  // D <= 8, 128 at D <= 16, 64 at D <= 32 -- i.e. the chunking every committed measurement ran with
  const long gran = (frpt && h->sweep_chunk_gran > 0) ? (long)h->sweep_chunk_gran : (long)TB;
  const long max_chunks = (nb + gran - 1) / gran;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  long b_chunk = (nb + nchunks - 1) / nchunks;
  b_chunk = (b_chunk + gran - 1) / gran * gran;
  nchunks = (nb + b_chunk - 1) / b_chunk;
  if (nchunks > 65535) return mgp_fail(h, MGP_E_SHAPE, "sweep: too many chunks");
  if (nblk * nchunks > 2147483647L) return mgp_fail(h, MGP_E_SHAPE, "sweep: grid too large");
  dim3 grid((unsigned)(nblk * nchunks));
  if constexpr (kFastEligible) {
    if (fast_on) {
      if ((nchunks & 7) != 0 && nchunks <= 4) grid = dim3((unsigned)((nblk + 7) / 8 * 8 * nchunks));  // see the decode
      // packed streamed set: reused across the iterations of a solve (pack_hold), else rebuilt
      mgp_handle::PackSlot* ps = nullptr;
      if (h->pack_hold)
        for (auto& c : h->pack)
          if (c.valid && c.src == (const void*)B && c.n == nb && c.D == D &&
              memcmp(c.inv_ls, prm.inv_ls, sizeof(c.inv_ls)) == 0)
            ps = &c;
      if (!ps) {
        ps = &h->pack[h->pack_next];
        h->pack_next ^= 1;
        ps->valid = false;
        MGP_TRY(mgp_reserve(h, &ps->buf, &ps->bytes,
                            64 + (size_t)(nb + 1) * (DP + 1) * sizeof(double) + h->pf_ahead + 8192));
        MGP_HIP(h, hipMemsetAsync(ps->buf, 0, 64, h->stream));
        constexpr int NTP = PackCfg<DP>::NTP;
        hipLaunchKernelGGL((pack_points_kernel<DP>), dim3((unsigned)((nb + NTP) / NTP)), dim3(NTP), 0, h->stream, B,
                           nb, D, prm, (double*)((char*)ps->buf + 64), (unsigned long long*)ps->buf);
        MGP_LAUNCH_CHECK(h);
        ps->src = (const void*)B;
        ps->n = nb;
        ps->D = D;
        memcpy(ps->inv_ls, prm.inv_ls, sizeof(ps->inv_ls));
        ps->valid = h->pack_hold;
      }
      const double* Pk = (const double*)((char*)ps->buf + 64);
      const unsigned long long* bmax = (const unsigned long long*)ps->buf;
      if constexpr (RC > 1) {
        MGP_TRY(mgp_reserve(h, &h->gen, &h->gen_bytes, (size_t)(nb + 1) * RC * sizeof(double) + h->pf_ahead + 8192));
        hipLaunchKernelGGL((transpose_weights_kernel<RC>), dim3((unsigned)((nb + 256) / 256)), dim3(256), 0,
                           h->stream, W, w_sj, w_sr, nb, (double*)h->gen, gate);
        MGP_LAUNCH_CHECK(h);
        W = (const T*)h->gen;
        w_sj = RC;
        nb += nb & 1;  // an even count: the two-points-per-trip loop needs no tail mask (pad row, zero weights)
      }
      T* dst = out;
      long d_si = o_si, d_sr = o_sr, d_chunk = 0;
      T a_alpha = alpha;
      const T* a_add = addend;
      long a_si = ad_si, a_sr = ad_sr;
      if (nchunks > 1) {  // partials [chunk][r][i], as reduce_partials_kernel reads them
        MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, (size_t)nchunks * na * RC * sizeof(T)));
        dst = (T*)h->ws;
        d_si = 1;
        d_sr = na;
        d_chunk = na * (long)RC;
        a_alpha = 0;
        a_add = nullptr;
        a_si = a_sr = 0;
      }
      unsigned long long* clk = mgp_prof_clk_next(h);
      hipEvent_t stop = mgp_prof_begin(h);
#define MGP_FAST_LAUNCH(RPTV, NTV, TB, DB)                                                                       \
  hipLaunchKernelGGL((sweep_fast_kernel<DP, KIND, RC, RPTV, NTV, TB, DB>), grid, dim3(NTV), 0, h->stream, A, na, Pk, \
                     nb, b_chunk, W, w_sj, dst, d_si, d_sr, d_chunk, D, prm, a_alpha, a_add, a_si, a_sr, gate,        \
                     (int)nblk, (int)nchunks, bmax, h->pf_trips - 1, h->pf_ahead,                                    \
                     (const double*)h->e2tabs + (TB == 13 ? 0 : 8192), clk)
      if constexpr (RC > 1) {
        // two SGPR copies of (row, weights) while they fit: (DP + 1 + RC) doubles each
        if constexpr (DP > 16) {
          MGP_FAST_LAUNCH(1, 256, 11, false);  // one owned point per lane: 64 VGPRs of coordinates + 2 RC of sums
        } else if constexpr (DP <= 8 && RC <= 4) {
          if (frpt == 3) MGP_FAST_LAUNCH(3, 512, 13, true);
          else MGP_FAST_LAUNCH(2, 512, 13, true);
        } else if constexpr (DP + 1 + RC <= 17) {
          MGP_FAST_LAUNCH(2, 512, 13, true);
        } else {
          MGP_FAST_LAUNCH(2, 512, 13, false);
        }
      } else if constexpr (DP <= 8) {
        if (fnt == 512) MGP_FAST_LAUNCH(4, 512, 13, true);
        else if (frpt == 2) MGP_FAST_LAUNCH(2, 256, 11, true);
        else if (frpt == 3) MGP_FAST_LAUNCH(3, 256, 11, true);
        else MGP_FAST_LAUNCH(4, 256, 11, true);
      } else if constexpr (DP <= 16) {
        if (fnt == 512) MGP_FAST_LAUNCH(2, 512, 13, true);
        else MGP_FAST_LAUNCH(2, 256, 11, true);
      } else {
        if (frpt == 1) MGP_FAST_LAUNCH(1, 256, 11, false);
        else MGP_FAST_LAUNCH(2, 256, 11, false);
      }
#undef MGP_FAST_LAUNCH
      mgp_prof_end(h, stop);
      MGP_LAUNCH_CHECK(h);
      if (nchunks > 1) {
        launch_reduce_partials<T>(h, (const T*)h->ws, na, RC, (int)nchunks, out, o_si, o_sr, alpha, addend, ad_si, ad_sr,
                                  gate);
        MGP_LAUNCH_CHECK(h);
      }
      return MGP_OK;
    }
  }
  if (nchunks == 1) {
    unsigned long long* clk = mgp_prof_clk_next(h);
    hipEvent_t stop = mgp_prof_begin(h);
    hipLaunchKernelGGL((sweep_kernel<T, DP, KIND, RC, SQ>), grid, dim3(kThreads), 0, h->stream, A, na, B, nb,
                       b_chunk, W, w_sj, w_sr, out, o_si, o_sr, 0L, D, prm, alpha, addend, ad_si, ad_sr, gate, (int)nblk,
                       (int)nchunks, clk);
    mgp_prof_end(h, stop);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  const size_t need = (size_t)nchunks * na * RC * sizeof(T);
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, need));
  T* part = (T*)h->ws;
  unsigned long long* clk = mgp_prof_clk_next(h);
  hipEvent_t stop = mgp_prof_begin(h);
  hipLaunchKernelGGL((sweep_kernel<T, DP, KIND, RC, SQ>), grid, dim3(kThreads), 0, h->stream, A, na, B, nb,
                     b_chunk, W, w_sj, w_sr, part, 1L, na, na * (long)RC, D, prm, (T)0, (const T*)nullptr, 0L,
                     0L, gate, (int)nblk, (int)nchunks, clk);
  mgp_prof_end(h, stop);
  MGP_LAUNCH_CHECK(h);
  launch_reduce_partials<T>(h, part, na, RC, (int)nchunks, out, o_si, o_sr, alpha, addend, ad_si, ad_sr, gate);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T, int DP, int KIND>
int sweep_rc(mgp_handle* h, const SweepParams& prm, int D, const T* A, long na, const T* B, long nb, const T* W,
             long w_sj, long w_sr, int R, T* out, long o_si, long o_sr, T alpha, const T* addend, long ad_si,
             long ad_sr, const int* gate) {
  int r0 = 0;
  while (r0 < R) {
    const int left = R - r0;
    const T* Wr = W + (long)r0 * w_sr;
    T* outr = out + (long)r0 * o_sr;
    const T* adr = addend ? addend + (long)r0 * ad_sr : nullptr;
    int rc;
    if (left >= 8) {
      rc = 8;
      MGP_TRY((launch_sweep<T, DP, KIND, 8>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr,
                                             ad_si, ad_sr, gate)));
    } else if (left >= 4) {
      rc = 4;
      MGP_TRY((launch_sweep<T, DP, KIND, 4>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr,
                                             ad_si, ad_sr, gate)));
    } else if (left >= 2) {
      rc = 2;
      MGP_TRY((launch_sweep<T, DP, KIND, 2>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr,
                                             ad_si, ad_sr, gate)));
    } else {
      rc = 1;
      MGP_TRY((launch_sweep<T, DP, KIND, 1>(h, prm, D, A, na, B, nb, Wr, w_sj, w_sr, outr, o_si, o_sr, alpha, adr,
                                             ad_si, ad_sr, gate)));
    }
    r0 += rc;
  }
  return MGP_OK;
}

template <typename T, int KIND>
int sweep_sq_dp(mgp_handle* h, const SweepParams& prm, int D, const T* A, long na, const T* B, long nb, const T* one,
                T* out) {
#define MGP_SQ_CASE(DPV)                                                                                      \
  return launch_sweep<T, DPV, KIND, 1, true>(h, prm, D, A, na, B, nb, one, 0L, 0L, out, 1L, na, (T)0, nullptr, 0L, \
                                             0L, nullptr)
  if (D <= 2) MGP_SQ_CASE(2);
  if (D <= 4) MGP_SQ_CASE(4);
  if (D <= 8) MGP_SQ_CASE(8);
  if (D <= 16) MGP_SQ_CASE(16);
  MGP_SQ_CASE(32);
#undef MGP_SQ_CASE
}

template <typename T>
int sweep_sq_kind(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* one, T* out) {
  const SweepParams prm = mgp_make_params(k);
  switch (k->kind) {
    case MGP_SE: return sweep_sq_dp<T, 0>(h, prm, k->D, A, na, B, nb, one, out);
    case MGP_MATERN12: return sweep_sq_dp<T, 1>(h, prm, k->D, A, na, B, nb, one, out);
    case MGP_MATERN32: return sweep_sq_dp<T, 2>(h, prm, k->D, A, na, B, nb, one, out);
    default: return sweep_sq_dp<T, 3>(h, prm, k->D, A, na, B, nb, one, out);
  }
}

template <typename T, int KIND>
int sweep_dp(mgp_handle* h, const SweepParams& prm, int D, const T* A, long na, const T* B, long nb, const T* W,
             long w_sj, long w_sr, int R, T* out, long o_si, long o_sr, T alpha, const T* addend, long ad_si,
             long ad_sr, const int* gate) {
#define MGP_DP_CASE(DPV)                                                                                   \
  return sweep_rc<T, DPV, KIND>(h, prm, D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, \
                                ad_si, ad_sr, gate)
  if (D <= 2) MGP_DP_CASE(2);
  if (D <= 4) MGP_DP_CASE(4);
  if (D <= 8) MGP_DP_CASE(8);
  if (D <= 16) MGP_DP_CASE(16);
  MGP_DP_CASE(32);
#undef MGP_DP_CASE
}

template <typename T>
int sweep_kind(mgp_handle* h, const mgp_kernel* k, const T* A, long na, const T* B, long nb, const T* W,
               long w_sj, long w_sr, int R, T* out, long o_si, long o_sr, T alpha, const T* addend, long ad_si,
               long ad_sr, const int* gate) {
  const SweepParams prm = mgp_make_params(k);
  switch (k->kind) {
    case MGP_SE:
      return sweep_dp<T, 0>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si,
                            ad_sr, gate);
    case MGP_MATERN12:
      return sweep_dp<T, 1>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si,
                            ad_sr, gate);
    case MGP_MATERN32:
      return sweep_dp<T, 2>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si,
                            ad_sr, gate);
    default:
      return sweep_dp<T, 3>(h, prm, k->D, A, na, B, nb, W, w_sj, w_sr, R, out, o_si, o_sr, alpha, addend, ad_si,
                            ad_sr, gate);
  }
}

// out(i, r) = alpha * addend(i, r) (or 0): the product over an empty streamed set
template <typename T>
__global__ __launch_bounds__(256) void empty_sum_kernel(T* __restrict__ out, long o_si, long o_sr, long na, int R,
                                                        T alpha, const T* __restrict__ addend, long ad_si, long ad_sr,
                                                        const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= na * R) return;
  const long i = e / R, r = e - i * R;
  out[i * o_si + r * o_sr] = addend != nullptr ? alpha * addend[i * ad_si + r * ad_sr] : (T)0;
}

}  // namespace

int mgp_sweep(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
              VecView W, int32_t R, VecViewMut out, double alpha, VecView addend, const int* gate) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (na < 0 || nb < 0 || R < 0) return mgp_fail(h, MGP_E_SHAPE, "negative size");
  if (na == 0 || R == 0) return MGP_OK;
  if (!A || !out.base || (nb > 0 && (!B || !W.base))) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (nb == 0) {
    // empty sum (no inducing points / no local rows): out = alpha*addend, or zeros -- what the dense product of
    // the reference gives for a [B,0].[0,R] contraction (models.py:351 with an empty Kmn)
    const long tot = (long)na * R;
    const unsigned g = (unsigned)((tot + 255) / 256);
    if (k->dtype == MGP_F64)
      hipLaunchKernelGGL((empty_sum_kernel<double>), dim3(g), dim3(256), 0, h->stream, (double*)out.base, out.si,
                         out.sr, (long)na, R, alpha, (const double*)addend.base, addend.si, addend.sr, gate);
    else
      hipLaunchKernelGGL((empty_sum_kernel<float>), dim3(g), dim3(256), 0, h->stream, (float*)out.base, out.si, out.sr,
                         (long)na, R, (float)alpha, (const float*)addend.base, addend.si, addend.sr, gate);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  if (k->D > MGP_FUSED_MAX_D) return mgp_sweep_generic(h, k, A, na, B, nb, W, R, out, alpha, addend, gate);
  if (k->dtype == MGP_F64 && h->sweep_mode == 1)
    return mgp_sweep_mfma_f64(h, k, (const double*)A, na, (const double*)B, nb, (const double*)W.base, W.si, W.sr,
                              R, (double*)out.base, out.si, out.sr, alpha, (const double*)addend.base, addend.si,
                              addend.sr, gate);
  if (k->dtype == MGP_F64)
    return sweep_kind<double>(h, k, (const double*)A, na, (const double*)B, nb, (const double*)W.base, W.si,
                              W.sr, R, (double*)out.base, out.si, out.sr, alpha, (const double*)addend.base,
                              addend.si, addend.sr, gate);
  return sweep_kind<float>(h, k, (const float*)A, na, (const float*)B, nb, (const float*)W.base, W.si, W.sr, R,
                           (float*)out.base, out.si, out.sr, (float)alpha, (const float*)addend.base, addend.si,
                           addend.sr, gate);
}

// out[m] = sum_i k(x_i, z_m)^2 = diag(K_mn K_nm)
extern "C" int mgp_kmn_sq_colsum(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                                 int64_t M, void* out) {
  MGP_TRY(mgp_check_kernel(h, k));
  if (N < 0 || M < 0) return mgp_fail(h, MGP_E_SHAPE, "negative size");
  if (M == 0) return MGP_OK;
  if (!Z || !out || (N > 0 && !X)) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (N == 0) {
    MGP_HIP(h, hipMemsetAsync(out, 0, (size_t)M * mgp_elem(k->dtype), h->stream));
    return MGP_OK;
  }
  if (k->D > MGP_FUSED_MAX_D) return mgp_kmn_sq_colsum_generic(h, k, X, N, Z, M, out);  // generic.hip: explicit panels
  if (k->dtype == MGP_F64)
    return sweep_sq_kind<double>(h, k, (const double*)Z, M, (const double*)X, N, (const double*)h->ones,
                                 (double*)out);
  return sweep_sq_kind<float>(h, k, (const float*)Z, M, (const float*)X, N, (const float*)((char*)h->ones + 8),
                              (float*)out);
}

extern "C" int mgp_knm_matvec(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                              int64_t M, const void* V, int32_t R, int v_layout, void* out, int out_layout) {
  if (!h) return MGP_E_BADARG;
  return mgp_sweep(h, k, X, N, Z, M, mgp_view(V, M, R, v_layout), R, mgp_view_mut(out, N, R, out_layout), 0.0,
                   VecView{nullptr, 0, 0}, nullptr);
}

extern "C" int mgp_kmn_matvec(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                              int64_t M, const void* W, int32_t R, int w_layout, void* out, int out_layout) {
  if (!h) return MGP_E_BADARG;
  return mgp_sweep(h, k, Z, M, X, N, mgp_view(W, N, R, w_layout), R, mgp_view_mut(out, M, R, out_layout), 0.0,
                   VecView{nullptr, 0, 0}, nullptr);
}

// Device copies of the fast sweep's two exp2 tables (8192 + 2048 entries), built once per handle (mgp_create*).
int mgp_build_e2tabs(mgp_handle* h) {
  MGP_HIP(h, hipMalloc(&h->e2tabs, (8192 + 2048) * sizeof(double)));
  hipLaunchKernelGGL((build_e2tab_kernel<13>), dim3(32), dim3(256), 0, nullptr, (double*)h->e2tabs);
  hipLaunchKernelGGL((build_e2tab_kernel<11>), dim3(8), dim3(256), 0, nullptr, (double*)h->e2tabs + 8192);
  MGP_LAUNCH_CHECK(h);
  MGP_HIP(h, hipDeviceSynchronize());
  return MGP_OK;
}
