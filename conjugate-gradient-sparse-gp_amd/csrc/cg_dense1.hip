// cg_dense1.hip -- one right-hand side on a dense symmetric matrix: the reference's literal CG loop
// (cggp/conjugate_gradient.py:59-98 with A = Kmm + Lambda, cggp/models.py:301-303,337-339) in TWO launches per
// iteration, neither of which contains a hand-off between workgroups.
//
//   iteration k (k = 1, 2, ...), the recurrence of conjugate_gradient.py:64-85:
//       Ap = p_k A ; gamma = rz_{k-1} / (p_k . Ap) ; v += gamma p_k ; r -= gamma Ap ;
//       z = M^-1 r ; rz_k = z . r ; p_{k+1} = z + (rz_k / rz_{k-1}) p_k
//
//   T_k  tile kernel, one workgroup per 64x64 tile of the upper triangle (A is read once, n^2/2 elements):
//          - every workgroup adds the per-chunk shares of rz_{k-1} and ||r||^2 that U_{k-1} left (64 numbers, the
//            same fixed-order sum everywhere), applies the stopping rule (:59-62) and forms beta;
//          - forms the entries of p_k it needs ON THE FLY, p_k = z + beta p_{k-1} (:77-84) -- the direction is never
//            a separate pass; the diagonal tiles store p_k for U_k and T_{k+1};
//          - tile products A_IJ p_J and A_IJ^T p_I into their slots (dense.hip's upper-triangle product), and the
//            tile's share of p_k . A p_k  (= p_I . (A_IJ p_J), twice for I < J).
//   U_k  update kernel, one workgroup per 64-element chunk, chunk-local: adds the 2080 tile shares (the same
//        fixed-order sum everywhere) -> gamma (:66-68); adds its chunk's slots -> Ap; v, r (:69,76); z; its shares of
//        rz_k and ||r||^2.
//
// Every global scalar of the recurrence is therefore produced by one launch and consumed by the NEXT one: the
// launch boundary is the only synchronisation.  Round 2 ran tile kernel + slot-reduce launch + a one-workgroup
// update launch (14.0 + 4.6 + 7.1 us at n = 4096); a single launch for slot sums and update with an arrival ticket
// measured 9.2 us, all of it dependent round trips (slots -> write-through publication -> ticket -> re-read ->
// two block reductions).  All sums are in fixed order: results are run-to-run identical.
#include "mgp_common.h"

namespace {

constexpr int CP = 128;  // per-chunk shares: n <= 8192

template <typename T>
__device__ __forceinline__ T wave_allsum(T v) {
  // xor butterfly: paired lanes add the same two numbers (a + b == b + a), so every lane ends with the same bits
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int HALF, int BIT>
__device__ __forceinline__ void rs_step(T (&x)[16], int l) {  // see dense.hip: lane_reduce_scatter_step
  const bool hi = (l & BIT) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const T keep = hi ? x[k + HALF] : x[k];
    const T send = hi ? x[k] : x[k + HALF];
    x[k] = keep + __shfl_xor(send, BIT, 64);
  }
}

// rz and ||r||^2 of the current residual from the per-chunk shares: the same arithmetic in T, U's host finish
// and nowhere else, so every consumer sees the same bits
template <typename T>
__device__ __forceinline__ void sum_shares(const T* __restrict__ cpart, int l, T& rz, T& rr) {
  // all 2 CP entries exist; those of chunks beyond nt were zeroed by mgp_dense1_begin
  rz = wave_allsum(cpart[l] + cpart[64 + l]);
  rr = wave_allsum(cpart[CP + l] + cpart[CP + 64 + l]);
}

// r = b - av ; z ; shares of rz_0 and ||r_0||^2 ; scal[1] = 0 makes T_1 take p_1 = z_0 (the beta-term dropped)
template <typename T>
__global__ __launch_bounds__(64) void d1_init_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ b,
                                                     const T* __restrict__ av, T* __restrict__ r,
                                                     const T* __restrict__ dinv, T* __restrict__ cpart,
                                                     T* __restrict__ scal, T* __restrict__ zpub, long n) {
  const int l = threadIdx.x;
  const long i = (long)blockIdx.x * 64 + l;
  T rv = 0, zv = 0;
  if (i < n) {
    rv = av ? b[i] - av[i] : b[i];
    r[i] = rv;
    zv = dinv ? rv * dinv[i] : rv;
    zpub[i] = zv;  // z_0 for the register-resident form
  }
  const T prz = wave_allsum(zv * rv), prr = wave_allsum(rv * rv);
  if (l == 0) {
    cpart[blockIdx.x] = prz;
    cpart[CP + blockIdx.x] = prr;
    if (blockIdx.x == 0) {
      scal[0] = 0;
      scal[1] = 0;
      ctrl->active = 1;
      ctrl->iters = 0;
      ctrl->ticket = 0;
      ctrl->pad = 0;
    }
  }
}

// statistics + gate for the host poll (once per enqueued batch)
template <typename T>
__global__ __launch_bounds__(64) void d1_finish_kernel(MgpCgCtrl* __restrict__ ctrl, const int* __restrict__ hand_off_err,
                                                       const T* __restrict__ cpart, T* __restrict__ rz,
                                                       T* __restrict__ err, int* __restrict__ over, T thr, int max_it) {
  T s_rz, s_rr;
  sum_shares(cpart, (int)threadIdx.x, s_rz, s_rr);
  if (threadIdx.x == 0) {
    rz[0] = s_rz;
    err[0] = (T)0.5 * s_rz;
    const int any = ((T)0.5 * s_rr > thr) ? 1 : 0;
    over[0] = any;
    if (*hand_off_err) ctrl->pad = 1;  // the register-resident form ran out of a poll budget
    if (ctrl->active) ctrl->active = (any && ctrl->iters < max_it && !*hand_off_err) ? 1 : 0;
  }
}

template <typename T, bool JAC>
__global__ __launch_bounds__(256) void d1_tile_kernel(const MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ A, long n,
                                                      const T* __restrict__ r, const T* __restrict__ dinv,
                                                      const T* __restrict__ p_old, T* __restrict__ p_new,
                                                      const T* __restrict__ cpart, T* __restrict__ scal, int k,
                                                      const int2* __restrict__ tab, T* __restrict__ Q,
                                                      T* __restrict__ tpart, T thr, T min_float, int max_it,
                                                      int* __restrict__ stopw) {
  // Gate: `active` is written by finish kernels only, stopw[(k-1) & 1] by the tile kernel of the PREVIOUS iteration --
  // nothing this launch writes (block 0 used to clear ctrl->active while other workgroups of the same
  // launch read it at entry).  A shut launch hands the shut on, so every later launch of the batch stays shut.
  if (ctrl->active == 0 || stopw[(k + 1) & 1] != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) stopw[k & 1] = 1;
    return;
  }
  constexpr int TS = 64;
  __shared__ T colp[4][TS];
  __shared__ T wsum[4];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const long b = blockIdx.x;
  const int2 ij = tab[b];  // uniform: a scalar load
  const int I = ij.x, J = ij.y;
  const long r0 = (long)I * TS + 16 * w, c = (long)J * TS + l, ci = (long)I * TS + l;
  // Issue order matters: vmcnt retires in order, so whatever is requested BEHIND the tile cannot be used before
  // the tile has landed.  The operands of the recurrence (L2-resident, ten loads) go first, the 16 tile loads
  // behind them, and beta and p_k are formed while the tile is in flight.  No branch anywhere between the loads
  // and their uses -- ragged edges by clamped addresses and selects, the Jacobi diagonal by a template parameter,
  // a converged solve runs this kernel once more with its stores switched off: the compiler sinks a load into the
  // conditional block that uses it, and at a join of two paths its wait counts are the pessimistic merge (the
  // first version waited for 12 of the 16 tile loads before it touched the shares).
  const T z0 = cpart[l], z1 = cpart[64 + l], q0 = cpart[CP + l], q1 = cpart[CP + 64 + l];
  const long cj = c < n ? c : n - 1, cic = ci < n ? ci : n - 1;
  const T rj = r[cj], poj = p_old[cj], ri = r[cic], poi = p_old[cic];
  const T dj = JAC ? dinv[cj] : (T)1, di = JAC ? dinv[cic] : (T)1;
  const T rz_old = scal[k & 1];
  const int it = ctrl->iters;
  __builtin_amdgcn_sched_barrier(0);
  T a[16];
  {
    // row q of the wave's 16: one pointer bump per row, frozen at the last row of the matrix (no 64-bit multiply
    // per load)
    const T* row = A + (r0 < n ? r0 : n - 1) * n + cj;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      a[q] = *row;
      row += (r0 + q + 1 < n) ? n : 0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = (r0 + q < n && c < n) ? a[q] : (T)0;
  const T rz_new = wave_allsum(z0 + z1), rr_new = wave_allsum(q0 + q1);
  // the stopping rule (:59-62): the same decision in every workgroup; `live` switches the stores off
  const bool live = (T)0.5 * rr_new > thr && it < max_it;
  if (b == 0 && t == 0) {
    if (live) scal[(k + 1) & 1] = rz_new;  // U_k's numerator of gamma, T_{k+1}'s rz_old
    stopw[k & 1] = live ? 0 : 1;           // read by U_k and T_{k+1}, one launch later each
  }
  const bool drop = rz_old <= min_float;  // :79; also how the first direction p_1 = z_0 comes out
  const T beta = drop ? (T)0 : rz_new / rz_old;
  const T zj = JAC ? rj * dj : rj, zi = JAC ? ri * di : ri;
  T pj = drop ? zj : mgp_fma(beta, poj, zj);  // a select, not 0 * p_old: the arena may hold anything at start-up
  T pi = drop ? zi : mgp_fma(beta, poi, zi);
  pj = c < n ? pj : (T)0;
  pi = ci < n ? pi : (T)0;
  if (live && I == J && w == 0 && c < n) p_new[c] = pj;
  T x[16];
  T cs = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    x[q] = a[q] * pj;
    cs = mgp_fma(a[q], mgp_read_lane(pi, 16 * w + q), cs);
  }
  colp[w][l] = cs;
  rs_step<T, 8, 32>(x, l);
  rs_step<T, 4, 16>(x, l);
  rs_step<T, 2, 8>(x, l);
  rs_step<T, 1, 4>(x, l);
  T s = x[0];
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);  // lane l: (A_IJ p_J)[16 w + (l >> 2)]
  const long i = r0 + (l >> 2);
  if (live && (l & 3) == 0 && i < n) Q[(long)J * n + i] = s;
  // the tile's share of p . A p: rows of this wave, then the four waves in order
  const T prow = __shfl(pi, 16 * w + (l >> 2), 64);
  const T u = wave_allsum((l & 3) == 0 ? s * prow : (T)0);
  if (l == 0) wsum[w] = u;
  __syncthreads();
  if (live && t < TS && I != J) {
    const T sc = (colp[0][t] + colp[1][t]) + (colp[2][t] + colp[3][t]);
    const long ic2 = (long)J * TS + t;
    if (ic2 < n) Q[(long)I * n + ic2] = sc;
  }
  if (live && t == 0) {
    const T tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    tpart[b] = I == J ? tot : tot + tot;
  }
}

template <typename T, bool JAC, int NT, int PER, int TPM>
__global__ __launch_bounds__(NT) void d1_update_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ Q, int nt,
                                                       const T* __restrict__ tpart, long ntiles,
                                                       const T* __restrict__ scal, int k, const T* __restrict__ p,
                                                       T* __restrict__ v, T* __restrict__ r,
                                                       const T* __restrict__ dinv, T* __restrict__ cpart, long n,
                                                       T min_float, const int* __restrict__ stopw) {
  if (ctrl->active == 0 || stopw[k & 1] != 0) return;  // T_k (the previous launch) found the solve finished
  constexpr int NW = NT / 64;
  __shared__ T part[NW][64];
  __shared__ T red[NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long c = blockIdx.x;
  const long i = c * 64 + lane;
  const long ic = i < n ? i : n - 1;
  // every load of the kernel is issued here, clamped instead of guarded
  const int kb = wave * PER;
  T sl[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int kk = kb + q < nt ? kb + q : nt - 1;
    sl[q] = Q[(long)kk * n + ic];
  }
  T tp[TPM];
#pragma unroll
  for (int m = 0; m < TPM; ++m) {
    const long e = (long)m * NT + t;
    tp[m] = tpart[e < ntiles ? e : ntiles - 1];
  }
  const T pc = p[ic], rc = r[ic], vc = v[ic];
  const T dc = JAC ? dinv[ic] : (T)1;
  const T rz_prev = scal[(k + 1) & 1];
  // p . A p: thread-sequential over its shares, lanes by butterfly, waves in order -- the same bits in every workgroup
  T d = 0;
#pragma unroll
  for (int m = 0; m < TPM; ++m) d += ((long)m * NT + t < ntiles) ? tp[m] : (T)0;
  d = wave_allsum(d);
  if (lane == 0) red[wave] = d;
  T s = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) s += (kb + q < nt) ? sl[q] : (T)0;
  part[wave][lane] = s;
  __syncthreads();
  if (wave != 0) return;
  d = red[0];
#pragma unroll
  for (int q = 1; q < NW; ++q) d += red[q];
  const T gamma = (d <= min_float) ? (T)0 : rz_prev / d;  // :66-68
  T a = part[0][lane];
#pragma unroll
  for (int q = 1; q < NW; ++q) a += part[q][lane];
  const T vn = mgp_fma(gamma, pc, vc);   // :69
  const T rn = mgp_fma(-gamma, a, rc);   // :76
  const T zn = JAC ? rn * dc : rn;       // :77
  const bool ok = i < n;
  if (ok) {
    v[i] = vn;
    r[i] = rn;
  }
  const T prz = wave_allsum(ok ? zn * rn : (T)0), prr = wave_allsum(ok ? rn * rn : (T)0);
  if (lane == 0) {
    cpart[c] = prz;
    cpart[CP + c] = prr;
    if (c == 0) ctrl->iters = ctrl->iters + 1;
  }
}

// write-through store / L1-bypassing load of one element (global_store/load ... sc1): the two halves of a hand-off
// between resident workgroups without fences (MI355X_MICROARCH.md, hand-off table)
// ---- cross-lane moves on the vector ALU (gfx950): `__shfl_xor` lowers to ds_bpermute -- the LDS crossbar, ~100
// cycles of latency per dependent step -- which the tile kernels above hide behind seven workgroups per CU; the
// register-resident kernel runs three waves per SIMD, so its reduction chains use DPP (within a row of 16 lanes) and
// v_permlane16_swap / v_permlane32_swap (between rows / halves) instead.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// value of lane l ^ M for M = 1, 2, 4, 8 (inside a row of 16 lanes)
template <int M, typename T>
__device__ __forceinline__ T lane_xor_row(T v) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8, "row-local masks");
  if (M == 1) return dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  if (M == 2) return dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  if (M == 8) return dpp_mov<0x128>(v);  // row_ror:8
  return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror (l -> l ^ 7) then quad_perm [3,2,1,0] (l -> l ^ 3)
}
// (a', b') = swap: HALF32: a.hi <-> b.lo (lanes 32..63 of a with lanes 0..31 of b); else rows: a.row1 <-> b.row0, a.row3 <-> b.row2
template <bool HALF32>
__device__ __forceinline__ void lane_swap(double& a, double& b) {
  int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  if (HALF32) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    alo = r0[0], blo = r0[1], ahi = r1[0], bhi = r1[1];
  } else {
    const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    alo = r0[0], blo = r0[1], ahi = r1[0], bhi = r1[1];
  }
  a = __hiloint2double(ahi, alo);
  b = __hiloint2double(bhi, blo);
}
template <bool HALF32>
__device__ __forceinline__ void lane_swap(float& a, float& b) {
  int ai = __float_as_int(a), bi = __float_as_int(b);
  if (HALF32) {
    const auto r = __builtin_amdgcn_permlane32_swap(ai, bi, false, false);
    ai = r[0], bi = r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(ai, bi, false, false);
    ai = r[0], bi = r[1];
  }
  a = __int_as_float(ai);
  b = __int_as_float(bi);
}
// reduce-scatter step across lane bit 32 (HALF32) or 16: lanes with the bit clear keep `lo` and receive the partner's
// `lo`, lanes with it set keep `hi` and receive the partner's `hi` -- after the swap both are a' + b'
template <bool HALF32, typename T>
__device__ __forceinline__ T rs_swap_add(T lo, T hi) {
  lane_swap<HALF32>(lo, hi);
  return lo + hi;
}
// sum over the 64 lanes, the same bits in every lane (each step adds the two partners' values, a + b == b + a)
template <typename T>
__device__ __forceinline__ T wave_allsum_valu(T v) {
  v = rs_swap_add<true>(v, v);
  v = rs_swap_add<false>(v, v);
  v += lane_xor_row<8>(v);
  v += lane_xor_row<4>(v);
  v += lane_xor_row<2>(v);
  v += lane_xor_row<1>(v);
  return v;
}
// reduce-scatter steps inside a row of 16 lanes (BIT = 8 or 4): x[k] <- keep + partner's send, HALF sums stay
template <typename T, int HALF, int BIT>
__device__ __forceinline__ void rs_step_row(T (&x)[16], int l) {
  const bool hi = (l & BIT) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const T keep = hi ? x[k + HALF] : x[k];
    const T send = hi ? x[k] : x[k + HALF];
    x[k] = keep + lane_xor_row<BIT>(send);
  }
}

// ------------------------------------------------------------------ round 4: 2 .. 8 right-hand sides on the tile scheme
// The reference's default probe count is 5 (`CGGP(num_probes=5)`, cggp/models.py:286): every `prior_kl` and
// `eval_logdet` of a training step is a 5-column CG on Kmm + Lambda.  Round 3 ran those through the skinny MFMA
// product (the whole matrix read, a third of the matrix cores' columns used) + the fused update: 35 us per iteration
// at n = 4096.  Here the two-launch iteration above carries BT columns: T reads each upper-triangle tile ONCE and
// forms, per column, the direction on the fly, both tile products and the share of p.Ap; U -- one workgroup per
// (chunk, column) -- is d1_update_kernel with a column index.  Same recurrence per column (:64-85), the reference's
// `any` over the columns as stopping rule (:59-62), the same guards per column (:68, :79), sums in fixed order.
template <typename T>
__global__ __launch_bounds__(64) void d1m_init_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ b,
                                                      const T* __restrict__ av, T* __restrict__ r,
                                                      const T* __restrict__ dinv, T* __restrict__ cpart,
                                                      T* __restrict__ scal, T* __restrict__ zpub, long n,
                                                      int* __restrict__ stopw) {
  const int l = threadIdx.x, col = blockIdx.y;
  const long i = (long)blockIdx.x * 64 + l, off = (long)col * n;
  T rv = 0, zv = 0;
  if (i < n) {
    rv = av ? b[off + i] - av[off + i] : b[off + i];
    r[off + i] = rv;
    zv = dinv ? rv * dinv[i] : rv;
    zpub[off + i] = zv;  // z_0 for the register-resident form
  }
  const T prz = wave_allsum(zv * rv), prr = wave_allsum(rv * rv);
  if (l == 0) {
    T* cp = cpart + (long)col * 2 * CP;
    cp[blockIdx.x] = prz;
    cp[CP + blockIdx.x] = prr;
    if (blockIdx.x == 0) {
      scal[2 * col] = 0;
      scal[2 * col + 1] = 0;
      if (col == 0) {
        ctrl->active = 1;
        ctrl->iters = 0;
        ctrl->ticket = 0;
        ctrl->pad = 0;
        stopw[0] = 0;
        stopw[1] = 0;
      }
    }
  }
}

// statistics per column + the gate for the host poll: one wave per column
template <typename T>
__global__ __launch_bounds__(512) void d1m_finish_kernel(MgpCgCtrl* __restrict__ ctrl,
                                                         const int* __restrict__ hand_off_err,
                                                         const T* __restrict__ cpart, T* __restrict__ rz,
                                                         T* __restrict__ err, int* __restrict__ over, T thr,
                                                         int max_it, int bt) {
  __shared__ int any_s[8];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (w < bt) {
    T s_rz, s_rr;
    sum_shares(cpart + (long)w * 2 * CP, l, s_rz, s_rr);
    if (l == 0) {
      rz[w] = s_rz;
      err[w] = (T)0.5 * s_rz;
      const int a = ((T)0.5 * s_rr > thr) ? 1 : 0;
      over[w] = a;
      any_s[w] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int any = 0;
    for (int q = 0; q < bt; ++q) any |= any_s[q];
    if (*hand_off_err) ctrl->pad = 1;  // the register-resident form ran out of a poll budget
    if (ctrl->active) ctrl->active = (any && ctrl->iters < max_it && !*hand_off_err) ? 1 : 0;
  }
}

template <typename T, bool JAC, int BT>
__global__ __launch_bounds__(256) void d1m_tile_kernel(const MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ A,
                                                       long n, const T* __restrict__ r, const T* __restrict__ dinv,
                                                       const T* __restrict__ p_old, T* __restrict__ p_new,
                                                       const T* __restrict__ cpart, T* __restrict__ scal, int k,
                                                       const int2* __restrict__ tab, T* __restrict__ Q, int nt,
                                                       T* __restrict__ tpart, long ntiles, T thr, T min_float,
                                                       int max_it, int* __restrict__ stopw) {
  if (ctrl->active == 0 || stopw[(k + 1) & 1] != 0) {  // see d1_tile_kernel
    if (blockIdx.x == 0 && threadIdx.x == 0) stopw[k & 1] = 1;
    return;
  }
  constexpr int TS = 64;
  __shared__ T colp[BT][4][TS];
  __shared__ T wsum[BT][4];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const long b = blockIdx.x;
  const int2 ij = tab[b];
  const int I = ij.x, J = ij.y;
  const long r0 = (long)I * TS + 16 * w, c = (long)J * TS + l, ci = (long)I * TS + l;
  const long cj = c < n ? c : n - 1, cic = ci < n ? ci : n - 1;
  // the recurrence's operands of every column first (L2-resident), the tile's 16 loads behind them (d1_tile_kernel)
  T z0[BT], z1[BT], q0[BT], q1[BT], rj[BT], poj[BT], ri[BT], poi[BT], rzo[BT];
#pragma unroll
  for (int e = 0; e < BT; ++e) {
    const T* cp = cpart + (long)e * 2 * CP;
    z0[e] = cp[l], z1[e] = cp[64 + l], q0[e] = cp[CP + l], q1[e] = cp[CP + 64 + l];
    rj[e] = r[(long)e * n + cj], poj[e] = p_old[(long)e * n + cj];
    ri[e] = r[(long)e * n + cic], poi[e] = p_old[(long)e * n + cic];
    rzo[e] = scal[2 * e + (k & 1)];
  }
  const T dj = JAC ? dinv[cj] : (T)1, di = JAC ? dinv[cic] : (T)1;
  const int it = ctrl->iters;
  __builtin_amdgcn_sched_barrier(0);
  T a[16];
  {
    const T* row = A + (r0 < n ? r0 : n - 1) * n + cj;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      a[q] = *row;
      row += (r0 + q + 1 < n) ? n : 0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // per column: rz, ||r||^2 of the current residual, beta, the direction's entries -- while the tile is in flight
  T pj[BT], pi[BT], rzn[BT];
  bool any = false;
#pragma unroll
  for (int e = 0; e < BT; ++e) {
    rzn[e] = wave_allsum_valu(z0[e] + z1[e]);  // the same bits as wave_allsum (commutative butterfly)
    const T rr_new = wave_allsum_valu(q0[e] + q1[e]);
    any = any || (T)0.5 * rr_new > thr;
  }
  const bool live = any && it < max_it;  // :59-62: `any` over the columns
  if (b == 0 && t == 0) stopw[k & 1] = live ? 0 : 1;
#pragma unroll
  for (int e = 0; e < BT; ++e) {
    if (b == 0 && t == 0 && live) scal[2 * e + ((k + 1) & 1)] = rzn[e];
    const bool drop = rzo[e] <= min_float;  // :79, per column
    const T beta = drop ? (T)0 : rzn[e] / rzo[e];
    const T zj = JAC ? rj[e] * dj : rj[e], zi = JAC ? ri[e] * di : ri[e];
    T vj = drop ? zj : mgp_fma(beta, poj[e], zj);
    T vi = drop ? zi : mgp_fma(beta, poi[e], zi);
    pj[e] = c < n ? vj : (T)0;
    pi[e] = ci < n ? vi : (T)0;
    if (live && I == J && w == 0 && c < n) p_new[(long)e * n + c] = pj[e];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = (r0 + q < n && c < n) ? a[q] : (T)0;
#pragma unroll
  for (int e = 0; e < BT; ++e) {
    // Cross-lane steps on the vector ALU (permlane swaps / DPP), not ds_bpermute: with several columns per tile the LDS
    // crossbar was the bound -- 72 bpermutes per tile, wave and column, ~7 us per column at n = 4096 against ~5 us this
    // way (alternating the columns between the two pipes was measured too: no better).  The additions are the same.
    T x[16];
    T cs = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      cs = mgp_fma(a[q], mgp_read_lane(pi[e], 16 * w + q), cs);
      x[q] = rs_swap_add<true>(a[q] * pj[e], a[q + 8] * pj[e]);  // rows q and q + 8 meet across lane bit 32
    }
#pragma unroll
    for (int q = 8; q < 16; ++q) cs = mgp_fma(a[q], mgp_read_lane(pi[e], 16 * w + q), cs);
    colp[e][w][l] = cs;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = rs_swap_add<false>(x[q], x[q + 4]);  // lane bit 16
    rs_step_row<T, 2, 8>(x, l);
    rs_step_row<T, 1, 4>(x, l);
    T s = x[0];
    s += lane_xor_row<2>(s);
    s += lane_xor_row<1>(s);  // lane l: (A_IJ p_J)[16 w + (l >> 2)]
    const long i = r0 + (l >> 2);
    if (live && (l & 3) == 0 && i < n) Q[((long)e * nt + J) * n + i] = s;
    const T prow = __shfl(pi[e], 16 * w + (l >> 2), 64);
    const T u = wave_allsum_valu((l & 3) == 0 ? s * prow : (T)0);
    if (l == 0) wsum[e][w] = u;
  }
  __syncthreads();
  if (live && I != J) {
    // column sums: BT x 64 outputs over the 256 threads
    for (int o = t; o < BT * TS; o += 256) {
      const int e = o >> 6, cc = o & 63;
      const T sc = (colp[e][0][cc] + colp[e][1][cc]) + (colp[e][2][cc] + colp[e][3][cc]);
      const long ic2 = (long)J * TS + cc;
      if (ic2 < n) Q[((long)e * nt + I) * n + ic2] = sc;
    }
  }
  if (live && t < BT) {
    const T tot = (wsum[t][0] + wsum[t][1]) + (wsum[t][2] + wsum[t][3]);
    tpart[(long)t * ntiles + b] = I == J ? tot : tot + tot;
  }
}

// U for column blockIdx.y: d1_update_kernel's arithmetic on that column's slots, shares and vectors
template <typename T, bool JAC, int NT, int PER, int TPM>
__global__ __launch_bounds__(NT) void d1m_update_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ Q, int nt,
                                                        const T* __restrict__ tpart, long ntiles,
                                                        const T* __restrict__ scal, int k, const T* __restrict__ p,
                                                        T* __restrict__ v, T* __restrict__ r,
                                                        const T* __restrict__ dinv, T* __restrict__ cpart, long n,
                                                        T min_float, const int* __restrict__ stopw) {
  if (ctrl->active == 0 || stopw[k & 1] != 0) return;
  constexpr int NW = NT / 64;
  __shared__ T part[NW][64];
  __shared__ T red[NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = blockIdx.y;
  const long c = blockIdx.x, off = (long)col * n;
  const long i = c * 64 + lane;
  const long ic = i < n ? i : n - 1;
  const T* Qc = Q + (long)col * nt * n;
  const T* tpc = tpart + (long)col * ntiles;
  const int kb = wave * PER;
  T sl[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int kk = kb + q < nt ? kb + q : nt - 1;
    sl[q] = Qc[(long)kk * n + ic];
  }
  T tp[TPM];
#pragma unroll
  for (int m = 0; m < TPM; ++m) {
    const long e = (long)m * NT + t;
    tp[m] = tpc[e < ntiles ? e : ntiles - 1];
  }
  const T pc = p[off + ic], rc = r[off + ic], vc = v[off + ic];
  const T dc = JAC ? dinv[ic] : (T)1;
  const T rz_prev = scal[2 * col + ((k + 1) & 1)];
  T d = 0;
#pragma unroll
  for (int m = 0; m < TPM; ++m) d += ((long)m * NT + t < ntiles) ? tp[m] : (T)0;
  d = wave_allsum(d);
  if (lane == 0) red[wave] = d;
  T s = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) s += (kb + q < nt) ? sl[q] : (T)0;
  part[wave][lane] = s;
  __syncthreads();
  if (wave != 0) return;
  d = red[0];
#pragma unroll
  for (int q = 1; q < NW; ++q) d += red[q];
  const T gamma = (d <= min_float) ? (T)0 : rz_prev / d;  // :66-68
  T a = part[0][lane];
#pragma unroll
  for (int q = 1; q < NW; ++q) a += part[q][lane];
  const T vn = mgp_fma(gamma, pc, vc);   // :69
  const T rn = mgp_fma(-gamma, a, rc);   // :76
  const T zn = JAC ? rn * dc : rn;       // :77
  const bool ok = i < n;
  if (ok) {
    v[off + i] = vn;
    r[off + i] = rn;
  }
  const T prz = wave_allsum(ok ? zn * rn : (T)0), prr = wave_allsum(ok ? rn * rn : (T)0);
  if (lane == 0) {
    T* cp = cpart + (long)col * 2 * CP;
    cp[c] = prz;
    cp[CP + c] = prr;
    if (c == 0 && col == 0) ctrl->iters = ctrl->iters + 1;
  }
}

// workgroup barrier that orders LDS only (no wait for vector-memory operations in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T ld_sc1(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_sc1(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

using gu64 = unsigned long long;

// ------------------------------------------------------------------ round 4: the matrix stays ON THE CHIP
// For n <= 4096 the upper triangle of A (67 MB of 64 x 64 tiles at n = 4096) fits the register files of the chip
// (256 CUs x 512 KB), for n <= 2048 the whole matrix: one launch loads every tile ONCE (d1_persist_blk_kernel: nine
// tiles per workgroup of 768 threads, two per four-wave group in registers and one in LDS; d1_persist_full_kernel: four
// per workgroup of 1024 threads) and then runs the WHOLE solve, conjugate_gradient.py:59-98, without touching A again.
// What is left of an iteration is its two global reductions, as hand-offs between resident workgroups in the form of
// cdna_hip_programming.md Guideline 16, R2 --
// THE DATA IS THE FLAG: every published number travels as 8-byte granules {epoch, 32 bits of the value}, each written
// by ONE write-through (sc1) store and re-read with sc1 loads until its tag is the epoch the reader waits for.
// No flag word, no fence, no atomic, no drain-then-signal: a consumer's poll IS its data load, so a phase costs one
// memory round trip after the last producer's store has landed (the flag form measured 13.3 us per iteration at
// n = 4096: four dependent round trips of ~1.5 us each).
//   A_k  every workgroup -> the chunk owners: its partial vectors of A p and one share of p.Ap, epoch k + 1
//   B_k  the chunk owners -> every workgroup: z = M^-1 r of the chunk and its shares of rz, ||r||^2, epoch k + 2
//        (epoch 1 = the initial residual, written by d1_persist_seed_kernel)
// Every workgroup forms the stopping rule (:59-62) and beta from the same 2 nt shares, so all leave the loop on the
// same iteration with no further word exchanged.  A buffer is rewritten only after every reader has published
// something that depends on having read it (partial vectors and shares of p.Ap: read before an owner publishes B; z and
// the chunk shares: read before a workgroup publishes A), so single buffers suffice and a reader never meets a later
// epoch.  Every wait is bounded (a poll budget, then the error word and out): a workgroup that is not resident --
// another stream holding CUs -- makes the solve fail over to the two-launch form, never hang.  Who polls matters: a
// wave polls only what it has a duty for, everybody else waits at the workgroup barrier (3072 waves re-reading all
// shares every round trip were ~10 TB/s of write-through-line reads, 15.5 us per iteration).  Loop-carried per-wave
// state lives in LDS, not registers: a spill reload forces `s_waitcnt vmcnt(0)`, i.e. a wait for the wave's own
// write-through stores.
constexpr int kPersistBudget = 1 << 16;  // polls of >= 1 us each

// The granule region of a solve as ONE buffer resource: a 16-byte `buffer_load/store_dwordx4 ... sc1` moves both
// granules of an fp64 element in one instruction (8-byte atomic loads were the bound of the polls: 61 KB of them into
// one CU per round at five columns); each 8-byte half still carries, and is checked by, its own tag.
typedef unsigned d1_v4u __attribute__((ext_vector_type(4)));
typedef unsigned d1_v2u __attribute__((ext_vector_type(2)));
struct GranRs {
  __amdgpu_buffer_rsrc_t rs;
  const gu64* base;
};
__device__ __forceinline__ GranRs make_gran_rs(const gu64* base, long bytes) {
  GranRs g;
  g.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)bytes, 0x00020000);
  g.base = base;
  return g;
}

template <typename T>
struct Gran;  // element <-> granules
template <>
struct Gran<double> {
  static constexpr int W = 2;  // granules per element
  static __device__ __forceinline__ void store(const GranRs& R, gu64* g, unsigned epoch, double v) {
    const d1_v4u x = {(unsigned)__double2loint(v), epoch, (unsigned)__double2hiint(v), epoch};
    __builtin_amdgcn_raw_buffer_store_b128(x, R.rs, (int)((const char*)g - (const char*)R.base), 0, 16);  // aux 16 = sc1
  }
  static __device__ __forceinline__ bool load(const GranRs& R, const gu64* g, unsigned epoch, double& v) {
    const d1_v4u x = __builtin_amdgcn_raw_buffer_load_b128(R.rs, (int)((const char*)g - (const char*)R.base), 0, 16);
    v = __hiloint2double((int)x.z, (int)x.x);
    return x.y == epoch && x.w == epoch;
  }
  // the two halves of `load`, for polls that issue all their loads before looking at any tag (poll_units)
  using Raw = d1_v4u;
  static constexpr int kBytes = 16;
  static __device__ __forceinline__ Raw raw(const GranRs& R, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(R.rs, voff, soff, 16);
  }
  static __device__ __forceinline__ bool unpack(const Raw& x, unsigned epoch, double& v) {
    v = __hiloint2double((int)x.z, (int)x.x);
    return (x.y == epoch) & (x.w == epoch);
  }
};
template <>
struct Gran<float> {
  static constexpr int W = 1;
  static __device__ __forceinline__ void store(const GranRs& R, gu64* g, unsigned epoch, float v) {
    const d1_v2u x = {(unsigned)__float_as_int(v), epoch};
    __builtin_amdgcn_raw_buffer_store_b64(x, R.rs, (int)((const char*)g - (const char*)R.base), 0, 16);
  }
  static __device__ __forceinline__ bool load(const GranRs& R, const gu64* g, unsigned epoch, float& v) {
    const d1_v2u x = __builtin_amdgcn_raw_buffer_load_b64(R.rs, (int)((const char*)g - (const char*)R.base), 0, 16);
    v = __int_as_float((int)x.x);
    return x.y == epoch;
  }
  using Raw = d1_v2u;
  static constexpr int kBytes = 8;
  static __device__ __forceinline__ Raw raw(const GranRs& R, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(R.rs, voff, soff, 16);
  }
  static __device__ __forceinline__ bool unpack(const Raw& x, unsigned epoch, float& v) {
    v = __int_as_float((int)x.x);
    return x.y == epoch;
  }
};

// A poll of U elements per lane: every round issues its U loads BACK TO BACK and only then looks at the tags.  With a
// branch between two loads (a guard per load) the compiler must end each basic block with `s_waitcnt vmcnt(0)`, and a
// poll of U elements costs U dependent round trips instead of one -- measured: 2.5-3 us per right-hand side.  So a
// load that is not needed (a lane beyond n, a unit beyond the count) is issued all the same, at an offset inside the
// granule region, and only its tag is ignored.  Element i is at byte vo[i] (per lane) + so[i] (wave-uniform) of the
// region.  Bounded; returns false when the budget ran out.  Elements not needed come back as 0.
// ... with offsets and needs given as functions of (unit, an opaque zero renewed every round): nothing about the units
// is loop-invariant to the compiler, so it keeps neither U offsets nor U lane masks in registers across the rounds
// (fourteen units per wave at five and six columns: the difference between spilling and not)
template <typename T, int U, typename VoFn, typename SoFn, typename NeedFn>
__device__ __forceinline__ bool poll_units_fn(const GranRs& R, VoFn vo, SoFn so, NeedFn need, unsigned epoch, T (&val)[U]) {
  bool ok = false;
  unsigned long long needed = 0;  // per lane: bit i = unit i was needed (for the zeroing below)
  for (int spin = 0; !ok && spin < kPersistBudget; ++spin) {
    int z = 0;
    asm volatile("" : "+s"(z));
    typename Gran<T>::Raw raw[U];
#pragma unroll
    for (int i = 0; i < U; ++i) raw[i] = Gran<T>::raw(R, vo(i, z), __builtin_amdgcn_readfirstlane(so(i, z)));
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < U; ++i) asm volatile("" : "+v"(raw[i]));
    bool mine = true;
    unsigned nd = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool nd_i = need(i, z);
      mine = mine & (Gran<T>::unpack(raw[i], epoch, val[i]) | !nd_i);
      nd |= nd_i ? (1u << i) : 0u;
    }
    needed = nd;
    ok = __builtin_amdgcn_ballot_w64(mine) == __builtin_amdgcn_ballot_w64(true);
    if (!ok) __builtin_amdgcn_s_sleep(2);
  }
#pragma unroll
  for (int i = 0; i < U; ++i) val[i] = ((needed >> i) & 1) ? val[i] : (T)0;
  return ok;
}

// one round of a poll's loads / its tag tests (see poll_units).  The scheduler must not interleave the loads with the
// tag tests either (it does, to save registers): no load moves below the `memory` statement, no use of a loaded value
// above the statements that name it
template <typename T, int U>
__device__ __forceinline__ void poll_issue(const GranRs& R, const int (&vo)[U], const int (&so)[U],
                                           typename Gran<T>::Raw (&raw)[U]) {
#pragma unroll
  for (int i = 0; i < U; ++i) raw[i] = Gran<T>::raw(R, vo[i], __builtin_amdgcn_readfirstlane(so[i]));
  asm volatile("" ::: "memory");
}
template <typename T, int U>
__device__ __forceinline__ bool poll_test(typename Gran<T>::Raw (&raw)[U], const bool (&need)[U], unsigned epoch, T (&val)[U]) {
#pragma unroll
  for (int i = 0; i < U; ++i) asm volatile("" : "+v"(raw[i]));
  bool mine = true;
#pragma unroll
  for (int i = 0; i < U; ++i) mine = mine & (Gran<T>::unpack(raw[i], epoch, val[i]) | !need[i]);
  return __builtin_amdgcn_ballot_w64(mine) == __builtin_amdgcn_ballot_w64(true);
}
template <typename T, int U>
__device__ __forceinline__ bool poll_units(const GranRs& R, const int (&vo)[U], const int (&so)[U], const bool (&need)[U],
                                           unsigned epoch, T (&val)[U]) {
  // (an idle unit must not read one fixed address -- a thousand waves' idle units on ONE line were a hot spot worth
  // 2 us per iteration: the callers point it at lane- and wave-distinct elements of z); a wave without any needed unit
  // does not poll at all
  bool any = false;
#pragma unroll
  for (int i = 0; i < U; ++i) any = any | need[i];
  bool ok = __builtin_amdgcn_ballot_w64(any) == 0;
  for (int spin = 0; !ok && spin < kPersistBudget; ++spin) {
    typename Gran<T>::Raw raw[U];
    poll_issue<T, U>(R, vo, so, raw);
    ok = poll_test<T, U>(raw, need, epoch, val);
    if (!ok) __builtin_amdgcn_s_sleep(2);
  }
#pragma unroll
  for (int i = 0; i < U; ++i) val[i] = need[i] ? val[i] : (T)0;
  return ok;
}

// granule arrays of one solve (zeroed by mgp_dense1_begin: a tag left by an earlier solve must never match)
struct D1PBuf {
  gu64* Qg;    // [nt][n] elements: slot k of output element i
  gu64* wpg;   // [256] elements: the workgroups' shares of p.Ap
  gu64* cg;    // [2][64] elements: the chunks' shares of rz and of ||r||^2
  gu64* zg;    // [n] elements: z = M^-1 r
  int* err;
  long bytes;  // of the whole granule region, which begins at Qg
};

// epoch-1 granules of the initial residual (d1_init_kernel left z_0 in zpub and the shares in cpart)
template <typename T>
__global__ __launch_bounds__(64) void d1_persist_seed_kernel(const T* __restrict__ zpub, const T* __restrict__ cpart,
                                                             D1PBuf pb, long n, int nt) {
  constexpr int W = Gran<T>::W;
  const GranRs grs = make_gran_rs(pb.Qg, pb.bytes);
  const int l = threadIdx.x, col = blockIdx.y;
  const long i = (long)blockIdx.x * 64 + l;
  if (i < n) Gran<T>::store(grs, pb.zg + ((long)col * n + i) * W, 1u, zpub[(long)col * n + i]);
  if (blockIdx.x == 0 && l < nt) {
    const T* cp = cpart + (long)col * 2 * CP;
    Gran<T>::store(grs, pb.cg + ((long)col * 128 + l) * W, 1u, cp[l]);
    Gran<T>::store(grs, pb.cg + ((long)col * 128 + 64 + l) * W, 1u, cp[CP + l]);
  }
}

// ------------------------------------------------------------------ the triangle in 3 x 3 SUPER-BLOCKS of tiles
// Dealt out round-robin (the first form of this kernel, 12.3 us per iteration at n = 4096 against 7.4 for this one in
// the same run, profiles/r04_ab_dense1_super_blocks_first.txt), a workgroup's nine tiles touch eighteen chunks of p,
// every tile publishes two 64-vectors (nt = 64 per chunk for the owner to read back), and each tile pays its own
// cross-lane reduction.  Here workgroup (SI, SJ), SI <= SJ, holds the 3 x 3 tiles (3 SI + g, 3 SJ + ly): S =
// ceil(nt / 3) <= 22 super-rows, S (S + 1) / 2 <= 253 workgroups -- the chip's 256 CUs, nine tiles each, as before.
// Four-wave group g holds tile row I_g = 3 SI + g (wave wq its rows 16 wq .. 16 wq + 15), layer ly the tile column
// J_ly = 3 SJ + ly (two layers in registers, the third in LDS).  Then
//   * a workgroup needs SIX chunks of p (three when SI == SJ), not eighteen: a third of the z polls;
//   * the row products of a wave's three tiles are added lane-locally BEFORE the cross-lane reduce-scatter -- one
//     reduction per wave and column instead of three -- and the workgroup publishes ONE vector per tile row:
//     sum_ly A_{I_g J_ly} p_{J_ly};
//   * the column products (lane-local) of a tile column are summed over the workgroup's 12 waves in LDS and published
//     as ONE vector per tile column: sum_g A_{I_g J_ly}^T p_{I_g} (diagonal tiles only in the row product);
//   * chunk c of super-row s therefore collects S + 1 vectors (slot j < s: columns of block (j, s); j = s: rows of
//     (s, s); s < j < S: rows of (s, j); j = S: columns of the diagonal block (s, s)) instead of nt = 64;
//   * a further right-hand side costs 96 fused multiply-adds and one reduction per wave: BT <= 8 columns (the
//     reference's 5 probes, models.py:286, at M = 4096; 7 and 8 from S = 12 on, with one column buffer), per column its
//     own recurrence (:64-85) and guards (:68, :79), `any` over the columns as stopping rule (:59-62).
// The owner of chunk c = 3 s + pos (one column) is a workgroup that holds p_c anyway: block (s, s + pos) (as a row
// chunk), or, where s + pos >= S, block (s + pos - S, s) (as a column chunk) -- one chunk per workgroup for S >= 6; with
// several columns the owners of a chunk's columns are different such workgroups (see "ownership" in the kernel).
// Hand-offs, epochs, bounds and fail-over as described above.  Granule arrays per column e: cg + e 128 W, zg + e n W, wpg + e 256 W,
// Qg + e (nt (S + 1) 64) W with slot j of chunk c at ((c (S + 1) + j) 64) W.
template <typename T, int BT>
struct D1Blk {
  static constexpr size_t kTile = (size_t)3 * 64 * 64 * sizeof(T);
  static constexpr size_t kColp = (size_t)3 * 12 * 64 * sizeof(T);  // one column's partial column products
  static constexpr size_t kStatic = (size_t)(6 * BT + 2 * 6) * 64 * sizeof(T) + 1536;  // p of six chunks, r / v of <= 6 owned items, small words
  static constexpr int kColBuf = kTile + 2 * kColp + kStatic <= (size_t)160 * 1024 ? 2 : 1;
  static constexpr size_t kDyn = kTile + kColBuf * kColp;
};

template <typename T, bool JAC, int BT>
__global__ __launch_bounds__(768) void d1_persist_blk_kernel(MgpCgCtrl* __restrict__ ctrl, D1PBuf pb,
                                                             const T* __restrict__ A, long n, int nt, int S,
                                                             T* __restrict__ r, T* __restrict__ v,
                                                             const T* __restrict__ dinv, T* __restrict__ cpart, T thr,
                                                             T min_float, int max_it, int first_poll_sleep, int absent_wg, int spread,
                                                             unsigned long long* __restrict__ trace) {
  constexpr int TS = 64;
  constexpr int W = Gran<T>::W;
  using Cfg = D1Blk<T, BT>;
  constexpr int CB = Cfg::kColBuf;
  constexpr int GB = Gran<T>::kBytes;  // bytes of an element's granules
  const GranRs grs = make_gran_rs(pb.Qg, pb.bytes);
  const int wpg_off = (int)((const char*)pb.wpg - (const char*)pb.Qg), cg_off = (int)((const char*)pb.cg - (const char*)pb.Qg),
            zg_off = (int)((const char*)pb.zg - (const char*)pb.Qg);
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // 0..11
  const int g = w >> 2, wq = w & 3;
  const int me = (int)blockIdx.x;
  // diagnosis (MGP_D1_TRACE=<file>): thread 0 of workgroups 0 (block (0, 0), owner of chunk 0) and 3 (block (0, 3), no
  // chunk) stamps the 100 MHz constant counter at points of the first 64 iterations: trace[row][iteration][point]
  const int trow = me == 0 ? 0 : (me == 3 ? 1 : -1);
  auto stamp = [&](int it, int point) {
    if (trace != nullptr && t == 0 && trow >= 0 && it < 64) trace[((long)trow * 64 + it) * 8 + point] = wall_clock64();
  };
  const int nblk = S * (S + 1) / 2;
  if (me >= nblk) return;  // the rest of the grid leaves at once
  if (me == absent_wg) return;  // fault injection (MGP_D1_INJECT_ABSENT): a workgroup that is not there -- the others must time out, not hang
  int SI = 0, SJ = me;     // row-major over the upper triangle of super-blocks
  while (SJ >= S - SI) {
    SJ -= S - SI;
    ++SI;
  }
  SJ += SI;
  const bool diag = SI == SJ;
  const int I = 3 * SI + g;                       // tile row of this four-wave group
  const long qstride = (long)nt * (S + 1) * TS;   // elements of one column's slot vectors
  __shared__ T pJ[3][BT][TS], pI[3][BT][TS];
  constexpr int NI = BT < 6 ? (BT < 2 ? 2 : 2 * ((BT + 1) / 2)) : 6;  // most (chunk, column) items a workgroup can own (BT > 6: S >= 12, <= 4)
  __shared__ T rOwn[NI][TS], vOwn[NI][TS], shOwn[NI][2];
  __shared__ T dP[12];                  // owner: per polling wave its sum of workgroup shares
  __shared__ int it_c[NI], it_e[NI], it_p[NI];  // an owned item: chunk, column, where its p lives (3 role + pos)
  __shared__ T gsum[BT][12], csum[BT][3];  // per column: the waves' shares of p.Ap from the row products, the layers' from the column products
  __shared__ T sh_s[BT][3];             // per column: rz, ||r||^2 of the current residual, p.Ap
  __shared__ T rzo_s[BT];               // per column: rz of the previous iteration
  __shared__ int fail_s;
  extern __shared__ __attribute__((aligned(16))) unsigned char d1b_dyn_lds[];
  T(*a2s)[TS] = reinterpret_cast<T(*)[TS]>(d1b_dyn_lds) + (long)g * TS;  // rows of this group's third tile
  T(*colp)[3][12][TS] = reinterpret_cast<T(*)[3][12][TS]>(d1b_dyn_lds + Cfg::kTile);  // [buffer][layer][wave][column]
  T(*apP)[TS] = reinterpret_cast<T(*)[TS]>(d1b_dyn_lds + Cfg::kTile);  // owner, [polling wave][element]: its sum of slot vectors -- in colp, which rests between the tile phases
  if (t == 0) fail_s = 0;
  for (int i = t; i < 3 * BT * TS; i += 768) {  // chunks beyond nt are never written again: 0 * p must be 0
    (&pJ[0][0][0])[i] = 0;
    (&pI[0][0][0])[i] = 0;
  }
  // ---- the tiles, ONCE: rows 16 wq .. 16 wq + 15 of tiles (I, 3 SJ + ly), lane = column; absent tiles and ragged edges
  // are zeros (in a diagonal block the tiles below the diagonal, g > ly)
  T a0[16], a1[16];
#pragma unroll
  for (int ly = 0; ly < 3; ++ly) {
    const int J = 3 * SJ + ly;
    const bool have = I < nt && J < nt && (!diag || g <= ly);
    const long r0 = (long)(have ? I : 0) * TS + 16 * wq, c = (long)(have ? J : 0) * TS + l;
    const long cj = c < n ? c : n - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long row = r0 + q < n ? r0 + q : n - 1;
      T x = A[row * n + cj];
      x = (have && r0 + q < n && c < n) ? x : (T)0;
      if (ly == 0) a0[q] = x;
      else if (ly == 1) a1[q] = x;
      else a2s[16 * wq + q][l] = x;
    }
  }
  // ---- ownership, by (chunk, column): E = min(BT, S / 3) columns of a chunk go to E different workgroups that hold its
  // p anyway -- (c = 3 s + pos, e) with d = pos + 3 (e mod E) belongs to block (s, s + d) as the row chunk `pos`, or, where
  // s + d >= S, to block (s + d - S, s) as the column chunk `pos`; columns e, e + E, ... share an owner.  At S = 22 and
  // BT <= 7 every column of a chunk has an owner of its own and a workgroup owns at most two items: the owners' reads
  // (S + 1 vectors and the shares per item) and updates spread over up to 64 BT workgroups instead of 64.
  const T(*pIr)[BT][TS] = diag ? pJ : pI;  // row chunks of a diagonal block ARE its column chunks
  int nitems = 0;
  {
    // MGP_D1_OWNER_SPREAD=0: all columns of a chunk at one owner (an A/B switch; up to six columns -- seven items do not fit)
    const int E = (!spread && BT <= 6) ? 1 : (BT < S / 3 ? BT : S / 3), dl = SJ - SI;
    if (t == 0) {
      int ni = 0;
      if (dl <= 3 * E - 1 && 3 * SI + dl % 3 < nt)
        for (int e = dl / 3; e < BT; e += E) it_c[ni] = 3 * SI + dl % 3, it_e[ni] = e, it_p[ni] = dl % 3, ++ni;
      const int dw = S - dl;
      if (dl > 0 && dw <= 3 * E - 1 && 3 * SJ + dw % 3 < nt)
        for (int e = dw / 3; e < BT; e += E) it_c[ni] = 3 * SJ + dw % 3, it_e[ni] = e, it_p[ni] = 3 + dw % 3, ++ni;
      for (int q = ni; q < NI; ++q) it_c[q] = -1, it_e[q] = 0, it_p[q] = 0;
    }
    __syncthreads();
    for (int q = 0; q < NI; ++q) nitems += it_c[q] >= 0 ? 1 : 0;
  }
  // a wave's part in the owner phase, as scalars for the whole solve: it polls for item p_it (units from p_sw on) and
  // updates item w
  const int wpi = (BT == 1 || nitems == 0) ? 12 : 12 / nitems;  // polling waves per item; nitems <= 6 (one column: <= 1)
  const int p_it = BT == 1 ? 0 : w / wpi, p_sw = w - p_it * wpi;
  const int p_c = __builtin_amdgcn_readfirstlane(p_it < nitems ? it_c[p_it] : 0),
            p_e = __builtin_amdgcn_readfirstlane(p_it < nitems ? it_e[p_it] : 0);
  const int u_c = __builtin_amdgcn_readfirstlane(w < nitems ? it_c[w] : 0),
            u_e = __builtin_amdgcn_readfirstlane(w < nitems ? it_e[w] : 0),
            u_p = __builtin_amdgcn_readfirstlane(w < nitems ? it_p[w] : 0);
  T dinv_own = 1;    // 1 / diag of the item's chunk: one register pair (LDS is full at six columns)
  if (w < nitems) {  // wave i: item i stays on the chip for the whole solve
    const int c = u_c, e = u_e;
    const long oi = (long)c * TS + l;
    const bool ook = oi < n;
    rOwn[w][l] = ook ? r[(long)e * n + oi] : (T)0;
    vOwn[w][l] = ook ? v[(long)e * n + oi] : (T)0;
    if (JAC && ook) dinv_own = dinv[oi];
    if (l == 0) {
      shOwn[w][0] = cpart[(long)e * 2 * CP + c];
      shOwn[w][1] = cpart[(long)e * 2 * CP + CP + c];
    }
  }
  // duties of the B phase, eight per column e (d = 8 e + j) over the twelve waves: j = 0, 1: the chunks' shares of rz /
  // of ||r||^2; j = 2..4: z of column chunk j - 2 (-> p_J); j = 5..7: z of row chunk j - 5 (-> p_I; none in a diagonal block)
  constexpr int ND = 8 * BT, DPW = (ND + 11) / 12;
  if (t < BT) rzo_s[t] = 0;  // 0 makes p_1 = z_0 (the beta-term dropped, :79-84)
  int k = 0;
  __syncthreads();
  while (true) {
    long kz = 0;  // an opaque zero in every offset: loop-invariant addresses are not hoisted out of the solve and spilled
    asm volatile("" : "+s"(kz));
    const int kzi = (int)kz;
    stamp(k, 0);
    // ================================================================= B_k
    const unsigned eb = (unsigned)k + 1u;
    T dv[DPW];
    {
      int vo[DPW], so[DPW];
      bool need[DPW];
#pragma unroll
      for (int s = 0; s < DPW; ++s) {
        const int d = w + 12 * s, e = d >> 3, j = d & 7;
        const bool sduty = d < ND && j < 2;
        const int c = j < 5 ? 3 * SJ + (j >= 2 ? j - 2 : 0) : 3 * SI + j - 5;
        const bool zduty = d < ND && j >= 2 && !(j >= 5 && diag) && c < nt;
        const long el = (long)c * TS + l;
        need[s] = sduty ? l < nt : (zduty && el < n);
        so[s] = sduty ? cg_off + (e * 128 + 64 * j) * GB : (zduty ? zg_off + e * (int)n * GB : zg_off);
        vo[s] = (sduty ? l : (zduty ? (int)(el < n ? el : n - 1) : w * TS + l)) * GB + kzi;  // idle: see poll_units
      }
      const bool ok = poll_units<T, DPW>(grs, vo, so, need, eb, dv);
      if (!ok && l == 0) fail_s = 1;
#pragma unroll
      for (int s = 0; s < DPW; ++s) {
        const int d = w + 12 * s, e = d >> 3, j = d & 7;
        if (d < ND && j < 2) {  // the same sums, in the same order, as the statistics kernel forms from the plain shares
          const T sum = wave_allsum_valu(dv[s]);
          if (l == 0) sh_s[e][j] = sum;
        }
      }
    }
    __syncthreads();
    stamp(k, 1);
    if (fail_s) break;
    bool any = false;
#pragma unroll
    for (int e = 0; e < BT; ++e) any = any || (T)0.5 * sh_s[e][1] > thr;
    if (!(any && k < max_it)) break;  // :59-62
#pragma unroll
    for (int s = 0; s < DPW; ++s) {
      const int d = w + 12 * s, e = d >> 3, j = d & 7;
      if (d < ND && j >= 2) {
        const int c = j < 5 ? 3 * SJ + j - 2 : 3 * SI + j - 5;
        if (!(j >= 5 && diag) && c < nt) {
          T* dst = j < 5 ? pJ[j - 2][e] : pI[j - 5][e];
          const T ro = rzo_s[e], rn = sh_s[e][0];
          const bool drop = ro <= min_float;  // :79, per column
          const T beta = drop ? (T)0 : rn / ro;
          dst[l] = drop ? dv[s] : mgp_fma(beta, dst[l], dv[s]);  // a select, not 0 * p
        }
      }
    }
    lds_barrier();
    if (t < BT) rzo_s[t] = sh_s[t][0];  // read again only after the next B phase's barrier
    stamp(k, 2);
    // ================================================================= tile products, epoch k + 1, column by column
    const unsigned ea = (unsigned)k + 1u;
#pragma nounroll
    for (int e = 0; e < BT; ++e) {
      const int cb = CB == 2 ? (e & 1) : 0;
      // single buffer (7, 8 columns: LDS is full): the sums of column e - 1 must have been read -- waited for HERE, where
      // nothing is live (after the products it spilled 35 registers)
      if (CB == 1 && e > 0) lds_barrier();
      const T pj0 = pJ[0][e][l], pj1 = pJ[1][e][l], pj2 = pJ[2][e][l];
      T x[16];
      T cs0 = 0, cs1 = 0, cs2 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const T a2lo = a2s[16 * wq + q][l], a2hi = a2s[16 * wq + q + 8][l];
        const T pilo = pIr[g][e][16 * wq + q], pihi = pIr[g][e][16 * wq + q + 8];  // LDS broadcast reads
        T lo = a0[q] * pj0, hi = a0[q + 8] * pj0;
        lo = mgp_fma(a1[q], pj1, lo);
        hi = mgp_fma(a1[q + 8], pj1, hi);
        lo = mgp_fma(a2lo, pj2, lo);
        hi = mgp_fma(a2hi, pj2, hi);
        x[q] = rs_swap_add<true>(lo, hi);  // rows q and q + 8 meet across lane bit 32
        cs0 = mgp_fma(a0[q], pilo, cs0);
        cs1 = mgp_fma(a1[q], pilo, cs1);
        cs2 = mgp_fma(a2lo, pilo, cs2);
        cs0 = mgp_fma(a0[q + 8], pihi, cs0);
        cs1 = mgp_fma(a1[q + 8], pihi, cs1);
        cs2 = mgp_fma(a2hi, pihi, cs2);
      }
      // a diagonal tile enters through its row product only
      colp[cb][0][w][l] = (diag && g == 0) ? (T)0 : cs0;
      colp[cb][1][w][l] = (diag && g == 1) ? (T)0 : cs1;
      colp[cb][2][w][l] = (diag && g == 2) ? (T)0 : cs2;
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = rs_swap_add<false>(x[q], x[q + 4]);  // lane bit 16
      rs_step_row<T, 2, 8>(x, l);
      rs_step_row<T, 1, 4>(x, l);
      T sr = x[0];
      sr += lane_xor_row<2>(sr);
      sr += lane_xor_row<1>(sr);  // lane l: (sum_ly A_{I J_ly} p_{J_ly})[16 wq + (l >> 2)]
      const int rr = 16 * wq + (l >> 2);
      if ((l & 3) == 0 && I < nt && (long)I * TS + rr < n)
        Gran<T>::store(grs, pb.Qg + ((long)e * qstride + ((long)I * (S + 1) + SJ) * TS + rr + kz) * W, ea, sr);
      const T u = wave_allsum_valu((l & 3) == 0 ? sr * pIr[g][e][rr] : (T)0);
      if (l == 0) gsum[e][w] = u;
      lds_barrier();
      if (e == 0) stamp(k, 6);
      // after the barrier of column e: three waves (rotating with e) sum a tile column's twelve partial products each and
      // publish it; a fourth publishes the workgroup's share of p.Ap of column e - 1
      const int w0 = (3 * e) % 12;
      if (w >= w0 && w < w0 + 3) {
        const int ly = w - w0, J = 3 * SJ + ly;
        T sc = 0;
#pragma unroll
        for (int g2 = 0; g2 < 3; ++g2)
          sc += (colp[cb][ly][4 * g2][l] + colp[cb][ly][4 * g2 + 1][l]) + (colp[cb][ly][4 * g2 + 2][l] + colp[cb][ly][4 * g2 + 3][l]);
        const long jc = (long)J * TS + l;
        if (J < nt && jc < n)
          Gran<T>::store(grs, pb.Qg + ((long)e * qstride + ((long)J * (S + 1) + (diag ? S : SI)) * TS + l + kz) * W, ea, sc);
        const T cd = wave_allsum_valu(sc * pJ[ly][e][l]);  // p_J . (sum_g A_IJ^T p_I); p_J is 0 where there is no column
        if (l == 0) csum[e][ly] = cd;
      } else if (e > 0 && w == (w0 + 3) % 12 && l == 0) {
        T sh = 0;
#pragma unroll
        for (int q = 0; q < 12; ++q) sh += gsum[e - 1][q];
        sh += (csum[e - 1][0] + csum[e - 1][1]) + csum[e - 1][2];
        Gran<T>::store(grs, pb.wpg + ((long)(e - 1) * 256 + me + kz) * W, ea, sh);
      }
    }
    lds_barrier();
    if (t == 0) {
      T sh = 0;
#pragma unroll
      for (int q = 0; q < 12; ++q) sh += gsum[BT - 1][q];
      sh += (csum[BT - 1][0] + csum[BT - 1][1]) + csum[BT - 1][2];
      Gran<T>::store(grs, pb.wpg + ((long)(BT - 1) * 256 + me + kz) * W, ea, sh);
    }
    stamp(k, 3);
    // ================================================================= the owner's update of iteration k + 1
    if (nitems > 0) {
      // 12 / nitems waves per item read its chunk's S + 1 vectors and the workgroups' shares of p.Ap of its column, all
      // in one round trip.  The first poll waits a little: one issued the moment this workgroup has published is served
      // before the slowest producer's store has landed and costs a second round trip
      {
        const int it = p_it, sw = p_sw;
        const int upw = (S + 5 + wpi - 1) / wpi;  // units of a wave: u = sw upw + i; u <= S: vector u; S < u <= S + 4: 64 shares
        if (it < nitems && sw * upw > S + 4) {  // a wave whose units all lie beyond S + 4 has nothing to read
          apP[w][l] = 0;
          if (l == 0) dP[w] = 0;
        } else if (it < nitems) {
          const int c = p_c, e = p_e;
          const bool ook = (long)c * TS + l < n;
          // (an idle unit -- beyond upw or beyond S + 4 -- re-reads the wave's unit 0, the same lines again, never one
          // fixed address: a thousand waves' idle units on ONE line were a hot spot worth 2 us per iteration)
          auto so_fn = [&](int i, int z) {
            const int ui = sw * upw + i + z, u = (i < upw && ui <= S + 4) ? ui : sw * upw + z, m = (u - S - 1) * 64;
            return u <= S ? (e * (int)qstride + (c * (S + 1) + u) * TS) * GB : wpg_off + (e * 256 + m) * GB;
          };
          auto vo_fn = [&](int, int z) { return (l + z) * GB; };
          auto need_fn = [&](int i, int z) {
            const int u = sw * upw + i + z, m = (u - S - 1) * 64;
            return i < upw && (u <= S ? ook : (u <= S + 4 && m + l < nblk));
          };
          for (int sl0 = 0; sl0 < first_poll_sleep; ++sl0) __builtin_amdgcn_s_sleep(1);
          T ap = 0, d = 0;  // in index order
          bool ok;
          auto sums = [&](auto& val, int U) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
              if (sw * upw + i <= S) ap += val[i];
              else d += val[i];
            }
          };
          // as many loads as the wave has units (27 units at S = 22 over 12 / 6 / 4 / 3 / 2 waves for 1 / 2 / 3 / 4 / 5-6
          // items): two idle loads per wave beside three needed ones cost 1.5 us per iteration at one column
          if (BT == 1 || upw <= 3) {  // (which of the forms exist follows from the most items a workgroup can own at this BT)
            T val[3];
            ok = poll_units_fn<T, 3>(grs, vo_fn, so_fn, need_fn, ea, val);
            sums(val, 3);
          } else if (NI >= 2 && upw <= 5) {
            T val[5];
            ok = poll_units_fn<T, 5>(grs, vo_fn, so_fn, need_fn, ea, val);
            sums(val, 5);
          } else if (NI >= 3 && upw <= 9) {
            T val[9];
            ok = poll_units_fn<T, 9>(grs, vo_fn, so_fn, need_fn, ea, val);
            sums(val, 9);
          } else {
            T val[14];
            ok = poll_units_fn<T, 14>(grs, vo_fn, so_fn, need_fn, ea, val);
            sums(val, 14);
          }
          if (!ok && l == 0) fail_s = 1;
          stamp(k, 4);
          apP[w][l] = ap;
          d = wave_allsum_valu(d);
          if (l == 0) dP[w] = d;
        }
      }
      __syncthreads();
      if (fail_s) break;
      stamp(k, 5);
      if (w < nitems) {  // wave i: item i
        const int c = u_c, e = u_e, ip = u_p;
        const long oi = (long)c * TS + l;
        const bool ook = oi < n;
        T ap = 0, d = 0;  // the polling waves in order, SB LDS reads in flight at a time (a chain of twelve dependent
                          // reads cost 0.5 us; twelve at once spill beside many columns)
        constexpr int SB = BT == 1 ? 12 : 6;
        for (int s0 = 0; s0 < wpi; s0 += SB) {
          T a4[SB], d4[SB];
#pragma unroll
          for (int q = 0; q < SB; ++q) {
            const int src = s0 + q < wpi ? w * wpi + s0 + q : 0;
            a4[q] = apP[src][l];
            d4[q] = dP[src];
          }
#pragma unroll
          for (int q = 0; q < SB; ++q) {
            ap += s0 + q < wpi ? a4[q] : (T)0;
            d += s0 + q < wpi ? d4[q] : (T)0;
          }
        }
        const T gamma = (d <= min_float) ? (T)0 : sh_s[e][0] / d;  // :66-68 (rz of the residual the direction came from)
        const T rc = mgp_fma(-gamma, ap, rOwn[w][l]);              // :76
        const T zn = JAC ? rc * dinv_own : rc;                     // :77
        if (ook) Gran<T>::store(grs, pb.zg + ((long)e * n + oi + kz) * W, ea + 1u, zn);
        const T prz = wave_allsum_valu(ook ? zn * rc : (T)0), prr = wave_allsum_valu(ook ? rc * rc : (T)0);
        if (l == 0) {
          Gran<T>::store(grs, pb.cg + ((long)e * 128 + c + kz) * W, ea + 1u, prz);
          Gran<T>::store(grs, pb.cg + ((long)e * 128 + 64 + c + kz) * W, ea + 1u, prr);
          shOwn[w][0] = prz;
          shOwn[w][1] = prr;
        }
        rOwn[w][l] = rc;
        const T pown = ip >= 3 ? pJ[ip - 3][e][l] : pIr[ip][e][l];
        vOwn[w][l] = mgp_fma(gamma, pown, vOwn[w][l]);  // :69
      }
      stamp(k, 7);
    }
    ++k;
  }
  if (w < nitems) {
    const int c = u_c, e = u_e;
    const long oi = (long)c * TS + l;
    if (oi < n) {
      v[(long)e * n + oi] = vOwn[w][l];
      r[(long)e * n + oi] = rOwn[w][l];
    }
    if (l == 0) {
      cpart[(long)e * 2 * CP + c] = shOwn[w][0];
      cpart[(long)e * 2 * CP + CP + c] = shOwn[w][1];
    }
  }
  if (fail_s && t == 0) *pb.err = 1;
  if (me == 0 && t == 0) ctrl->iters = k;
}

// ------------------------------------------------------------------ n <= 2048: the FULL matrix on the chip
// At nt <= 32 tile rows all nt x nt tiles fit (1024 tiles, four per workgroup of 1024 threads), and a full tile row /
// column set removes the expensive half of the triangle scheme: workgroup (J, rg) holds the tiles (I, J) of its row
// group rg (up to four tile rows I, one per four-wave group) and forms ONLY the lane-local product
//     y_J[c] += sum_r A[I rows r][c] * p_I[r]     (= (A_JI p_I)[c] by symmetry: lane = column, no cross-lane step),
// sums its four tiles in LDS and publishes ONE 64-vector per workgroup and column -- a chunk's (A p)_J is the sum of
// R = G / nt such vectors (8 at nt = 32, against 32 slots in the triangle scheme), and the owner of chunk J polls them
// and the workgroups' shares of p.Ap in the SAME round trip.  With no cross-lane work a further right-hand side costs
// 16 fused multiply-adds per tile and wave, so the form carries BT <= 8 columns (the reference's default of 5 probes,
// models.py:286, at C2's M = 2048): per column its own recurrence (:64-85) and guards (:68, :79), `any` over the
// columns as stopping rule (:59-62).  Hand-offs, epochs, bounds and fail-over as described above.  Granule arrays
// per column e: cg + e 128 W, zg + e n W, wpg + e 256 W, Qg + e (nt R 64) W.
template <typename T, bool JAC, int BT>
__global__ __launch_bounds__(1024) void d1_persist_full_kernel(MgpCgCtrl* __restrict__ ctrl, D1PBuf pb,
                                                              const T* __restrict__ A, long n, int nt, int R, int rpg,
                                                              T* __restrict__ r, T* __restrict__ v,
                                                              const T* __restrict__ dinv, T* __restrict__ cpart,
                                                              T thr, T min_float, int max_it, int first_poll_sleep, int absent_wg, int spread,
                                                              unsigned long long* __restrict__ trace) {
  constexpr int TS = 64;
  constexpr int W = Gran<T>::W;
  const GranRs grs = make_gran_rs(pb.Qg, pb.bytes);
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // 0..15
  const int q = w >> 2, wq = w & 3;                       // four-wave group = tile of the workgroup, wave in the group
  const int me = (int)blockIdx.x;
  auto stamp = [&](int it, int point) {
    if (trace != nullptr && t == 0 && blockIdx.x < 2 && it < 64)
      trace[((long)blockIdx.x * 64 + it) * 8 + point] = wall_clock64();
  };
  const int nact = nt * R;  // workgroups with tiles; the rest of the grid leaves at once
  if (me >= nact) return;
  if (me == absent_wg) return;  // fault injection, see d1_persist_blk_kernel
  const int J = me / R, rg = me - J * R;
  const int I = rg * rpg + q;               // tile row of this group
  const bool have = q < rpg && I < nt;      // uniform per group
  // chunk J, column e belongs to workgroup (J, e mod R): all R row groups of a tile column hold p_J, so the columns'
  // owners -- their reads of R vectors + the shares, their updates -- are R different workgroups (MGP_D1_OWNER_SPREAD=0:
  // row group 0 owns every column).  Local item i of this workgroup is column e = e0 + i estep
  const int e0 = spread ? rg : 0, estep = spread ? R : 1;
  const int nown = (spread || rg == 0) && e0 < BT ? (BT - e0 + estep - 1) / estep : 0;
  const long qstride = (long)nt * R * TS;   // elements of one column's vectors
  __shared__ T pJ[BT][TS], pI[4][BT][TS];
  extern __shared__ __attribute__((aligned(16))) unsigned char d1f_dyn_lds[];  // colp[BT][4][4][TS]: 8 KB per column in fp64
  T(*colp)[4][4][TS] = reinterpret_cast<T(*)[4][4][TS]>(d1f_dyn_lds);
  __shared__ T rOwn[BT][TS], vOwn[BT][TS], dOwn[TS], shOwn[BT][2], apOwn[BT][TS];
  __shared__ T sh_s[BT][3];   // per column: rz, ||r||^2 of the current residual, p.Ap
  __shared__ T rzo_s[BT];     // per column: rz of the previous iteration (LDS, not BT registers in every thread)
  __shared__ int fail_s;
  if (t == 0) fail_s = 0;
  // ---- the group's tile, once: rows 16 wq .. 16 wq + 15 of tile (I, J), lane = column; ragged edges are zeros
  T a[16];
  {
    const long r0 = (long)(have ? I : 0) * TS + 16 * wq, c = (long)J * TS + l;
    const long cj = c < n ? c : n - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long row = r0 + e < n ? r0 + e : n - 1;
      const T x = A[row * n + cj];
      a[e] = (have && r0 + e < n && c < n) ? x : (T)0;
    }
  }
  const long oi = (long)J * TS + l;  // element of chunk J
  const bool ook = oi < n;
  if (w < nown) {  // wave i: item i of the owner's chunk
    const int e = e0 + w * estep;
    rOwn[w][l] = ook ? r[(long)e * n + oi] : (T)0;
    vOwn[w][l] = ook ? v[(long)e * n + oi] : (T)0;
    if (w == 0) dOwn[l] = (JAC && ook) ? dinv[oi] : (T)1;
    if (l == 0) {
      shOwn[w][0] = cpart[(long)e * 2 * CP + J];
      shOwn[w][1] = cpart[(long)e * 2 * CP + CP + J];
    }
  }
  // duties of the B phase, spread over the sixteen waves: duty d < BT: the chunks' shares of rz, column d; < 2 BT: of
  // ||r||^2; < 3 BT: z of chunk J (-> p_J); then z of tile row I_g, column e (-> p_I[g]), d = 3 BT + 4 e + g
  constexpr int ND = 7 * BT, DPW = (ND + 15) / 16;  // duties per wave
  constexpr int GB = Gran<T>::kBytes;
  const int wpg_off = (int)((const char*)pb.wpg - (const char*)pb.Qg), cg_off = (int)((const char*)pb.cg - (const char*)pb.Qg),
            zg_off = (int)((const char*)pb.zg - (const char*)pb.Qg);
  if (t < BT) rzo_s[t] = 0;  // 0 makes p_1 = z_0 (the beta-term dropped, :79-84)
  int k = 0;
  __syncthreads();
  while (true) {
    long kz = 0;  // an opaque zero in every offset: loop-invariant addresses are not hoisted out of the solve and spilled
    asm volatile("" : "+s"(kz));
    const int kzi = (int)kz;
    stamp(k, 0);
    // ================================================================= B_k
    const unsigned eb = (unsigned)k + 1u;
    T dv[DPW];
    {
      int vo[DPW], so[DPW];
      bool need[DPW];
#pragma unroll
      for (int s = 0; s < DPW; ++s) {
        const int d = w + 16 * s;
        const bool sduty = d < 2 * BT, jduty = !sduty && d < 3 * BT, iduty = d >= 3 * BT && d < ND;
        const int e = sduty ? (d < BT ? d : d - BT) : (jduty ? d - 2 * BT : (d - 3 * BT) >> 2);
        const int g2 = (d - 3 * BT) & 3, I2 = rg * rpg + g2;
        const bool irow = iduty && g2 < rpg && I2 < nt;
        const long el = jduty ? oi : (long)(irow ? I2 : 0) * TS + l;
        need[s] = sduty ? l < nt : ((jduty || irow) && el < n);
        so[s] = sduty ? cg_off + (e * 128 + (d < BT ? 0 : 64)) * GB : ((jduty || irow) ? zg_off + e * (int)n * GB : zg_off);
        vo[s] = (sduty ? l : ((jduty || irow) ? (int)(el < n ? el : n - 1) : w * TS + l)) * GB + kzi;  // idle: see poll_units
      }
      const bool ok = poll_units<T, DPW>(grs, vo, so, need, eb, dv);
      if (!ok && l == 0) fail_s = 1;
#pragma unroll
      for (int s = 0; s < DPW; ++s) {
        const int d = w + 16 * s;
        if (d < 2 * BT) {  // the same sums, in the same order, as the statistics kernel forms from the plain shares
          const T sum = wave_allsum_valu(dv[s]);
          if (l == 0) sh_s[d < BT ? d : d - BT][d < BT ? 0 : 1] = sum;
        }
      }
    }
    __syncthreads();
    stamp(k, 1);
    if (fail_s) break;
    bool any = false;
#pragma unroll
    for (int e = 0; e < BT; ++e) any = any || (T)0.5 * sh_s[e][1] > thr;
    if (!(any && k < max_it)) break;  // :59-62
#pragma unroll
    for (int s = 0; s < DPW; ++s) {
      const int d = w + 16 * s;
      if (d >= 2 * BT && d < ND) {
        const int e = d < 3 * BT ? d - 2 * BT : (d - 3 * BT) >> 2;
        T* dst = d < 3 * BT ? pJ[e] : pI[(d - 3 * BT) & 3][e];
        const T ro = rzo_s[e], rn = sh_s[e][0];
        const bool drop = ro <= min_float;  // :79, per column
        const T beta = drop ? (T)0 : rn / ro;
        dst[l] = drop ? dv[s] : mgp_fma(beta, dst[l], dv[s]);  // a select, not 0 * p
      }
    }
    lds_barrier();
    if (t < BT) rzo_s[t] = sh_s[t][0];  // read again only after the next B phase's barrier
    stamp(k, 2);
    // ================================================================= tile products, epoch k + 1: lane-local only
    const unsigned ea = (unsigned)k + 1u;
#pragma unroll
    for (int e = 0; e < BT; ++e) {
      T cs = 0;
      if (have) {
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) cs = mgp_fma(a[i2], pI[q][e][16 * wq + i2], cs);  // LDS broadcast reads
      }
      colp[e][q][wq][l] = cs;
    }
    lds_barrier();
    if (w < BT) {  // wave e: column e -- the workgroup's 64-vector: groups in order, waves in order
      T y = 0;
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2)
        y += (colp[w][g2][0][l] + colp[w][g2][1][l]) + (colp[w][g2][2][l] + colp[w][g2][3][l]);
      if (ook) Gran<T>::store(grs, pb.Qg + ((long)w * qstride + ((long)J * R + rg) * TS + l + kz) * W, ea, y);
      const T share = wave_allsum_valu(ook ? pJ[w][l] * y : (T)0);  // p_J . (partial of (A p)_J)
      if (l == 0) Gran<T>::store(grs, pb.wpg + ((long)w * 256 + me + kz) * W, ea, share);
    }
    stamp(k, 3);
    // ================================================================= the owner's update of iteration k + 1
    if (nown > 0) {
      // wave 2 i: the R <= 8 vectors of chunk J, column e_i; wave 2 i + 1: the workgroups' shares of p.Ap, column e_i --
      // all in the same round trip.  The first poll waits a little: one issued the moment this workgroup has published its
      // own vector is served before the slowest producer's store has landed and costs a second round trip
      if (w < 2 * nown) {
        const int it = w >> 1, e = e0 + it * estep;
        const bool slots = (w & 1) == 0;
        for (int sl0 = 0; sl0 < first_poll_sleep; ++sl0) __builtin_amdgcn_s_sleep(1);
        T acc = 0;  // row groups / blocks of 64 workgroups in order
        bool ok;
        if (slots) {  // as many loads as units (idle loads are traffic): R <= 8 vectors ...
          T val[8];
          int vo[8], so[8];
          bool need[8];
#pragma unroll
          for (int i2 = 0; i2 < 8; ++i2) {
            need[i2] = i2 < R && ook;
            so[i2] = (e * (int)qstride + (J * R + (i2 < R ? i2 : 0)) * TS) * GB;  // (an idle unit re-reads vector 0)
            vo[i2] = l * GB + kzi;
          }
          ok = poll_units<T, 8>(grs, vo, so, need, ea, val);
#pragma unroll
          for (int i2 = 0; i2 < 8; ++i2) acc += val[i2];
        } else {  // ... or four blocks of 64 workgroups' shares
          T val[4];
          int vo[4], so[4];
          bool need[4];
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) {
            need[i2] = i2 * 64 + l < nact;
            so[i2] = wpg_off + (e * 256 + i2 * 64) * GB;
            vo[i2] = l * GB + kzi;
          }
          ok = poll_units<T, 4>(grs, vo, so, need, ea, val);
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) acc += val[i2];
        }
        if (!ok && l == 0) fail_s = 1;
        if (slots) {
          apOwn[it][l] = acc;
        } else {
          acc = wave_allsum_valu(acc);
          if (l == 0) sh_s[e][2] = acc;
        }
      }
      __syncthreads();
      if (fail_s) break;
      stamp(k, 5);
      if (w < nown) {  // wave i: item i
        const int e = e0 + w * estep;
        const T d = sh_s[e][2];
        const T gamma = (d <= min_float) ? (T)0 : sh_s[e][0] / d;  // :66-68 (rz of the residual the direction came from)
        const T rc = mgp_fma(-gamma, apOwn[w][l], rOwn[w][l]);  // :76
        const T zn = JAC ? rc * dOwn[l] : rc;                   // :77
        if (ook) Gran<T>::store(grs, pb.zg + ((long)e * n + oi + kz) * W, ea + 1u, zn);
        const T prz = wave_allsum_valu(ook ? zn * rc : (T)0), prr = wave_allsum_valu(ook ? rc * rc : (T)0);
        if (l == 0) {
          Gran<T>::store(grs, pb.cg + ((long)e * 128 + J + kz) * W, ea + 1u, prz);
          Gran<T>::store(grs, pb.cg + ((long)e * 128 + 64 + J + kz) * W, ea + 1u, prr);
          shOwn[w][0] = prz;
          shOwn[w][1] = prr;
        }
        rOwn[w][l] = rc;
        vOwn[w][l] = mgp_fma(gamma, pJ[e][l], vOwn[w][l]);  // :69
      }
      stamp(k, 7);
    }
    ++k;
  }
  if (w < nown) {
    const int e = e0 + w * estep;
    if (ook) {
      v[(long)e * n + oi] = vOwn[w][l];
      r[(long)e * n + oi] = rOwn[w][l];
    }
    if (l == 0) {
      cpart[(long)e * 2 * CP + J] = shOwn[w][0];
      cpart[(long)e * 2 * CP + CP + J] = shOwn[w][1];
    }
  }
  if (fail_s && t == 0) *pb.err = 1;
  if (me == 0 && t == 0) ctrl->iters = k;
}

template <typename T>
int d1_layout(MgpDense1* st, void* arena, long n) {
  // arena (bt = columns): tpart[bt][ntiles] | cpart[bt][2 CP] | scal[bt][2] | pb[2][bt][n] | zpub[n] |
  //        (bt > 1) the slots Q[bt][nt][n] | hand-off words (a 128-byte line of their own: error word, two stop words)
  //        | granules of the register-resident form: Qg[nt n] | wpg[256] | cg[128] | zg[n]  (x W 8-byte words each)
  const long nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2, bt = st->bt;
  T* a = (T*)arena;
  st->tpart = a;
  a += bt * ntiles;
  st->cpart = a;
  a += bt * 2 * CP;
  st->scal = a;
  a += bt * 2;
  st->pb[0] = a;
  st->pb[1] = a + bt * n;
  a += 2 * bt * n;
  st->zpub = a;
  a += bt * n;
  st->Qm = nullptr;
  if (bt > 1) {
    st->Qm = a;
    a += bt * nt * n;
  }
  st->sync = (void*)(((uintptr_t)a + 127) & ~(uintptr_t)127);
  st->gran = (char*)st->sync + 128;
  return MGP_OK;
}

// granule elements: Qg (round-robin triangle form: nt n; full form: bt nt 8 64; super-block form: bt nt (S + 1) 64 with
// S + 1 <= 23) | wpg bt 256 | cg bt 128 | zg bt n
static long d1_qg_elems(long n, long bt) {
  const long nt = (n + 63) / 64;
  const long blk = bt * nt * 23 * 64;
  return nt * n > blk ? nt * n : blk;
}
static size_t d1_gran_bytes(int dtype, long n, long bt) {
  return (size_t)(d1_qg_elems(n, bt) + bt * (256 + 128 + n)) * (dtype == MGP_F64 ? 2 : 1) * sizeof(gu64);
}

}  // namespace

size_t mgp_dense1_bytes(const mgp_handle* h, int dtype, int64_t n, int64_t bt) {
  const long nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
  size_t e = (size_t)(bt * (ntiles + 2 * CP + 2 + 3 * n));
  if (bt > 1) e += (size_t)bt * nt * n;
  return e * mgp_elem(dtype) + 128 + 128 + (mgp_dense1_persist_eligible(h, n, bt) ? d1_gran_bytes(dtype, n, bt) : 0) + 64;
}

bool mgp_dense1_eligible(const mgp_handle* h, int64_t n) {
  return h->cg_dense1 != 0 && n >= h->tri_min_n && n <= 64L * CP;
}

// the register-resident form: every tile on the chip at once -- at most nine per workgroup, one workgroup per CU
static int d1_persist_grid(const mgp_handle* h) { return h->num_cus < 256 ? h->num_cus : 256; }
// full-matrix form (d1_persist_full_kernel): nt <= 32 tile rows, R = min(8, G / nt) row groups of <= 4 tile rows
static bool d1_full_geometry(const mgp_handle* h, long nt, int* R, int* rpg) {
  const int G = d1_persist_grid(h);
  int r = nt > 0 ? (int)(G / nt) : 0;
  if (r > 8) r = 8;
  if (r < 1 || nt > 32) return false;
  const int g = (int)((nt + r - 1) / r);
  if (g > 4) return false;
  *R = r;
  *rpg = g;
  return true;
}
// super-block form (d1_persist_blk_kernel): S = ceil(nt / 3) super-rows, one workgroup per block of the upper triangle;
// S >= 6 so that every chunk finds an owner of its own among the blocks that hold it
static bool d1_blk_geometry(const mgp_handle* h, long nt, int* S) {
  const long s = (nt + 2) / 3;
  if (nt > 64 || s < 6 || s * (s + 1) / 2 > d1_persist_grid(h)) return false;
  *S = (int)s;
  return true;
}
// which register-resident form a solve takes: 1 = full matrix, 2 = super-blocks of the triangle, 0 = none
// (MGP_CG_DENSE1: 3 = full matrix where it fits, 4 = super-blocks wherever they fit)
static int d1_persist_form(const mgp_handle* h, long n, long bt) {
  const long nt = (n + 63) / 64;
  if (h->cg_dense1 < 3 || !mgp_dense1_eligible(h, n) || bt < 1 || bt > 8) return 0;
  int R = 0, rpg = 0, S = 0;
  if (h->cg_dense1 == 3 && d1_full_geometry(h, nt, &R, &rpg)) return 1;
  if (d1_blk_geometry(h, nt, &S) && (bt <= 6 || S >= 12)) return 2;  // 7, 8 columns: at most four owned items per workgroup from S = 12 on
  if (d1_full_geometry(h, nt, &R, &rpg)) return 1;
  return 0;
}
bool mgp_dense1_persist_eligible(const mgp_handle* h, int64_t n, int64_t bt) { return d1_persist_form(h, n, bt) != 0; }

int mgp_dense1_begin(mgp_handle* h, MgpDense1* st, int dtype, const void* A, int64_t n, const void* B, const void* av,
                     void* V, void* r, const void* dinv, MgpCgCtrl* ctrl, void* arena, double thr, double min_float,
                     int64_t max_it, int persist, int bt) {
  st->dtype = dtype;
  st->bt = bt;
  st->A = A;
  st->n = n;
  st->V = V;
  st->r = r;
  st->dinv = dinv;
  st->ctrl = ctrl;
  st->thr = thr;
  st->min_float = min_float;
  st->max_it = (int)(max_it > 2147483647L ? 2147483647L : max_it);
  st->nt = (int)((n + 63) / 64);
  st->ntiles = (long)st->nt * (st->nt + 1) / 2;
  st->persist = persist;
  if (dtype == MGP_F64) d1_layout<double>(st, arena, n);
  else d1_layout<float>(st, arena, n);
  MGP_HIP(h, hipMemsetAsync(st->cpart, 0, (size_t)bt * 2 * CP * mgp_elem(dtype), h->stream));  // shares of chunks beyond nt stay 0
  // the hand-off error word and -- a tag left by an earlier solve must never match an epoch of this one -- every granule
  MGP_HIP(h, hipMemsetAsync(st->sync, 0, 128 + (persist ? d1_gran_bytes(dtype, n, bt) : 0), h->stream));
  // the product's slots and the tile table (dense.hip owns both)
  MGP_TRY(mgp_symm_gemv_tri_prepare(h, dtype, n, &st->Q, &st->tab));
  if (bt > 1) {
    const dim3 g((unsigned)st->nt, (unsigned)bt);
    if (dtype == MGP_F64)
      hipLaunchKernelGGL((d1m_init_kernel<double>), g, dim3(64), 0, h->stream, ctrl, (const double*)B, (const double*)av,
                         (double*)r, (const double*)dinv, (double*)st->cpart, (double*)st->scal, (double*)st->zpub,
                         (long)n, (int*)st->sync + 1);
    else
      hipLaunchKernelGGL((d1m_init_kernel<float>), g, dim3(64), 0, h->stream, ctrl, (const float*)B, (const float*)av,
                         (float*)r, (const float*)dinv, (float*)st->cpart, (float*)st->scal, (float*)st->zpub, (long)n,
                         (int*)st->sync + 1);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((d1_init_kernel<double>), dim3((unsigned)st->nt), dim3(64), 0, h->stream, ctrl, (const double*)B,
                       (const double*)av, (double*)r, (const double*)dinv, (double*)st->cpart, (double*)st->scal,
                       (double*)st->zpub, (long)n);
  else
    hipLaunchKernelGGL((d1_init_kernel<float>), dim3((unsigned)st->nt), dim3(64), 0, h->stream, ctrl, (const float*)B,
                       (const float*)av, (float*)r, (const float*)dinv, (float*)st->cpart, (float*)st->scal,
                       (float*)st->zpub, (long)n);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T>
static D1PBuf d1_pbuf(const MgpDense1* st) {
  constexpr long W = Gran<T>::W;
  D1PBuf pb;
  const long bt = st->bt;
  pb.Qg = (gu64*)st->gran;
  pb.wpg = pb.Qg + d1_qg_elems((long)st->n, bt) * W;
  pb.cg = pb.wpg + bt * 256 * W;
  pb.zg = pb.cg + bt * 128 * W;
  pb.err = (int*)st->sync;
  pb.bytes = (long)d1_gran_bytes(st->dtype, (long)st->n, bt);
  return pb;
}

// the whole solve in one launch (after begin + finish have left the statistics of r_0 and the first gate)
int mgp_dense1_persist_run(mgp_handle* h, const MgpDense1* st) {
  const dim3 grid((unsigned)d1_persist_grid(h));
  const dim3 sgrid((unsigned)st->nt, (unsigned)st->bt);
  if (st->dtype == MGP_F64)
    hipLaunchKernelGGL((d1_persist_seed_kernel<double>), sgrid, dim3(64), 0, h->stream, (const double*)st->zpub,
                       (const double*)st->cpart, d1_pbuf<double>(st), (long)st->n, st->nt);
  else
    hipLaunchKernelGGL((d1_persist_seed_kernel<float>), sgrid, dim3(64), 0, h->stream, (const float*)st->zpub,
                       (const float*)st->cpart, d1_pbuf<float>(st), (long)st->n, st->nt);
  MGP_LAUNCH_CHECK(h);
  unsigned long long* trace = nullptr;
  const char* trace_path = getenv("MGP_D1_TRACE");
  constexpr size_t kTraceWords = 2 * 64 * 8;
  if (trace_path && *trace_path) {
    MGP_HIP(h, hipMalloc((void**)&trace, kTraceWords * sizeof(unsigned long long)));
    MGP_HIP(h, hipMemsetAsync(trace, 0, kTraceWords * sizeof(unsigned long long), h->stream));
  }
  // n <= 2048 (nt <= 32): the full matrix on the chip, four tiles per workgroup (d1_persist_full_kernel, 1..8 columns)
  int Rg = 0, rpg = 0, Sb = 0;
  const int form = d1_persist_form(h, (long)st->n, st->bt);
  const bool full = form == 1 && d1_full_geometry(h, st->nt, &Rg, &rpg);
  const bool blk = form == 2 && d1_blk_geometry(h, st->nt, &Sb);
  if (form == 0 || (st->bt > 1 && !full && !blk))
    return mgp_fail(h, MGP_E_BADARG, "dense CG: no register-resident form for n = %ld with %d columns", (long)st->n, st->bt);
#define MGP_D1B(TT, JV, BTV)                                                                                         \
  do {                                                                                                               \
    const size_t dyn = D1Blk<TT, BTV>::kDyn;                                                                         \
    MGP_HIP(h, hipFuncSetAttribute((const void*)d1_persist_blk_kernel<TT, JV, BTV>,                                  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                           \
    hipLaunchKernelGGL((d1_persist_blk_kernel<TT, JV, BTV>), grid, dim3(768), dyn, h->stream, st->ctrl,                \
                       d1_pbuf<TT>(st), (const TT*)st->A, (long)st->n, st->nt, Sb, (TT*)st->r, (TT*)st->V,            \
                       (const TT*)st->dinv, (TT*)st->cpart, (TT)st->thr, (TT)st->min_float, st->max_it,               \
                       h->d1_first_poll_sleep, h->d1_inject_absent, h->d1_owner_spread, trace);                                                               \
  } while (0)
#define MGP_D1BB(TT, JV)                     \
  switch (st->bt) {                          \
    case 1: MGP_D1B(TT, JV, 1); break;       \
    case 2: MGP_D1B(TT, JV, 2); break;       \
    case 3: MGP_D1B(TT, JV, 3); break;       \
    case 4: MGP_D1B(TT, JV, 4); break;       \
    case 5: MGP_D1B(TT, JV, 5); break;       \
    case 6: MGP_D1B(TT, JV, 6); break;       \
    case 7: MGP_D1B(TT, JV, 7); break;       \
    default: MGP_D1B(TT, JV, 8); break;      \
  }
#define MGP_D1F(TT, JV, BTV)                                                                                         \
  do {                                                                                                               \
    const size_t dyn = (size_t)BTV * 16 * 64 * sizeof(TT);                                                           \
    MGP_HIP(h, hipFuncSetAttribute((const void*)d1_persist_full_kernel<TT, JV, BTV>,                                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                           \
    hipLaunchKernelGGL((d1_persist_full_kernel<TT, JV, BTV>), grid, dim3(1024), dyn, h->stream, st->ctrl,              \
                       d1_pbuf<TT>(st), (const TT*)st->A, (long)st->n, st->nt, Rg, rpg, (TT*)st->r, (TT*)st->V,       \
                       (const TT*)st->dinv, (TT*)st->cpart, (TT)st->thr, (TT)st->min_float, st->max_it,               \
                       h->d1_first_poll_sleep, h->d1_inject_absent, h->d1_owner_spread, trace);                                                               \
  } while (0)
#define MGP_D1FB(TT, JV)                     \
  switch (st->bt) {                          \
    case 1: MGP_D1F(TT, JV, 1); break;       \
    case 2: MGP_D1F(TT, JV, 2); break;       \
    case 3: MGP_D1F(TT, JV, 3); break;       \
    case 4: MGP_D1F(TT, JV, 4); break;       \
    case 5: MGP_D1F(TT, JV, 5); break;       \
    case 6: MGP_D1F(TT, JV, 6); break;       \
    case 7: MGP_D1F(TT, JV, 7); break;       \
    default: MGP_D1F(TT, JV, 8); break;      \
  }
#define MGP_D1P(TT, JV)                                                                                             \
  do {                                                                                                              \
    if (full) {                                                                                                     \
      MGP_D1FB(TT, JV);                                                                                             \
      break;                                                                                                        \
    }                                                                                                               \
    if (blk) {                                                                                                      \
      MGP_D1BB(TT, JV);                                                                                             \
      break;                                                                                                        \
    }                                                                                                               \
  } while (0)
  if (st->dtype == MGP_F64) {
    if (st->dinv) MGP_D1P(double, true);
    else MGP_D1P(double, false);
  } else {
    if (st->dinv) MGP_D1P(float, true);
    else MGP_D1P(float, false);
  }
#undef MGP_D1F
#undef MGP_D1FB
#undef MGP_D1B
#undef MGP_D1BB
#undef MGP_D1P
  MGP_LAUNCH_CHECK(h);
  if (trace) {  // diagnosis only: drains the stream
    std::vector<unsigned long long> host(kTraceWords);
    MGP_HIP(h, hipMemcpyAsync(host.data(), trace, kTraceWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    MGP_HIP(h, hipStreamSynchronize(h->stream));
    (void)hipFree(trace);
    if (FILE* f = fopen(trace_path, "a")) {
      fprintf(f, "# n=%ld: 100 MHz ticks relative to the workgroup's first stamp; columns: top, B read, p formed, tiles "
                 "done and published, own poll done (owner), A read (owner), first column's tiles done, B published (owner)\n", (long)st->n);
      for (int wg = 0; wg < 2; ++wg)
        for (int it = 0; it < 64; ++it) {
          const unsigned long long* e = &host[((size_t)wg * 64 + it) * 8];
          if (!e[0]) continue;
          fprintf(f, "wg %d it %2d:", wg, it);
          for (int q = 0; q < 8; ++q) fprintf(f, " %7lld", e[q] ? (long long)(e[q] - host[(size_t)wg * 64 * 8]) : -1LL);
          fprintf(f, "\n");
        }
      fclose(f);
    }
  }
  return MGP_OK;
}

template <typename T, bool JAC>
static int d1_step_t(mgp_handle* h, const MgpDense1* st, long k) {
  const int kk = (int)(k & 1);
  const T* p_old = (const T*)st->pb[(k + 1) & 1];
  T* p_new = (T*)st->pb[k & 1];
  hipLaunchKernelGGL((d1_tile_kernel<T, JAC>), dim3((unsigned)st->ntiles), dim3(256), 0, h->stream, st->ctrl,
                     (const T*)st->A, (long)st->n, (const T*)st->r, (const T*)st->dinv, p_old, p_new,
                     (const T*)st->cpart, (T*)st->scal, kk, (const int2*)st->tab, (T*)st->Q, (T*)st->tpart, (T)st->thr,
                     (T)st->min_float, st->max_it, (int*)st->sync + 1);
  MGP_LAUNCH_CHECK(h);
#define MGP_D1U(NTV, PERV, TPMV)                                                                                     \
  hipLaunchKernelGGL((d1_update_kernel<T, JAC, NTV, PERV, TPMV>), dim3((unsigned)st->nt), dim3(NTV), 0, h->stream,     \
                     st->ctrl, (const T*)st->Q, st->nt, (const T*)st->tpart, st->ntiles, (const T*)st->scal, kk,      \
                     (const T*)p_new, (T*)st->V, (T*)st->r, (const T*)st->dinv, (T*)st->cpart, (long)st->n,           \
                     (T)st->min_float, (const int*)st->sync + 1)
  if (st->nt <= 32) MGP_D1U(256, 8, 3);         // 528 tiles
  else if (st->nt <= 64) MGP_D1U(256, 16, 9);   // 2080 tiles
  else MGP_D1U(512, 16, 17);                    // nt <= 128: 8256 tiles
#undef MGP_D1U
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T, bool JAC, int BT>
static int d1m_step_t(mgp_handle* h, const MgpDense1* st, long k) {
  const int kk = (int)(k & 1);
  const T* p_old = (const T*)st->pb[(k + 1) & 1];
  T* p_new = (T*)st->pb[k & 1];
  int* stopw = (int*)st->sync + 1;
  hipLaunchKernelGGL((d1m_tile_kernel<T, JAC, BT>), dim3((unsigned)st->ntiles), dim3(256), 0, h->stream, st->ctrl,
                     (const T*)st->A, (long)st->n, (const T*)st->r, (const T*)st->dinv, p_old, p_new,
                     (const T*)st->cpart, (T*)st->scal, kk, (const int2*)st->tab, (T*)st->Qm, st->nt, (T*)st->tpart,
                     st->ntiles, (T)st->thr, (T)st->min_float, st->max_it, stopw);
  MGP_LAUNCH_CHECK(h);
  const dim3 g((unsigned)st->nt, (unsigned)BT);
#define MGP_D1MU(NTV, PERV, TPMV)                                                                                    \
  hipLaunchKernelGGL((d1m_update_kernel<T, JAC, NTV, PERV, TPMV>), g, dim3(NTV), 0, h->stream, st->ctrl,               \
                     (const T*)st->Qm, st->nt, (const T*)st->tpart, st->ntiles, (const T*)st->scal, kk,               \
                     (const T*)p_new, (T*)st->V, (T*)st->r, (const T*)st->dinv, (T*)st->cpart, (long)st->n,           \
                     (T)st->min_float, (const int*)stopw)
  if (st->nt <= 32) MGP_D1MU(256, 8, 3);
  else if (st->nt <= 64) MGP_D1MU(256, 16, 9);
  else MGP_D1MU(512, 16, 17);
#undef MGP_D1MU
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

template <typename T, bool JAC>
static int d1m_step_bt(mgp_handle* h, const MgpDense1* st, long k) {
  switch (st->bt) {
    case 2: return d1m_step_t<T, JAC, 2>(h, st, k);
    case 3: return d1m_step_t<T, JAC, 3>(h, st, k);
    case 4: return d1m_step_t<T, JAC, 4>(h, st, k);
    case 5: return d1m_step_t<T, JAC, 5>(h, st, k);
    case 6: return d1m_step_t<T, JAC, 6>(h, st, k);
    case 7: return d1m_step_t<T, JAC, 7>(h, st, k);
    default: return d1m_step_t<T, JAC, 8>(h, st, k);
  }
}

// enqueue iteration k (k = 1, 2, ...): T_k, U_k
int mgp_dense1_step(mgp_handle* h, const MgpDense1* st, int64_t k) {
  if (st->bt > 1) {
    if (st->dtype == MGP_F64)
      return st->dinv ? d1m_step_bt<double, true>(h, st, k) : d1m_step_bt<double, false>(h, st, k);
    return st->dinv ? d1m_step_bt<float, true>(h, st, k) : d1m_step_bt<float, false>(h, st, k);
  }
  if (st->dtype == MGP_F64)
    return st->dinv ? d1_step_t<double, true>(h, st, k) : d1_step_t<double, false>(h, st, k);
  return st->dinv ? d1_step_t<float, true>(h, st, k) : d1_step_t<float, false>(h, st, k);
}

// statistics (rz, err, over) and the gate word for the host poll
int mgp_dense1_finish(mgp_handle* h, MgpDense1* st, void* rz, void* err, int* over) {
  if (st->bt > 1) {
    if (st->dtype == MGP_F64)
      hipLaunchKernelGGL((d1m_finish_kernel<double>), dim3(1), dim3(512), 0, h->stream, st->ctrl, (const int*)st->sync,
                         (const double*)st->cpart, (double*)rz, (double*)err, over, (double)st->thr, st->max_it, st->bt);
    else
      hipLaunchKernelGGL((d1m_finish_kernel<float>), dim3(1), dim3(512), 0, h->stream, st->ctrl, (const int*)st->sync,
                         (const float*)st->cpart, (float*)rz, (float*)err, over, (float)st->thr, st->max_it, st->bt);
    MGP_LAUNCH_CHECK(h);
    return MGP_OK;
  }
  if (st->dtype == MGP_F64)
    hipLaunchKernelGGL((d1_finish_kernel<double>), dim3(1), dim3(64), 0, h->stream, st->ctrl,
                       (const int*)st->sync, (const double*)st->cpart, (double*)rz, (double*)err,
                       over, (double)st->thr, st->max_it);
  else
    hipLaunchKernelGGL((d1_finish_kernel<float>), dim3(1), dim3(64), 0, h->stream, st->ctrl,
                       (const int*)st->sync, (const float*)st->cpart, (float*)rz, (float*)err,
                       over, (float)st->thr, st->max_it);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

