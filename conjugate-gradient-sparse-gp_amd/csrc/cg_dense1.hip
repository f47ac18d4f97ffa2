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
                                                     T* __restrict__ scal, long n) {
  const int l = threadIdx.x;
  const long i = (long)blockIdx.x * 64 + l;
  T rv = 0, zv = 0;
  if (i < n) {
    rv = av ? b[i] - av[i] : b[i];
    r[i] = rv;
    zv = dinv ? rv * dinv[i] : rv;
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
    }
  }
}

// statistics + gate for the host poll (once per enqueued batch)
template <typename T>
__global__ __launch_bounds__(64) void d1_finish_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ cpart,
                                                       T* __restrict__ rz, T* __restrict__ err,
                                                       int* __restrict__ over, T thr, int max_it) {
  T s_rz, s_rr;
  sum_shares(cpart, (int)threadIdx.x, s_rz, s_rr);
  if (threadIdx.x == 0) {
    rz[0] = s_rz;
    err[0] = (T)0.5 * s_rz;
    const int any = ((T)0.5 * s_rr > thr) ? 1 : 0;
    over[0] = any;
    if (ctrl->active) ctrl->active = (any && ctrl->iters < max_it) ? 1 : 0;
  }
}

template <typename T, bool JAC>
__global__ __launch_bounds__(256) void d1_tile_kernel(MgpCgCtrl* __restrict__ ctrl, const T* __restrict__ A, long n,
                                                      const T* __restrict__ r, const T* __restrict__ dinv,
                                                      const T* __restrict__ p_old, T* __restrict__ p_new,
                                                      const T* __restrict__ cpart, T* __restrict__ scal, int k,
                                                      const int2* __restrict__ tab, T* __restrict__ Q,
                                                      T* __restrict__ tpart, T thr, T min_float, int max_it) {
  if (ctrl->active == 0) return;
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
    else ctrl->active = 0;
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
                                                       T min_float) {
  if (ctrl->active == 0) return;
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

template <typename T>
int d1_layout(MgpDense1* st, void* arena, long n) {
  // arena: tpart[ntiles] | cpart[2 CP] | scal[2] | pb[2][n]
  const long nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
  T* a = (T*)arena;
  st->tpart = a;
  st->cpart = a + ntiles;
  st->scal = a + ntiles + 2 * CP;
  st->pb[0] = a + ntiles + 2 * CP + 2;
  st->pb[1] = a + ntiles + 2 * CP + 2 + n;
  return MGP_OK;
}

}  // namespace

size_t mgp_dense1_bytes(int dtype, int64_t n) {
  const long nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
  return (size_t)(ntiles + 2 * CP + 2 + 2 * n) * mgp_elem(dtype) + 64;
}

bool mgp_dense1_eligible(const mgp_handle* h, int64_t n) {
  return h->cg_dense1 != 0 && n >= h->tri_min_n && n <= 64L * CP;
}

int mgp_dense1_begin(mgp_handle* h, MgpDense1* st, int dtype, const void* A, int64_t n, const void* B, const void* av,
                     void* V, void* r, const void* dinv, MgpCgCtrl* ctrl, void* arena, double thr, double min_float,
                     int64_t max_it) {
  st->dtype = dtype;
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
  if (dtype == MGP_F64) d1_layout<double>(st, arena, n);
  else d1_layout<float>(st, arena, n);
  MGP_HIP(h, hipMemsetAsync(st->cpart, 0, 2 * CP * mgp_elem(dtype), h->stream));  // shares of chunks beyond nt stay 0
  // the product's slots and the tile table (dense.hip owns both)
  MGP_TRY(mgp_symm_gemv_tri_prepare(h, dtype, n, &st->Q, &st->tab));
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((d1_init_kernel<double>), dim3((unsigned)st->nt), dim3(64), 0, h->stream, ctrl, (const double*)B,
                       (const double*)av, (double*)r, (const double*)dinv, (double*)st->cpart, (double*)st->scal, (long)n);
  else
    hipLaunchKernelGGL((d1_init_kernel<float>), dim3((unsigned)st->nt), dim3(64), 0, h->stream, ctrl, (const float*)B,
                       (const float*)av, (float*)r, (const float*)dinv, (float*)st->cpart, (float*)st->scal, (long)n);
  MGP_LAUNCH_CHECK(h);
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
                     (T)st->min_float, st->max_it);
  MGP_LAUNCH_CHECK(h);
#define MGP_D1U(NTV, PERV, TPMV)                                                                                     \
  hipLaunchKernelGGL((d1_update_kernel<T, JAC, NTV, PERV, TPMV>), dim3((unsigned)st->nt), dim3(NTV), 0, h->stream,     \
                     st->ctrl, (const T*)st->Q, st->nt, (const T*)st->tpart, st->ntiles, (const T*)st->scal, kk,      \
                     (const T*)p_new, (T*)st->V, (T*)st->r, (const T*)st->dinv, (T*)st->cpart, (long)st->n,           \
                     (T)st->min_float)
  if (st->nt <= 32) MGP_D1U(256, 8, 3);         // 528 tiles
  else if (st->nt <= 64) MGP_D1U(256, 16, 9);   // 2080 tiles
  else MGP_D1U(512, 16, 17);                    // nt <= 128: 8256 tiles
#undef MGP_D1U
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

// enqueue iteration k (k = 1, 2, ...): T_k, U_k
int mgp_dense1_step(mgp_handle* h, const MgpDense1* st, int64_t k) {
  if (st->dtype == MGP_F64)
    return st->dinv ? d1_step_t<double, true>(h, st, k) : d1_step_t<double, false>(h, st, k);
  return st->dinv ? d1_step_t<float, true>(h, st, k) : d1_step_t<float, false>(h, st, k);
}

// statistics (rz, err, over) and the gate word for the host poll
int mgp_dense1_finish(mgp_handle* h, const MgpDense1* st, void* rz, void* err, int* over) {
  if (st->dtype == MGP_F64)
    hipLaunchKernelGGL((d1_finish_kernel<double>), dim3(1), dim3(64), 0, h->stream, st->ctrl, (const double*)st->cpart,
                       (double*)rz, (double*)err, over, (double)st->thr, st->max_it);
  else
    hipLaunchKernelGGL((d1_finish_kernel<float>), dim3(1), dim3(64), 0, h->stream, st->ctrl, (const float*)st->cpart,
                       (float*)rz, (float*)err, over, (float)st->thr, st->max_it);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}
