// cg.hip -- device-resident batched preconditioned conjugate gradient (rows CG1, CG3-CG5).
//
// Restates the recurrence of cggp/conjugate_gradient.py:59-98 with every scalar
// (gamma, beta, rz, the stopping test) kept on the device:
//
//   while any_b(0.5 ||r_b||^2 > thr) and i < max_it:                  (:59-62)
//       Ap = p @ A                                                     (:65)   operator kernel(s)
//       gamma = rz / (p . Ap), 0 where p.Ap <= min_float               (:66-68)  \
//       v += gamma p ; r -= gamma Ap   (or r = b - v @ A on a refresh) (:69-76)   | cg_update_kernel
//       z, rz' = M^-1 r ; p = z + p rz'/rz (0 where rz <= min_float)   (:77-84)  /  one block per RHS
//
// The host never sees a scalar: a device word `active` gates every kernel of an iteration, so
// `check_every` iterations can be enqueued back to back and the ones after convergence are
// no-ops -- the step count stays exactly the reference's.  The host polls `active` once per
// batch.  Vectors are [Bt, n] row-major (the function-level layout, :24-32); one workgroup owns
// one right-hand side, dot products are wavefront shuffles + one LDS hop.
#include <chrono>
#include <cstdio>

#include "mgp_common.h"

namespace {

using CgCtrl = MgpCgCtrl;  // mgp_common.h (shared with cg_dense1.hip)

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // red reuse
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T s = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) s += red[w];  // fixed order: deterministic
  return s;
}

struct PrecondDev {
  int kind;
  int bs;
  long nb;
  const void* diag_inv;
  const long* block_index;
  const void* block_inv;
};

// z = M^-1 r for one RHS row (all threads of the block cooperate); returns nothing, z in memory
template <typename T>
__device__ void apply_precond(const PrecondDev& pc, const T* __restrict__ r, T* __restrict__ z, long n) {
  if (pc.kind == MGP_PRE_JACOBI) {
    const T* dinv = (const T*)pc.diag_inv;
    for (long j = threadIdx.x; j < n; j += blockDim.x) z[j] = r[j] * dinv[j];
  } else {  // MGP_PRE_BLOCK: uncovered indices pass through, covered ones get Binv_k r[idx_k]
    for (long j = threadIdx.x; j < n; j += blockDim.x) z[j] = r[j];
    __syncthreads();
    const T* binv = (const T*)pc.block_inv;
    const long tot = pc.nb * pc.bs;
    for (long e = threadIdx.x; e < tot; e += blockDim.x) {
      const long k = e / pc.bs, a = e - k * pc.bs;
      const long* idx = pc.block_index + k * pc.bs;
      const T* row = binv + (k * pc.bs + a) * pc.bs;
      T s = 0;
      for (int c = 0; c < pc.bs; ++c) s = mgp_fma(row[c], r[idx[c]], s);
      z[idx[a]] = s;
    }
  }
  __syncthreads();
}

// r = b - av (av may be null => r = b); z, rz, p, stopping flag.  Also used for the refresh step.
template <typename T>
__global__ __launch_bounds__(256) void cg_init_kernel(const T* __restrict__ b, const T* __restrict__ av,
                                                      T* __restrict__ r, T* __restrict__ z, T* __restrict__ p,
                                                      T* __restrict__ rz, int* __restrict__ over,
                                                      T* __restrict__ err, long n, T thr, PrecondDev pc) {
  __shared__ T red[8];
  const long off = (long)blockIdx.x * n;
  for (long j = threadIdx.x; j < n; j += blockDim.x) r[off + j] = av ? b[off + j] - av[off + j] : b[off + j];
  __syncthreads();
  const T* zz = r + off;
  if (pc.kind != MGP_PRE_EYE) {
    apply_precond<T>(pc, r + off, z + off, n);
    zz = z + off;
  }
  T s_rz = 0, s_rr = 0;
  for (long j = threadIdx.x; j < n; j += blockDim.x) {
    const T rv = r[off + j], zv = zz[j];
    p[off + j] = zv;
    s_rz = mgp_fma(zv, rv, s_rz);
    s_rr = mgp_fma(rv, rv, s_rr);
  }
  s_rz = block_sum(s_rz, red);
  s_rr = block_sum(s_rr, red);
  if (threadIdx.x == 0) {
    rz[blockIdx.x] = s_rz;
    err[blockIdx.x] = (T)0.5 * s_rz;
    over[blockIdx.x] = ((T)0.5 * s_rr > thr) ? 1 : 0;
  }
}

// Generic step pieces (refresh steps, block / dense preconditioners).  Modes:
//   0  first half (gamma, v, r -= gamma Ap) + native preconditioner + second half
//   1  first half without the r update (a residual refresh follows)
//   2  native preconditioner + second half with beta = 0 (after a refresh: p = z)
//   3  first half only, with the r update (a dense preconditioner product follows)
//   4  second half with z already in memory (dense preconditioner), normal beta
//   5  second half with z already in memory, beta = 0 (start-up and refresh: p = z)
// `force` bypasses the gate (start-up of the dense-preconditioner path).
// Block size: 256 threads per right-hand side up to n = 8192, 1024 beyond (the fused kernel covers n <= 8192; one
// 256-thread workgroup per right-hand side was a cliff for larger systems).
// NTB is the launch bound AND the launch size (256 or 1024): small systems keep the register budget of a 256-thread
// workgroup (one 1024-thread bound capped it for every launch).
template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void cg_update_kernel(const CgCtrl* __restrict__ ctrl, T* __restrict__ v,
                                                        T* __restrict__ r, T* __restrict__ p,
                                                        T* __restrict__ z, const T* __restrict__ ap,
                                                        T* __restrict__ rz, int* __restrict__ over,
                                                        T* __restrict__ err, long n, T thr, T min_float,
                                                        PrecondDev pc, int mode, int force) {
  if (!force && ctrl->active == 0) return;
  __shared__ T red[16];
  const long off = (long)blockIdx.x * n;
  const T rz_old = rz[blockIdx.x];
  const bool first = mode == 0 || mode == 1 || mode == 3;
  if (first) {
    T d = 0;
    for (long j = threadIdx.x; j < n; j += blockDim.x) d = mgp_fma(p[off + j], ap[off + j], d);
    d = block_sum(d, red);
    const T gamma = (d <= min_float) ? (T)0 : rz_old / d;
    for (long j = threadIdx.x; j < n; j += blockDim.x) {
      v[off + j] = mgp_fma(gamma, p[off + j], v[off + j]);
      if (mode != 1) r[off + j] = mgp_fma(-gamma, ap[off + j], r[off + j]);
    }
    if (mode != 0) return;
    __syncthreads();
  }
  const T* zz = r + off;
  if (mode >= 4) {
    zz = z + off;  // computed by the dense preconditioner product
  } else if (pc.kind != MGP_PRE_EYE) {
    apply_precond<T>(pc, r + off, z + off, n);
    zz = z + off;
  }
  T s_rz = 0, s_rr = 0;
  for (long j = threadIdx.x; j < n; j += blockDim.x) {
    const T rv = r[off + j], zv = zz[j];
    s_rz = mgp_fma(zv, rv, s_rz);
    s_rr = mgp_fma(rv, rv, s_rr);
  }
  s_rz = block_sum(s_rz, red);
  s_rr = block_sum(s_rr, red);
  const T beta = (mode == 2 || mode == 5 || rz_old <= min_float) ? (T)0 : s_rz / rz_old;
  // p = z where the beta-term is dropped (:79-84): a select, not 0 * p -- at start-up (mode 5) p is
  // whatever the arena held, and 0 * Inf/NaN would poison the direction
  const bool drop = mode == 2 || mode == 5 || rz_old <= min_float;
  for (long j = threadIdx.x; j < n; j += blockDim.x) p[off + j] = drop ? zz[j] : mgp_fma(beta, p[off + j], zz[j]);
  if (threadIdx.x == 0) {
    rz[blockIdx.x] = s_rz;
    err[blockIdx.x] = (T)0.5 * s_rz;
    over[blockIdx.x] = ((T)0.5 * s_rr > thr) ? 1 : 0;
  }
}

// Fused common-case step (no refresh, Eye or Jacobi): the whole update of one RHS with its
// elements held in registers (EPT per thread), two block reductions instead of four, and the
// iteration bookkeeping (`any` over the RHS flags, step counter, next gate) done by the last
// workgroup to arrive -- write-through flag + ticket, no extra launch.
template <typename T, int EPT, int NT>
__global__ __launch_bounds__(NT) void cg_update_fused_kernel(CgCtrl* __restrict__ ctrl, T* __restrict__ v,
                                                             T* __restrict__ r, T* __restrict__ p,
                                                             const T* __restrict__ ap, T* __restrict__ rz,
                                                             int* __restrict__ over, T* __restrict__ err, long n,
                                                             T thr, T min_float, const T* __restrict__ dinv,
                                                             int max_it, int ap_slices, long ap_stride,
                                                             const T* __restrict__ agree, int world) {
  // multi-rank SGPR operator: `ap` is the all-reduced partial itself and `agree` the word behind it -- the sum of
  // the ranks' gate words.  Unless every rank computed this application nobody uses it: the gate closes and all
  // ranks leave the loop on the same iteration (what finish_allreduce_kernel did in a launch of its own).
  // Tested BEFORE the gate word: in this branch workgroup 0 clears ctrl->active, and no workgroup of the launch may be
  // reading that word at its entry meanwhile -- here none does, every one of them leaves on `agree`.  The
  // other writer of ctrl->active, the last arriver below, runs after every workgroup has passed this point.
  if (agree != nullptr && *agree != (T)world) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->active = 0;
    return;
  }
  if (ctrl->active == 0) return;
  __shared__ T red[2][NT / 64];
  __shared__ int last_flag;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long off = (long)blockIdx.x * n;
  const T rz_old = rz[blockIdx.x];
  T pv[EPT], av[EPT], rv[EPT], vv[EPT], dv[EPT];
  T d = 0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const long j = (long)e * NT + t;
    const bool ok = j < n;
    pv[e] = ok ? p[off + j] : (T)0;
    av[e] = ok ? ap[off + j] : (T)0;
    rv[e] = ok ? r[off + j] : (T)0;
    // the solution and the Jacobi diagonal are not needed before gamma is known, but requested now: behind the
    // block reduction they were a third dependent round trip to memory in a kernel that is nothing but latency
    vv[e] = ok ? v[off + j] : (T)0;
    dv[e] = (dinv != nullptr && ok) ? dinv[j] : (T)0;
  }
  if (ap_slices > 1) {
    // A.p left as contraction slices by the skinny product (at most 8): every slice of every element is requested
    // before the first is used -- one workgroup per right-hand side is latency-bound, a load-add chain per element
    // cost what the separate reduce launch had -- then added in slice order, as skinny_reduce_kernel does
    T sl[7][EPT];
#pragma unroll
    for (int z = 1; z < 8; ++z)
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const long j = (long)e * NT + t;
        sl[z - 1][e] = (z < ap_slices && j < n) ? ap[(long)z * ap_stride + off + j] : (T)0;
      }
#pragma unroll
    for (int z = 1; z < 8; ++z)
      if (z < ap_slices) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) av[e] += sl[z - 1][e];
      }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) d = mgp_fma(pv[e], av[e], d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if (lane == 0) red[0][wave] = d;
  __syncthreads();
  d = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) d += red[0][w];
  const T gamma = (d <= min_float) ? (T)0 : rz_old / d;
  T s_rz = 0, s_rr = 0;
  T zv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const long j = (long)e * NT + t;
    if (j < n) v[off + j] = mgp_fma(gamma, pv[e], vv[e]);
    rv[e] = mgp_fma(-gamma, av[e], rv[e]);
    zv[e] = dinv != nullptr ? rv[e] * dv[e] : rv[e];
    s_rz = mgp_fma(zv[e], rv[e], s_rz);
    s_rr = mgp_fma(rv[e], rv[e], s_rr);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_rz += __shfl_xor(s_rz, o, 64);
    s_rr += __shfl_xor(s_rr, o, 64);
  }
  __syncthreads();  // red[0] fully read
  if (lane == 0) {
    red[0][wave] = s_rz;
    red[1][wave] = s_rr;
  }
  __syncthreads();
  s_rz = 0;
  s_rr = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    s_rz += red[0][w];
    s_rr += red[1][w];
  }
  const T beta = (rz_old <= min_float) ? (T)0 : s_rz / rz_old;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const long j = (long)e * NT + t;
    if (j < n) {
      r[off + j] = rv[e];
      p[off + j] = (rz_old <= min_float) ? zv[e] : mgp_fma(beta, pv[e], zv[e]);
    }
  }
  if (t == 0) {
    rz[blockIdx.x] = s_rz;
    err[blockIdx.x] = (T)0.5 * s_rz;
    __hip_atomic_store(&over[blockIdx.x], ((T)0.5 * s_rr > thr) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gridDim.x == 1) {
      last_flag = 1;  // one right-hand side: nothing to hand over
    } else {
      // hand-off without fences (cdna_hip_programming.md G16, sc1 form): the flag above is a
      // write-through store of this lane, drained here, then the ticket; the last arriver reads the
      // flags with sc1 loads behind the barrier.  (__threadfence() on both sides cost ~3.5 us each.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned tk = atomicAdd(&ctrl->ticket, 1u);
      last_flag = (tk == gridDim.x - 1) ? 1 : 0;
    }
  }
  __syncthreads();
  if (last_flag) {  // last workgroup to arrive: every flag has been published
    __shared__ int any;
    if (t == 0) any = 0;
    __syncthreads();
    int a = 0;
    for (long b = t; b < (long)gridDim.x; b += NT)
      a |= __hip_atomic_load(&over[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a) atomicOr(&any, 1);
    __syncthreads();
    if (t == 0) {
      const int it = ctrl->iters + 1;
      ctrl->iters = it;
      ctrl->ticket = 0;
      ctrl->active = (any && it < max_it) ? 1 : 0;
    }
  }
}

// r = b - av, gated (refresh step, conjugate_gradient.py:72-75)
template <typename T>
__global__ __launch_bounds__(256) void cg_residual_kernel(const CgCtrl* __restrict__ ctrl,
                                                          const T* __restrict__ b, const T* __restrict__ av,
                                                          T* __restrict__ r, long total, int force) {
  if (!force && ctrl->active == 0) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) r[i] = av ? b[i] - av[i] : b[i];
}

__global__ void cg_advance_kernel(CgCtrl* ctrl, const int* __restrict__ over, long Bt, int inc, int max_it) {
  __shared__ int any;
  if (inc && ctrl->active == 0) return;  // uniform: every thread reads the same word
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int a = 0;
  for (long b = threadIdx.x; b < Bt; b += blockDim.x) a |= over[b];
  if (a) atomicOr(&any, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int it = ctrl->iters + inc;
    ctrl->iters = it;
    ctrl->active = (any && it < max_it) ? 1 : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const int* __restrict__ gate, T a, const T* __restrict__ x, T b,
                                                    const T* __restrict__ y, T* __restrict__ out, long total) {
  if (gate != nullptr && *gate == 0) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = a * x[i] + b * y[i];
}

// t[b, j] += a * x[b, j] for the columns j in [rb, re)
template <typename T>
__global__ __launch_bounds__(256) void add_rows_slab_kernel(const int* __restrict__ gate, T a, const T* __restrict__ x,
                                                            T* __restrict__ t, long n, long total, long rb, long re) {
  if (gate != nullptr && *gate == 0) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long j = i % n;
  if (j >= rb && j < re) t[i] = mgp_fma(a, x[i], t[i]);
}

// out[b, j] += lam[j] * p[b, j]
template <typename T>
__global__ __launch_bounds__(256) void add_diag_prod_kernel(const int* __restrict__ gate, const T* __restrict__ lam,
                                                            const T* __restrict__ p, T* __restrict__ out, long n,
                                                            long total) {
  if (gate != nullptr && *gate == 0) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = mgp_fma(lam[i % n], p[i], out[i]);
}

// Multi-rank agreement: the word behind the [Bt,M] partial carries this rank's gate
// (1 = it computed this application) through the all-reduce; the sum equals the number of ranks
// iff every rank did.
template <typename T>
__global__ void put_gate_word_kernel(const int* __restrict__ gate, T* __restrict__ word) {
  *word = (gate == nullptr || *gate != 0) ? (T)1 : (T)0;
}

// After the all-reduce: out = reduced partial when every rank took part; otherwise nobody uses this
// application and the local gate closes, so all ranks leave the CG loop on the same iteration.
template <typename T>
__global__ __launch_bounds__(256) void finish_allreduce_kernel(int* __restrict__ gate, const T* __restrict__ tt,
                                                               T* __restrict__ out, long tot, int world) {
  const int a = gate == nullptr ? 1 : *gate;
  if (!a) return;
  if (tt[tot] != (T)world) {  // exact: a sum of at most `world` ones
    if (gate != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *gate = 0;
    return;
  }
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tot) out[i] = tt[i];
}

inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }


// out[Bt, n] = P[Bt, n] @ Op ; gate may be null
template <typename T>
int apply_operator(mgp_handle* h, const mgp_operator* op, const T* P, long Bt, T* out, const int* gate) {
  const long n = op->n;
  switch (op->kind) {
    case MGP_OP_DENSE:
      return mgp_symm_matmul_gated(h, op->dtype, op->A, n, P, Bt, out, gate);
    case MGP_OP_KMM_LAMBDA: {
      MGP_TRY(mgp_sweep(h, op->kernel, op->Z, op->M, op->Z, op->M, VecView{P, 1, n}, (int)Bt,
                        VecViewMut{out, 1, n}, 0.0, VecView{nullptr, 0, 0}, gate));
      hipLaunchKernelGGL((add_diag_prod_kernel<T>), dim3(nblk(Bt * n)), dim3(256), 0, h->stream, gate,
                         (const T*)op->lambda, P, out, n, Bt * n);
      MGP_LAUNCH_CHECK(h);
      return MGP_OK;
    }
    case MGP_OP_SGPR: {
      // u[Bt,N] = (K_nm p^T)^T ; t[Bt,M] = (K_mn u^T)^T over the local rows ; the replicated
      // s2*Kmm.p term is added as this rank's row slab of it, so ONE all-reduce of t finishes S.p
      const long N = op->N, M = op->M;
      const bool coll = op->allreduce != nullptr || op->comm != nullptr;
      // operator arena: u [Bt,N] | partial [Bt,M] + agreement word (padded to 2) | Kmm.p [Bt,M] (only for Bt > 1)
      const size_t need = ((size_t)Bt * N + 2 * (size_t)Bt * M + 2) * sizeof(T);
      MGP_TRY(mgp_reserve(h, &h->opws, &h->opws_bytes, need));
      T* u = (T*)h->opws;
      // the collective works on the caller's buffer when one is given; single rank: straight into out
      T* tt = coll ? (op->partial_buf ? (T*)op->partial_buf : u + Bt * N) : out;
      long rb = op->kmm_row_begin, re = op->kmm_row_end;
      if (rb == 0 && re == 0) re = M;  // unset: this rank owns every row of Kmm
      bool word_written = false;
      // One right-hand side: this rank's slab of Kmm.p does not depend on the sweeps -- it goes to a stream of its own,
      // forked when p is ready, and comes back as the addend of the K_mn sweep: 31 us of HBM streaming at M = 4096
      // beside 2.3 ms of vector-ALU-bound sweep instead of behind it
      // (worth its two events from ~50 MB of slab on: one rank's share of 8 at M = 4096 is 17 MB = 4 us, measured 5 us slower)
      const bool aside = Bt == 1 && N > 0 && re > rb &&
                         (h->kmm_aside > 1 || (h->kmm_aside == 1 && (size_t)(re - rb) * M * sizeof(T) >= ((size_t)48 << 20)));
      if (aside) {
        if (!h->aside_stream) MGP_HIP(h, hipStreamCreateWithFlags(&h->aside_stream, hipStreamNonBlocking));
        for (int e = 0; e < 2; ++e)
          if (!h->aside_ev[e]) MGP_HIP(h, hipEventCreateWithFlags(&h->aside_ev[e], hipEventDisableTiming));
        T* kmp = u + Bt * N + Bt * M + 2;
        hipStream_t main_stream = h->stream;
        MGP_HIP(h, hipEventRecord(h->aside_ev[0], main_stream));
        MGP_HIP(h, hipStreamWaitEvent(h->aside_stream, h->aside_ev[0], 0));
        h->stream = h->aside_stream;
        word_written = coll && h->fuse_agree;  // the slab product also writes this rank's agreement word behind the partial
        int rc = MGP_OK;
        if (rb == 0 && re == M && !word_written) {  // every row is this rank's: written, not accumulated
          rc = mgp_symm_gemv_assign(h, op->dtype, op->Kmm, M, P, kmp, gate);
        } else {
          if (hipMemsetAsync(kmp, 0, (size_t)M * sizeof(T), h->stream) != hipSuccess) rc = mgp_fail(h, MGP_E_HIP, "memset of Kmm.p failed");
          if (rc == MGP_OK)
            rc = mgp_symm_gemv_rows_acc(h, op->dtype, op->Kmm, M, P, rb, re, 1.0, kmp, gate,
                                        word_written ? (void*)(tt + Bt * M) : nullptr);
        }
        h->stream = main_stream;
        MGP_TRY(rc);
        MGP_HIP(h, hipEventRecord(h->aside_ev[1], h->aside_stream));
        MGP_TRY(mgp_sweep(h, op->kernel, op->X, N, op->Z, M, VecView{P, 1, M}, (int)Bt, VecViewMut{u, 1, N}, 0.0,
                          VecView{nullptr, 0, 0}, gate));
        MGP_HIP(h, hipStreamWaitEvent(main_stream, h->aside_ev[1], 0));
        MGP_TRY(mgp_sweep(h, op->kernel, op->Z, M, op->X, N, VecView{u, 1, N}, (int)Bt, VecViewMut{tt, 1, M}, op->s2,
                          VecView{kmp, 1, M}, gate));
      } else if (N > 0) {
        MGP_TRY(mgp_sweep(h, op->kernel, op->X, N, op->Z, M, VecView{P, 1, M}, (int)Bt, VecViewMut{u, 1, N}, 0.0,
                          VecView{nullptr, 0, 0}, gate));
        MGP_TRY(mgp_sweep(h, op->kernel, op->Z, M, op->X, N, VecView{u, 1, N}, (int)Bt, VecViewMut{tt, 1, M}, 0.0,
                          VecView{nullptr, 0, 0}, gate));
      } else {
        MGP_HIP(h, hipMemsetAsync(tt, 0, (size_t)Bt * M * sizeof(T), h->stream));
      }
      if (aside) {
        // Kmm.p is the K_mn sweep's addend already
      } else if (Bt == 1) {
        // with a collective the slab product also writes this rank's agreement word behind the partial
        word_written = coll && re > rb && h->fuse_agree;
        MGP_TRY(mgp_symm_gemv_rows_acc(h, op->dtype, op->Kmm, M, P, rb, re, op->s2, tt, gate,
                                       word_written ? (void*)(tt + Bt * M) : nullptr));
      } else {
        // several RHS: full replicated product, then only this rank's slab of it is added
        T* kmp = u + Bt * N + Bt * M + 2;
        MGP_TRY(mgp_symm_matmul_gated(h, op->dtype, op->Kmm, M, P, Bt, kmp, gate));
        hipLaunchKernelGGL((add_rows_slab_kernel<T>), dim3(nblk(Bt * M)), dim3(256), 0, h->stream, gate, (T)op->s2,
                           (const T*)kmp, tt, M, Bt * M, rb, re);
        MGP_LAUNCH_CHECK(h);
      }
      if (coll) {
        const long tot = Bt * M;
        if (!word_written) {
          hipLaunchKernelGGL((put_gate_word_kernel<T>), dim3(1), dim3(1), 0, h->stream, gate, tt + tot);
          MGP_LAUNCH_CHECK(h);
        }
        if (op->allreduce) {  // rehearsal hook (gloo with host staging); never used with RCCL
          const int rc = op->allreduce(op->allreduce_ctx, tt, (size_t)(tot + 1), op->dtype, (void*)h->stream);
          if (rc != 0) return mgp_fail(h, MGP_E_COMM, "allreduce callback returned %d", rc);
        } else {  // native: one ncclAllReduce on the solve's stream
          MGP_TRY(mgp_comm_allreduce_on(h, op->comm, tt, (size_t)(tot + 1), op->dtype));
        }
        const int world = op->comm ? mgp_comm_size(op->comm) : op->world_size;
        if (h->defer_finish && gate != nullptr) {  // the fused update tests the word and reads the partial in place
          h->deferred_tt = tt;
          h->deferred_world = world;
          return MGP_OK;
        }
        hipLaunchKernelGGL((finish_allreduce_kernel<T>), dim3(nblk(tot)), dim3(256), 0, h->stream, (int*)gate,
                           (const T*)tt, out, tot, world);
        MGP_LAUNCH_CHECK(h);
      }
      return MGP_OK;
    }
    default:
      return mgp_fail(h, MGP_E_BADARG, "unknown operator kind %d", op->kind);
  }
}

int check_operator(mgp_handle* h, const mgp_operator* op) {
  if (!op) return mgp_fail(h, MGP_E_BADARG, "operator is NULL");
  if (op->dtype != MGP_F32 && op->dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad operator dtype");
  if (op->n <= 0) return mgp_fail(h, MGP_E_SHAPE, "operator n must be > 0");
  if (op->kind == MGP_OP_DENSE) {
    if (!op->A) return mgp_fail(h, MGP_E_BADARG, "dense operator without matrix");
  } else if (op->kind == MGP_OP_SGPR || op->kind == MGP_OP_KMM_LAMBDA) {
    MGP_TRY(mgp_check_kernel(h, op->kernel));
    if (op->kernel->dtype != op->dtype) return mgp_fail(h, MGP_E_DTYPE, "operator/kernel dtype mismatch");
    if (op->M != op->n || !op->Z) return mgp_fail(h, MGP_E_SHAPE, "operator needs Z with M == n");
    if (op->kind == MGP_OP_SGPR && (!op->Kmm || op->N < 0 || (op->N > 0 && !op->X)))
      return mgp_fail(h, MGP_E_BADARG, "SGPR operator needs X and Kmm");
    if (op->kind == MGP_OP_KMM_LAMBDA && !op->lambda) return mgp_fail(h, MGP_E_BADARG, "needs lambda");
    if (op->kind == MGP_OP_SGPR && op->allreduce && op->comm)
      return mgp_fail(h, MGP_E_BADARG, "give either the allreduce hook or a communicator, not both");
    if (op->kind == MGP_OP_SGPR && op->allreduce && op->world_size < 1)
      return mgp_fail(h, MGP_E_BADARG, "allreduce hook needs world_size >= 1");
  } else {
    return mgp_fail(h, MGP_E_BADARG, "unknown operator kind %d", op->kind);
  }
  return MGP_OK;
}

constexpr int kRetryWithoutPersist = 1;  // internal: never crosses the C ABI

template <typename T>
int pcg_solve_t(mgp_handle* h, const mgp_operator* op, const mgp_precond* pre, const T* B, const T* V0, long Bt,
                double thr, long max_it, long cycle, double min_float, int check_every, T* V, T* err_out,
                mgp_cg_stats* stats) {
  const long n = op->n;
  const long tot = Bt * n;
  const auto t0 = std::chrono::steady_clock::now();
  PackHold pack_hold(h);  // the operator's X / Z / kernel are constant for the whole solve
  PrecondDev pc{MGP_PRE_EYE, 0, 0, nullptr, nullptr, nullptr};
  const void* dense_inv = nullptr;  // MGP_PRE_DENSE: z = r @ Pinv through the symmetric product kernels
  const mgp_precond* cb = nullptr;  // MGP_PRE_CALLBACK: z produced by the caller's function, same step structure
  if (pre) {
    pc.kind = pre->kind;
    if (pre->kind == MGP_PRE_JACOBI) {
      if (!pre->diag_inv) return mgp_fail(h, MGP_E_BADARG, "jacobi preconditioner without diag_inv");
      pc.diag_inv = pre->diag_inv;
    } else if (pre->kind == MGP_PRE_BLOCK) {
      if (!pre->block_index || !pre->block_inv || pre->block_size <= 0 || pre->num_blocks < 0)
        return mgp_fail(h, MGP_E_BADARG, "block preconditioner incomplete");
      pc.bs = pre->block_size;
      pc.nb = pre->num_blocks;
      pc.block_index = (const long*)pre->block_index;
      pc.block_inv = pre->block_inv;
    } else if (pre->kind == MGP_PRE_DENSE) {
      if (!pre->dense_inv) return mgp_fail(h, MGP_E_BADARG, "dense preconditioner without matrix");
      dense_inv = pre->dense_inv;
    } else if (pre->kind == MGP_PRE_CALLBACK) {
      if (!pre->apply || !pre->cb_r || !pre->cb_z)
        return mgp_fail(h, MGP_E_BADARG, "callback preconditioner needs apply, cb_r and cb_z");
      cb = pre;
    } else if (pre->kind != MGP_PRE_EYE) {
      return mgp_fail(h, MGP_E_BADARG, "unknown preconditioner kind %d", pre->kind);
    }
  }
  const bool dense_pre = dense_inv != nullptr || cb != nullptr;  // z comes from outside the update kernels
  const bool need_z = pc.kind != MGP_PRE_EYE && cb == nullptr;   // dense: z = r @ Pinv lives in the arena
  if (dense_pre) pc.kind = MGP_PRE_EYE;                          // the update kernels never apply it themselves
  // arena: r, p, ap, [z], rz[Bt], over[Bt] (int), ctrl
  // one right-hand side on a dense matrix, no residual refresh inside the solve: the two-launch iteration of
  // cg_dense1.hip (tile shares, chunk shares, two direction buffers live behind the control word)
  // ... and, since round 4, two to eight right-hand sides on the same tile scheme (the reference's default num_probes = 5)
  // ... up to 8 where the full matrix fits the chip (n <= 2048: the register-resident form carries the columns for 16
  // fused multiply-adds each), else where the tile scheme was measured faster than the skinny product: 4, 6 above 4096
  const long d1_cols = h->cg_dense1_cols > 0 ? h->cg_dense1_cols
                       : ((!h->d1_persist_off && mgp_dense1_persist_eligible(h, n, 8))   ? 8
                          : (!h->d1_persist_off && mgp_dense1_persist_eligible(h, n, 6)) ? 6
                                                                                         : (n <= 4096 ? 4 : 6));
  const bool dense1 = op->kind == MGP_OP_DENSE && Bt >= 1 && Bt <= d1_cols && !dense_pre &&
                      pc.kind != MGP_PRE_BLOCK && cycle > max_it && mgp_dense1_eligible(h, n);
  size_t bytes = (size_t)tot * sizeof(T) * (need_z ? 4 : 3) + (size_t)Bt * sizeof(T) + (size_t)Bt * sizeof(int) + 64 +
                 (dense1 ? mgp_dense1_bytes(h, op->dtype, n, Bt) : 0);
  MGP_TRY(mgp_reserve(h, &h->cg, &h->cg_bytes, bytes));
  T* r = (T*)h->cg;
  T* p = r + tot;
  T* ap = p + tot;
  T* z = need_z ? ap + tot : (cb ? (T*)cb->cb_z : nullptr);
  T* rz = (need_z ? ap + tot : ap) + tot;
  int* over = (int*)(rz + Bt);
  CgCtrl* ctrl = (CgCtrl*)(((uintptr_t)(over + Bt) + 15) & ~(uintptr_t)15);
  void* d1_arena = (void*)(((uintptr_t)(ctrl + 1) + 15) & ~(uintptr_t)15);
  hipStream_t s = h->stream;
  const unsigned upd_threads = n > 8192 ? 1024u : 256u;  // generic update kernel: threads per right-hand side
#define MGP_UPDATE_LAUNCH(...)                                                                                   \
  do {                                                                                                           \
    if (upd_threads == 1024u)                                                                                    \
      hipLaunchKernelGGL((cg_update_kernel<T, 1024>), dim3((unsigned)Bt), dim3(1024), 0, s, __VA_ARGS__);        \
    else                                                                                                         \
      hipLaunchKernelGGL((cg_update_kernel<T, 256>), dim3((unsigned)Bt), dim3(256), 0, s, __VA_ARGS__);          \
  } while (0)
  // z = M^-1 r for the preconditioners applied outside the update kernels
  auto external_z = [&](const int* gate) -> int {
    if (cb) {
      MGP_HIP(h, hipMemcpyAsync(cb->cb_r, r, (size_t)tot * sizeof(T), hipMemcpyDeviceToDevice, s));
      const int rc = cb->apply(cb->apply_ctx, cb->cb_r, cb->cb_z, Bt, n, (void*)s);
      if (rc != 0) return mgp_fail(h, MGP_E_BADARG, "preconditioner callback returned %d", rc);
      return MGP_OK;
    }
    return mgp_symm_matmul_gated(h, op->dtype, dense_inv, n, r, Bt, z, gate);
  };

  MGP_HIP(h, hipMemsetAsync(ctrl, 0, sizeof(CgCtrl), s));
  const T* av = nullptr;
  if (V0) {
    if (V != V0) MGP_HIP(h, hipMemcpyAsync(V, V0, (size_t)tot * sizeof(T), hipMemcpyDeviceToDevice, s));
    MGP_TRY(apply_operator<T>(h, op, V, Bt, ap, nullptr));  // vA (:87)
    av = ap;
  } else {
    MGP_HIP(h, hipMemsetAsync(V, 0, (size_t)tot * sizeof(T), s));
  }
  MgpDense1 d1;
  // n <= 4096: the whole solve in one launch, the upper triangle of A in registers (cg_dense1.hip)
  const bool persist = dense1 && !h->d1_persist_off && mgp_dense1_persist_eligible(h, n, Bt);
  if (dense1) {
    MGP_TRY(mgp_dense1_begin(h, &d1, op->dtype, op->A, n, B, av, V, r,
                             pc.kind == MGP_PRE_JACOBI ? pc.diag_inv : nullptr, ctrl, d1_arena, thr, min_float, max_it,
                             persist ? 1 : 0, (int)Bt));
    MGP_TRY(mgp_dense1_finish(h, &d1, rz, err_out, over));  // statistics of r_0 and the first gate
    if (persist) {
      MGP_TRY(mgp_dense1_persist_run(h, &d1));
      MGP_TRY(mgp_dense1_finish(h, &d1, rz, err_out, over));  // final statistics; closes the gate
    }
  } else if (!dense_pre) {
    hipLaunchKernelGGL((cg_init_kernel<T>), dim3((unsigned)Bt), dim3(256), 0, s, B, av, r, z, p, rz, over, err_out,
                       n, (T)thr, pc);
    MGP_LAUNCH_CHECK(h);
  } else {  // r = b - vA ; z = r @ Pinv ; p = z, rz = z.r, flags
    hipLaunchKernelGGL((cg_residual_kernel<T>), dim3(nblk(tot)), dim3(256), 0, s, ctrl, B, av, r, tot, 1);
    MGP_LAUNCH_CHECK(h);
    MGP_TRY(external_z(nullptr));
    MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz, over,
                       err_out, n, (T)thr, (T)min_float, pc, 5, 1);
    MGP_LAUNCH_CHECK(h);
  }
  if (!dense1) {
    hipLaunchKernelGGL(cg_advance_kernel, dim3(1), dim3(256), 0, s, ctrl, over, Bt, 0, (int)max_it);
    MGP_LAUNCH_CHECK(h);
  }

  // fused step kernel: elements of one RHS in registers.  Code = EPT for 256 threads (n <= 1024),
  // 14/12/24/8 for 1024 threads with EPT 1/2/4/8 (n <= 8192); 0 = generic loop kernels.
  int fused_ept = 0;
  if (pc.kind != MGP_PRE_BLOCK && !dense_pre && Bt < 2147483647L) {
    if (n <= 256) fused_ept = 1;
    else if (n <= 512) fused_ept = 2;
    else if (n <= 1024) fused_ept = 14;
    else if (n <= 2048) fused_ept = 12;
    else if (n <= 4096) fused_ept = 24;
    else if (n <= 8192) fused_ept = 8;
  }
  if (check_every < 1) check_every = 1;
  long enq = 0;  // iterations enqueued so far (index of the next one)
  CgCtrl host{1, 0};
  if (persist) {
    MGP_HIP(h, hipMemcpyAsync(h->host_flag, ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, s));
    MGP_HIP(h, hipStreamSynchronize(s));
    memcpy(&host, h->host_flag, sizeof(CgCtrl));
    // a hand-off ran out of its poll budget (a workgroup was not resident: the chip was shared): the caller retries
    // this solve with the two-launch form
    if (host.pad) return kRetryWithoutPersist;
  } else if (dense1 && h->poll_pipeline) {
    // One polled batch stays in flight: the device works on batch i + 1 while the host waits for the control word of
    // batch i (round 3: every poll drained the stream, ~20 us each, 10 of them in a 248-step solve).  When a batch
    // reports the end, the one behind it is already enqueued -- its launches are gated off on the device.
    MGP_HIP(h, hipMemcpyAsync(h->host_flag, ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, s));
    MGP_HIP(h, hipStreamSynchronize(s));
    memcpy(&host, h->host_flag, sizeof(CgCtrl));
    for (int e = 0; e < 2; ++e)
      if (!h->poll_ev[e]) MGP_HIP(h, hipEventCreateWithFlags(&h->poll_ev[e], hipEventDisableTiming));
    CgCtrl* slots = (CgCtrl*)h->host_flag;  // 64 pinned bytes: two control words fit behind the first
    int inflight = 0, wr = 0, rd = 0;
    bool more = host.active != 0;
    while (more || inflight > 0) {
      while (more && inflight < 2 && enq < max_it) {
        long batch = check_every;
        if (enq + batch > max_it) batch = max_it - enq;
        for (long q = 0; q < batch; ++q, ++enq) MGP_TRY(mgp_dense1_step(h, &d1, enq + 1));
        MGP_TRY(mgp_dense1_finish(h, &d1, rz, err_out, over));
        MGP_HIP(h, hipMemcpyAsync(&slots[1 + wr], ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, s));
        MGP_HIP(h, hipEventRecord(h->poll_ev[wr], s));
        wr ^= 1;
        ++inflight;
      }
      if (inflight == 0) break;
      MGP_HIP(h, hipEventSynchronize(h->poll_ev[rd]));
      memcpy(&host, &slots[1 + rd], sizeof(CgCtrl));
      rd ^= 1;
      --inflight;
      if (host.pad) return mgp_fail(h, MGP_E_HIP, "dense CG: a hand-off inside the iteration kernel timed out");
      if (!host.active || enq >= max_it) more = false;
      if (!host.active) break;  // what is still in flight does nothing (device gate); no need to wait for it
    }
  }
  while (!persist && !(dense1 && h->poll_pipeline)) {
    MGP_HIP(h, hipMemcpyAsync(h->host_flag, ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, s));
    MGP_HIP(h, hipStreamSynchronize(s));
    memcpy(&host, h->host_flag, sizeof(CgCtrl));
    if (dense1 && host.pad) return mgp_fail(h, MGP_E_HIP, "dense CG: a hand-off inside the iteration kernel timed out");
    if (!host.active || enq >= max_it) break;
    long batch = check_every;
    if (enq + batch > max_it) batch = max_it - enq;
    if (dense1) {
      for (long q = 0; q < batch; ++q, ++enq) MGP_TRY(mgp_dense1_step(h, &d1, enq + 1));
      MGP_TRY(mgp_dense1_finish(h, &d1, rz, err_out, over));
      continue;
    }
    for (long q = 0; q < batch; ++q, ++enq) {
      const bool reset = (enq % cycle) == (cycle - 1);  // :71 (enq == state.i while active)
      // dense operator + fused update: the skinny product may leave its slices for the update to add
      const bool defer = !reset && fused_ept > 0 && op->kind == MGP_OP_DENSE && h->skinny_defer;
      h->defer_slices = defer;
      h->deferred_ks = 1;
      h->defer_finish = !reset && fused_ept > 0 && op->kind == MGP_OP_SGPR && h->fuse_agree;
      h->deferred_tt = nullptr;
      const int rc_apply = apply_operator<T>(h, op, p, Bt, ap, &ctrl->active);
      h->defer_slices = false;
      h->defer_finish = false;
      MGP_TRY(rc_apply);
      if (!reset && fused_ept > 0) {
        const T* dinv = pc.kind == MGP_PRE_JACOBI ? (const T*)pc.diag_inv : nullptr;
        const T* ap_src = ap;
        int ap_slices = 1;
        long ap_stride = 0;
        if (defer && h->deferred_ks > 1) {
          ap_src = (const T*)h->deferred_part;
          ap_slices = h->deferred_ks;
          ap_stride = h->deferred_stride;
        }
        const T* agree = nullptr;
        int world = 0;
        if (h->deferred_tt != nullptr) {  // all-reduced partial left in place by the operator: [Bt, n] + word
          ap_src = (const T*)h->deferred_tt;
          agree = ap_src + tot;
          world = h->deferred_world;
        }
#define MGP_FUSED(EPTV, NTV)                                                                                   \
  hipLaunchKernelGGL((cg_update_fused_kernel<T, EPTV, NTV>), dim3((unsigned)Bt), dim3(NTV), 0, s, ctrl, V, r, p,    \
                     ap_src, rz, over, err_out, n, (T)thr, (T)min_float, dinv, (int)max_it, ap_slices, ap_stride, agree, \
                     world)
        switch (fused_ept) {
          case 1: MGP_FUSED(1, 256); break;
          case 2: MGP_FUSED(2, 256); break;
          case 4: MGP_FUSED(4, 256); break;
          case 14: MGP_FUSED(1, 1024); break;
          case 12: MGP_FUSED(2, 1024); break;
          case 24: MGP_FUSED(4, 1024); break;
          default: MGP_FUSED(8, 1024); break;
        }
#undef MGP_FUSED
        MGP_LAUNCH_CHECK(h);
        continue;  // bookkeeping done inside the kernel
      } else if (!reset) {
        if (!dense_pre) {
          MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz,
                             over, err_out, n, (T)thr, (T)min_float, pc, 0, 0);
        } else {
          MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz,
                             over, err_out, n, (T)thr, (T)min_float, pc, 3, 0);
          MGP_LAUNCH_CHECK(h);
          MGP_TRY(external_z(&ctrl->active));
          MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz,
                             over, err_out, n, (T)thr, (T)min_float, pc, 4, 0);
        }
      } else {
        MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz,
                           over, err_out, n, (T)thr, (T)min_float, pc, 1, 0);
        MGP_LAUNCH_CHECK(h);
        MGP_TRY(apply_operator<T>(h, op, V, Bt, ap, &ctrl->active));
        hipLaunchKernelGGL((cg_residual_kernel<T>), dim3(nblk(tot)), dim3(256), 0, s, ctrl, B, ap, r, tot, 0);
        MGP_LAUNCH_CHECK(h);
        if (dense_pre) MGP_TRY(external_z(&ctrl->active));
        MGP_UPDATE_LAUNCH(ctrl, V, r, p, z, ap, rz,
                           over, err_out, n, (T)thr, (T)min_float, pc, dense_pre ? 5 : 2, 0);
      }
      MGP_LAUNCH_CHECK(h);
      hipLaunchKernelGGL(cg_advance_kernel, dim3(1), dim3(256), 0, s, ctrl, over, Bt, 1, (int)max_it);
      MGP_LAUNCH_CHECK(h);
    }
  }
  if (stats) {
    stats->iterations = host.iters;
    // converged iff the loop ended because no RHS was over threshold
    int any = 0;
    {
      // `active` == 0 with iters < max_it means converged; at the cap, look at the flags
      if (host.iters < max_it) {
        any = 0;
      } else {
        std::string tmp;
        tmp.resize((size_t)Bt * sizeof(int));
        MGP_HIP(h, hipMemcpyAsync(&tmp[0], over, (size_t)Bt * sizeof(int), hipMemcpyDeviceToHost, s));
        MGP_HIP(h, hipStreamSynchronize(s));
        const int* f = (const int*)tmp.data();
        for (long b = 0; b < Bt; ++b) any |= f[b];
      }
    }
    stats->converged = any ? 0 : 1;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
#undef MGP_UPDATE_LAUNCH
  return MGP_OK;
}

}  // namespace

extern "C" int mgp_pcg_solve(mgp_handle* h, const mgp_operator* op, const mgp_precond* pre, const void* B,
                             const void* V0, int64_t Bt, double error_threshold, int64_t max_iterations,
                             int64_t max_steps_cycle, double min_float, int32_t check_every, void* V_out,
                             void* err_out, mgp_cg_stats* stats) {
  if (!h) return MGP_E_BADARG;
  MGP_TRY(check_operator(h, op));
  if (Bt < 0) return mgp_fail(h, MGP_E_SHAPE, "Bt < 0");
  if (stats) {
    stats->iterations = 0;
    stats->converged = 1;
    stats->seconds = 0.0;
  }
  if (Bt == 0) return MGP_OK;
  if (!B || !V_out || !err_out) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (max_iterations < 0) return mgp_fail(h, MGP_E_BADARG, "max_iterations < 0");
  if (max_steps_cycle < 1) return mgp_fail(h, MGP_E_BADARG, "max_steps_cycle < 1");
  if (Bt > 2147483647L) return mgp_fail(h, MGP_E_SHAPE, "Bt too large");
  auto run = [&]() -> int {
    if (op->dtype == MGP_F64)
      return pcg_solve_t<double>(h, op, pre, (const double*)B, (const double*)V0, Bt, error_threshold, max_iterations,
                                 max_steps_cycle, min_float, check_every, (double*)V_out, (double*)err_out, stats);
    return pcg_solve_t<float>(h, op, pre, (const float*)B, (const float*)V0, Bt, error_threshold, max_iterations,
                              max_steps_cycle, min_float, check_every, (float*)V_out, (float*)err_out, stats);
  };
  int rc = run();
  if (rc == kRetryWithoutPersist) {
    if (V0 != nullptr && V0 == V_out)
      return mgp_fail(h, MGP_E_HIP, "dense CG: a hand-off between resident workgroups timed out (is the GPU shared?) and "
                                    "the initial solution was overwritten in place; set MGP_CG_DENSE1=1");
    fprintf(stderr, "libmgp: dense CG: a hand-off between resident workgroups timed out (is the GPU shared?); this solve "
                    "runs again with two launches per iteration\n");
    h->d1_persist_off = true;  // the solve again, from its inputs, two launches per iteration
    rc = run();
    h->d1_persist_off = false;
  }
  return rc;
}

extern "C" int mgp_operator_apply(mgp_handle* h, const mgp_operator* op, const void* P, int64_t Bt, void* out) {
  if (!h) return MGP_E_BADARG;
  MGP_TRY(check_operator(h, op));
  if (Bt <= 0) return Bt == 0 ? MGP_OK : mgp_fail(h, MGP_E_SHAPE, "Bt < 0");
  if (!P || !out) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  if (op->dtype == MGP_F64) return apply_operator<double>(h, op, (const double*)P, Bt, (double*)out, nullptr);
  return apply_operator<float>(h, op, (const float*)P, Bt, (float*)out, nullptr);
}

// Named form of the matrix-free (Kmm + Lambda) product (SURVEY 8b lists it as its own entry point):
// out[R, M] = V[R, M] @ (k(Z, Z) + diag(lambda)), rows = right-hand sides, nothing M x M materialised.
extern "C" int mgp_kmm_lambda_matvec(mgp_handle* h, const mgp_kernel* k, const void* Z, int64_t M,
                                     const void* lambda, const void* V, int64_t R, void* out) {
  if (!h) return MGP_E_BADARG;
  MGP_TRY(mgp_check_kernel(h, k));
  mgp_operator op;
  memset(&op, 0, sizeof(op));
  op.kind = MGP_OP_KMM_LAMBDA;
  op.dtype = k->dtype;
  op.n = M;
  op.kernel = k;
  op.Z = Z;
  op.M = M;
  op.lambda = lambda;
  return mgp_operator_apply(h, &op, V, R, out);
}
