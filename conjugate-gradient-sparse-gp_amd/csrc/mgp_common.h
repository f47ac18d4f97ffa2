// mgp_common.h -- handle, workspace arena, error plumbing shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mgp.h"
#include "mgp_math.h"

struct mgp_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  // device workspace (partials of the two-stage reductions, CG state); grown on demand
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // CG state arena, separate from ws so operator applications inside the solve can use ws
  void* cg = nullptr;
  size_t cg_bytes = 0;
  // operator scratch (u = K_nm p and the [Bt,M] partial of the SGPR operator)
  void* opws = nullptr;
  size_t opws_bytes = 0;
  // generic-D scratch (transposed multipliers, kernel panel, chunk output)
  void* gen = nullptr;
  size_t gen_bytes = 0;
  // packed streamed points of the fp64 SE fast sweep (sweep.hip).  Two slots: a solve alternates the K_nm
  // direction (streamed set Z) and the K_mn direction (streamed set X).  A pack is reused only while
  // `pack_hold` is set -- inside mgp_pcg_solve, where the operator's X, Z and kernel cannot change.
  struct PackSlot {
    void* buf = nullptr;
    size_t bytes = 0;
    const void* src = nullptr;
    long n = 0;
    int D = 0;
    double inv_ls[MGP_FUSED_MAX_D] = {0};
    bool valid = false;
  } pack[2];
  int pack_next = 0;
  bool pack_hold = false;
  // mgp_create_ex: one caller-sized block that serves every arena above; no hipMalloc/hipFree after create
  void* pool = nullptr;
  size_t pool_bytes = 0, pool_used = 0;
  // pinned host word for the convergence poll
  int* host_flag = nullptr;
  hipEvent_t poll_ev[2] = {nullptr, nullptr};  // one polled batch of the dense one-RHS CG in flight (cg.hip)
  // SGPR operator, one right-hand side: the s2*Kmm.p slab product runs on a stream of its own beside the K_nm sweep
  // (it reads Kmm from HBM while the sweep is bound by vector-ALU issue) and enters the K_mn sweep as its addend
  // (MGP_SGPR_KMM_ASIDE=0: on the solve's stream after both sweeps, as rounds 1-3; 1: for slabs of 48 MB and more; 2: always)
  int kmm_aside = 1;
  hipStream_t aside_stream = nullptr;
  hipEvent_t aside_ev[2] = {nullptr, nullptr};  // fork (p is ready), join (Kmm.p is ready)
  // most right-hand sides the tile scheme takes (MGP_CG_DENSE1_COLS: 1 = one only, as round 3; up to 8); 0 = by size,
  // where it was measured faster than the skinny product + fused update: 4 for n <= 4096, 6 above
  int cg_dense1_cols = 0;
  int d1_first_poll_sleep = 16;  // register-resident dense CG, owners: x 64 cycles before the first poll of the slots (MGP_D1_FIRST_POLL)
  int d1_inject_absent = -1;  // test only (MGP_D1_INJECT_ABSENT=<workgroup>): that workgroup of the register-resident solve leaves at once
  int d1_owner_spread = 1;  // super-block form, several columns: the columns of a chunk owned by different workgroups (MGP_D1_OWNER_SPREAD=0: by one)
  bool d1_persist_off = false;  // set for the retry of a solve whose register-resident launch reported a timed-out hand-off
  int poll_pipeline = 1;  // MGP_CG_PIPELINE_POLLS=0: drain the stream at every poll (round 3)
  void* ones = nullptr;  // device constants: double 1.0 at +0, float 1.0f at +8
  void* e2tabs = nullptr;  // exp2 tables of the fast fp64 sweep (8192 + 2048 entries, sweep.hip: mgp_build_e2tabs)
  double* dparams = nullptr;  // device copy of c/lengthscale_d for the generic-D kernels [MGP_MAX_D]
  int num_cus = 256;
  // 0 = fused sweeps on the VALU (sweep.hip, default: measured faster), 1 = fp64 distance
  // cross-term on the matrix cores (sweep_mfma.hip); MGP_SWEEP=mfma selects 1 for A/B runs
  int sweep_mode = 0;
  // fp64 sweeps: sweep_fast_kernel (scalar-loaded packed points, integer exponent scaling) with 1 = 256 threads /
  // 2048-entry table (one right-hand side only), 2 = 512 threads / 8192-entry table and a one-instruction table
  // offset at D <= 16; 0 = the LDS-tile kernel it replaced (MGP_SWEEP_FAST, for A/B runs)
  int sweep_fast = 2;
  int pf_trips = 16, pf_ahead = 6144;  // L2 prefetch of streamed rows: every pf_trips loop trips (power of two), pf_ahead bytes on
  // chunking of the streamed set aims at this many 256-thread workgroups per CU (MGP_SWEEP_TARGET).  16 since the
  // per-workgroup timeline of round 2: the two workgroups of a CU do not share it, the older one runs at full speed
  // and the younger fills its stalls, so each CU ends with one workgroup running alone at ~0.82 of the rate of two --
  // the shorter the workgroups, the shorter that tail (C3: 4.75 -> 4.66 ms per CG step against 8)
  int sweep_target_per_cu = 16;
  int sweep_chunk_gran = 0;  // fast kernel: streamed points per chunk are a multiple of this (MGP_SWEEP_GRAN: 64, 128, 256); 0 = the tile size TB of the shape
  int sweep_fast_rpt32 = 2;  // the same for 16 < D <= 32: 2 (2 waves/SIMD) or 1 (4 waves/SIMD) -- MGP_SWEEP_RPT32
  int sweep_fast_rpt_rc = 2;  // owned points per lane with 2 or 4 right-hand sides at D <= 8 (2 or 3) -- MGP_SWEEP_RPT_RC
  int sweep_fast_rpt = 4;  // owned points per lane of the fast kernel: 4 (4 waves/SIMD), 3 (5), 2 (8) -- MGP_SWEEP_RPT
  // K^T panel size per launch of the two-stage contraction: small enough to stay in the 256 MiB
  // Infinity Cache between its write (k_dense) and its ~33 re-reads (MGP_CONTRACT_PANEL_MB)
  size_t contract_panel_mb = 2048;
  int contract_nz = 16;
  // one-RHS symmetric product on the upper triangle (dense.hip); sizes below tri_min_n use the
  // row-streaming GEMV (MGP_TRI_MIN_N)
  long tri_min_n = 1024;
  // form of that product's tile kernel: 1 = row sums by a cross-lane reduce-scatter (2 KB of LDS), 0 = round 1's
  // LDS-staged row sums (35 KB, four workgroups per CU) -- MGP_TRI_FORM
  int tri_form = 1;
  void* tri_tab = nullptr;  // (I, J) of the upper-triangle tiles in launch order, for tri_tab_nt tile rows
  size_t tri_tab_bytes = 0;
  int tri_tab_nt = 0;
  // one-right-hand-side dense CG (the reference's literal loop, conjugate_gradient.py:65-84): 3 = for n <= 4096 the
  // whole solve in ONE launch with the upper triangle of A held in registers (cg_dense1.hip, round 4) and the
  // two-launch iteration above that, 1 = two launches per iteration everywhere, 0 = product (tile kernel + slot
  // reduce) + fused update launch (MGP_CG_DENSE1)
  int cg_dense1 = 3;
  int kdense_ta = 0;  // rows of A per block of k_dense_kernel: 0 = by shape (16, or 64 at D > 8), else 16 or 64 (MGP_KDENSE_TA)
  int gemm_ksplit = 1;  // mid-size GEMMs: 128x128 tiles x K slices instead of 64x64 tiles (MGP_GEMM_KSPLIT=0 disables)
  int skinny_blocks_per_cu = 0;  // k slices of the skinny product: workgroups per CU to aim for; 0 = by panel width (MGP_SKINNY_BPC)
  int skinny_stagger = 0;  // experiment (MGP_SKINNY_STAGGER): start-up delay units between workgroup phases
  // Deferred slice sum of the skinny product (dense.hip -> cg.hip): while `defer_slices` is set the product leaves its
  // contraction slices in `ws` instead of launching skinny_reduce_kernel and reports them here; the fused CG update
  // adds them in slice order as it reads A.p (the same sums in the same order, one launch fewer per iteration).
  bool defer_slices = false;
  const void* deferred_part = nullptr;
  int deferred_ks = 1;
  long deferred_stride = 0;
  // Deferred agreement check of the multi-rank SGPR operator (cg.hip): while `defer_finish` is set the operator
  // leaves the all-reduced partial where the collective put it and reports it here; the fused CG update reads A.p
  // from there and does finish_allreduce_kernel's test itself (agreement word == ranks, else the gate closes)
  int fuse_agree = 1;  // MGP_FUSE_AGREE=0: put_gate_word_kernel + finish_allreduce_kernel as launches of their own
  bool defer_finish = false;
  const void* deferred_tt = nullptr;
  int deferred_world = 0;
  int skinny_defer = 1;  // MGP_SKINNY_DEFER=0 keeps the separate reduce launch inside the CG loop
  int skinny_pipe = 1;  // software-pipelined form of the LDS-staged product when n % 64 == 0 and Bt <= 64 (MGP_SKINNY_PIPE=0: the round-1 form)
  int skinny_mode = 1;  // 2 <= Bt <= 128 product: 1 = P staged through LDS, 0 = register operands (MGP_SKINNY=reg)
  int nosplit_per_cu = 4;  // owned-side workgroups per CU above which the streamed set is not split (MGP_NOSPLIT_PER_CU)
  // bench-only: event pairs around sweep launches (mgp_profile_enable / mgp_profile_read)
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  size_t prof_used = 0;
  // bench-only: clock stamps of the profiled sweep launches (mgp_profile_read_clocks): 16 slots of 4 words each
  void* prof_clk = nullptr;
  size_t prof_clk_bytes = 0;
  size_t prof_clk_launches = 0;
};

// scope guard of mgp_pcg_solve: packs made during the solve are reused by its later iterations
struct PackHold {
  mgp_handle* h;
  explicit PackHold(mgp_handle* hh) : h(hh) {
    h->pack[0].valid = h->pack[1].valid = false;
    h->pack_hold = true;
  }
  ~PackHold() {
    h->pack_hold = false;
    h->pack[0].valid = h->pack[1].valid = false;
  }
};

constexpr int MGP_PROF_CLK_WORDS = 64;        // per profiled launch: 16 slots x (real0, clk0, real1, clk1)
constexpr int MGP_PROF_CLK_LAUNCHES = 8192;  // launches with stamps between two mgp_profile_enable(1) calls

// the stamp slot block of the NEXT profiled launch (nullptr when profiling is off or the block is used up)
inline unsigned long long* mgp_prof_clk_next(mgp_handle* h) {
  if (!h->prof_on || !h->prof_clk || h->prof_clk_launches >= (size_t)MGP_PROF_CLK_LAUNCHES) return nullptr;
  return (unsigned long long*)h->prof_clk + (h->prof_clk_launches++) * MGP_PROF_CLK_WORDS;
}

// which = 0 at the start of a workgroup, 1 when its loop has ended: 16 evenly spaced workgroups of the launch stamp
// (constant 100 MHz counter, shader-clock counter).  One lane, two scalar reads, one 16-byte store; nothing is kept
// in registers in between.
__device__ __forceinline__ void mgp_prof_stamp(unsigned long long* clk, int which) {
  if (clk == nullptr) return;
  const unsigned stride = gridDim.x >= 16 ? gridDim.x >> 4 : 1;
  const unsigned slot = blockIdx.x / stride;
  if (blockIdx.x == slot * stride && slot < 16 && threadIdx.x == 0) {
    clk[4 * slot + 2 * which] = __builtin_amdgcn_s_memrealtime();
    clk[4 * slot + 2 * which + 1] = __builtin_amdgcn_s_memtime();
  }
}

// returns the stop event to record after the launch (nullptr when profiling is off)
inline hipEvent_t mgp_prof_begin(mgp_handle* h) {
  if (!h->prof_on) return nullptr;
  if (h->prof_used == h->prof_ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
    h->prof_ev.emplace_back(a, b);
  }
  auto& pr = h->prof_ev[h->prof_used++];
  (void)hipEventRecord(pr.first, h->stream);
  return pr.second;
}
inline void mgp_prof_end(mgp_handle* h, hipEvent_t stop) {
  if (stop) (void)hipEventRecord(stop, h->stream);
}

inline int mgp_fail(mgp_handle* h, int code, const char* fmt, ...) {
  if (h) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(h->err, sizeof(h->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define MGP_HIP(h, call)                                                                  \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess)                                                                \
      return mgp_fail((h), MGP_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                                \
  } while (0)

#define MGP_TRY(call)          \
  do {                         \
    int rc__ = (call);         \
    if (rc__ != MGP_OK) return rc__; \
  } while (0)

#define MGP_LAUNCH_CHECK(h) MGP_HIP((h), hipGetLastError())

// grow-only arenas; never called during stream capture (callers size up front).  With a fixed
// workspace (mgp_create_ex) an arena that must grow takes a fresh region of the pool (its old one is
// not reused) and the request fails when the pool is exhausted -- the library never allocates then.
inline int mgp_reserve(mgp_handle* h, void** p, size_t* have, size_t need) {
  if (need <= *have) return MGP_OK;
  if (h->pool) {
    const size_t at = (h->pool_used + 255) & ~(size_t)255;
    if (at + need > h->pool_bytes)
      return mgp_fail(h, MGP_E_NOMEM, "fixed workspace exhausted: %zu bytes requested, %zu of %zu in use "
                      "(size it with mgp_workspace_bytes on a growing handle)", need, h->pool_used, h->pool_bytes);
    if (*p) MGP_HIP(h, hipStreamSynchronize(h->stream));  // the old region may still be read by queued work
    *p = (char*)h->pool + at;
    *have = need;
    h->pool_used = at + need;
    return MGP_OK;
  }
  if (*p) {
    MGP_HIP(h, hipStreamSynchronize(h->stream));
    MGP_HIP(h, hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  size_t want = need + (need >> 2) + 4096;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) return mgp_fail(h, MGP_E_NOMEM, "workspace hipMalloc(%zu) failed: %s", want,
                                        hipGetErrorString(e));
  *have = want;
  return MGP_OK;
}

// value of `v` in lane `src` (wave-uniform) as a scalar operand: v_readlane, no LDS-pipe traffic
__device__ __forceinline__ double mgp_read_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float mgp_read_lane(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

int mgp_build_e2tabs(mgp_handle* h);  // sweep.hip
// dense.hip: slots Q[nt][n] (h->ws) and the (I, J) table of the one-RHS upper-triangle product, nt = ceil(n/64);
// Q[k][i] = contribution of chunk k to output element i, summed by the caller in k order
int mgp_symm_gemv_tri_prepare(mgp_handle* h, int dtype, int64_t n, void** Q, const void** tab);

// cg_dense1.hip: one right-hand side on a dense matrix, two launches per iteration
struct MgpCgCtrl {
  int active;
  int iters;
  unsigned ticket;  // arrivals of the fused update kernel's workgroups (reset by the last one)
  int pad;
};
struct MgpDense1 {
  int dtype = 0, nt = 0, max_it = 0;
  int64_t n = 0;
  long ntiles = 0;
  const void* A = nullptr;
  const void* dinv = nullptr;
  void *V = nullptr, *r = nullptr, *Q = nullptr, *tpart = nullptr, *cpart = nullptr, *scal = nullptr;
  void* pb[2] = {nullptr, nullptr};
  // register-resident form (cg_dense1.hip, round 4): published z, the workgroups' shares of p.Ap, the hand-off flags
  int persist = 0, bt = 1;  // bt: right-hand sides (2..8: the multi-column kernels, their slots in Qm)
  void* Qm = nullptr;
  void *zpub = nullptr, *sync = nullptr, *gran = nullptr;
  const void* tab = nullptr;
  MgpCgCtrl* ctrl = nullptr;
  double thr = 0, min_float = 0;
};
size_t mgp_dense1_bytes(const mgp_handle* h, int dtype, int64_t n, int64_t bt);
bool mgp_dense1_eligible(const mgp_handle* h, int64_t n);
bool mgp_dense1_persist_eligible(const mgp_handle* h, int64_t n, int64_t bt);
int mgp_dense1_begin(mgp_handle* h, MgpDense1* st, int dtype, const void* A, int64_t n, const void* B, const void* av,
                     void* V, void* r, const void* dinv, MgpCgCtrl* ctrl, void* arena, double thr, double min_float,
                     int64_t max_it, int persist, int bt);
int mgp_dense1_persist_run(mgp_handle* h, const MgpDense1* st);
int mgp_dense1_step(mgp_handle* h, const MgpDense1* st, int64_t k);
int mgp_dense1_finish(mgp_handle* h, MgpDense1* st, void* rz, void* err, int* over);

inline size_t mgp_elem(int dtype) { return dtype == MGP_F64 ? 8 : 4; }

inline int mgp_check_kernel(mgp_handle* h, const mgp_kernel* k) {
  if (!h) return MGP_E_BADARG;
  if (!k) return mgp_fail(h, MGP_E_BADARG, "kernel is NULL");
  if (k->kind < MGP_SE || k->kind > MGP_MATERN52) return mgp_fail(h, MGP_E_BADARG, "bad kernel kind %d", k->kind);
  if (k->dtype != MGP_F32 && k->dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", k->dtype);
  if (k->D < 1 || k->D > MGP_MAX_D) return mgp_fail(h, MGP_E_SHAPE, "D=%d outside [1,%d]", k->D, MGP_MAX_D);
  if (!(k->variance > 0.0)) return mgp_fail(h, MGP_E_BADARG, "variance must be > 0");
  for (int d = 0; d < k->D; ++d)
    if (!(k->lengthscales[d] > 0.0)) return mgp_fail(h, MGP_E_BADARG, "lengthscale[%d] must be > 0", d);
  return MGP_OK;
}

// Scaled kernel parameters passed by value to device code.
struct SweepParams {
  double inv_ls[MGP_FUSED_MAX_D];  // c_kind / lengthscale_d (fused kernels only: D <= 32)
  double variance;
  double clamp;  // c_kind^2 * 1e-36
};

inline SweepParams mgp_make_params(const mgp_kernel* k) {
  SweepParams p;
  const double c = mgp_profile_scale(k->kind);
  for (int d = 0; d < MGP_FUSED_MAX_D; ++d) p.inv_ls[d] = d < k->D ? c / k->lengthscales[d] : 0.0;
  p.variance = k->variance;
  p.clamp = c * c * 1e-36;
  return p;
}

// internal cross-TU entry points (layout-generic forms of the public calls)
struct VecView {  // element (i, r) of a batch of vectors lives at base[i*si + r*sr]
  const void* base;
  int64_t si, sr;
};
struct VecViewMut {
  void* base;
  int64_t si, sr;
};
inline VecView mgp_view(const void* p, int64_t n, int64_t R, int layout) {
  return layout == MGP_COLS ? VecView{p, R, 1} : VecView{p, 1, n};
}
inline VecViewMut mgp_view_mut(void* p, int64_t n, int64_t R, int layout) {
  return layout == MGP_COLS ? VecViewMut{p, R, 1} : VecViewMut{p, 1, n};
}

// out(i,r) = variance * sum_j k(a_i, b_j) w(j,r) [+ alpha * addend(i,r)];  gate: device int, skip if 0
int mgp_sweep(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
              VecView W, int32_t R, VecViewMut out, double alpha, VecView addend, const int* gate);
// generic-D forms (generic.hip): explicit panels + NT GEMM, any D <= MGP_MAX_D
int mgp_k_dense_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                        void* out, int64_t ld, double jitter, const void* diag_add, const int* gate);
int mgp_sweep_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                      VecView W, int32_t R, VecViewMut out, double alpha, VecView addend, const int* gate);
int mgp_nearest_generic(mgp_handle* h, const mgp_kernel* k, int dist_type, const void* X, int64_t N, const void* Z,
                        int64_t M, int64_t* idx, void* best);
int mgp_k_dense_vjp_generic(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B, int64_t nb,
                            const void* G, int64_t ldg, double* dvariance, double* dlengthscales);
int mgp_kmn_sq_colsum_generic(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z, int64_t M,
                              void* out);
int mgp_gemm_nt(mgp_handle* h, int dtype, const void* P, int64_t ldp, int64_t m, const void* A, int64_t lda, int64_t n,
                int64_t K, void* out, int64_t ldo, int accumulate, const int* gate);
int mgp_sweep_mfma_f64(mgp_handle* h, const mgp_kernel* k, const double* A, long na, const double* B, long nb,
                       const double* W, long w_sj, long w_sr, int R, double* out, long o_si, long o_sr,
                       double alpha, const double* addend, long ad_si, long ad_sr, const int* gate);
int mgp_syrk_nt_upper(mgp_handle* h, int dtype, const void* Kt, int64_t n, int64_t K, int64_t ld, void* out,
                      int accumulate, int nz, const int* tile_tab, int ntiles);
int mgp_mirror_upper(mgp_handle* h, int dtype, void* out, const void* slices, int nz, int64_t n, double scale);
int mgp_symm_gemv_assign(mgp_handle* h, int dtype, const void* A, int64_t n, const void* p, void* out, const int* gate);
int mgp_symm_gemv_rows_acc(mgp_handle* h, int dtype, const void* A, int64_t n, const void* p, int64_t rb, int64_t re,
                           double alpha, void* out, const int* gate, void* word = nullptr);
// comm.hip: the operator's all-reduce on the handle's stream (errors land in the handle)
int mgp_comm_allreduce_on(mgp_handle* h, mgp_comm* comm, void* buf, size_t count, int dtype);
int mgp_symm_matmul_gated(mgp_handle* h, int dtype, const void* A, int64_t n, const void* P, int64_t Bt,
                          void* out, const int* gate);
