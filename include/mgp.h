/*
 * mgp.h -- C ABI of libmgp (MI355X / gfx950 HIP kernels for the CG + kernel-matvec path).
 *
 * Drop-in boundary of SURVEY.md §8(b).  The reference (awav/conjugate-gradient-sparse-gp)
 * has no FFI of its own: its boundary is the Python call surface of cggp/conjugate_gradient.py,
 * cggp/models.py and cggp/distance.py.  Each entry point below names the reference
 * interface (file:line under /root/reference) whose arithmetic it replaces; the Python
 * host side (conjugate-gradient-sparse-gp_amd/cggp) binds these with ctypes and mirrors the
 * reference's names and argument meaning on top.
 *
 * Conventions
 *  - plain pointers and sizes only; every data pointer is a DEVICE pointer unless a
 *    parameter says "host"; row-major, contiguous, 16-byte aligned base.
 *  - the caller owns every buffer it passes.  The library owns only its handle and a
 *    device workspace that it grows on demand (never while a stream capture is active).
 *  - every function returns 0 on success or a negative MGP_E_* code and never throws or
 *    exits; mgp_last_error() gives the message.  CG non-convergence is NOT an error (the
 *    reference returns the last iterate silently, conjugate_gradient.py:93-98); it is
 *    reported through mgp_cg_stats.
 *  - all work is enqueued on the handle's stream (mgp_set_stream); the only host
 *    synchronisations are the convergence poll of mgp_pcg_solve and explicit stats readback.
 *  - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef MGP_H
#define MGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGP_VERSION 210 /* 0.2.1: + mgp_profile_read_clocks (0.2.0: mgp_comm, mgp_operator.comm, mgp_create_ex) */
#define MGP_MAX_D 512      /* input dimensions the library accepts (capacity of mgp_kernel) */
#define MGP_FUSED_MAX_D 32 /* up to here the fused register-resident kernels run; above, every entry point takes
                            * the generic route (csrc/generic.hip): products through row-chunked explicit kernel
                            * panels + the NT GEMM (the reference's dense form), nearest-centre, the kernel VJP
                            * and the k^2 column sum tile by tile with the dimensions staged through LDS */

enum { MGP_OK = 0, MGP_E_BADARG = -1, MGP_E_SHAPE = -2, MGP_E_DTYPE = -3, MGP_E_HIP = -4,
       MGP_E_COMM = -5, MGP_E_NOMEM = -6 };

enum { MGP_F32 = 0, MGP_F64 = 1 };

/* GPflow stationary kernels reachable in the reference (cggp/cli_utils.py:105-108,455-473;
 * cg_test.py:14): K2 of SURVEY §8a. */
enum { MGP_SE = 0, MGP_MATERN12 = 1, MGP_MATERN32 = 2, MGP_MATERN52 = 3 };

/* layout of a batch of vectors: MGP_COLS = [n, R] (the facade layout of
 * ConjugateGradient.__call__, conjugate_gradient.py:180-212), MGP_ROWS = [R, n] (the
 * function-level layout of conjugate_gradient(), conjugate_gradient.py:24-32). */
enum { MGP_COLS = 0, MGP_ROWS = 1 };

enum { MGP_PRE_EYE = 0, MGP_PRE_JACOBI = 1, MGP_PRE_BLOCK = 2, MGP_PRE_DENSE = 3, MGP_PRE_CALLBACK = 4 };

enum { MGP_OP_DENSE = 0, MGP_OP_SGPR = 1, MGP_OP_KMM_LAMBDA = 2 };

typedef struct mgp_handle mgp_handle;
/* one RCCL communicator rank (wraps ncclComm_t); see "collectives" below */
typedef struct mgp_comm mgp_comm;

/* Kernel hyper-parameters (host memory).  lengthscales has D entries (ARD; repeat the
 * value for an isotropic kernel).  Replaces gpflow.kernels.* parameter objects. */
typedef struct {
  int32_t kind;  /* MGP_SE ... */
  int32_t dtype; /* MGP_F32 | MGP_F64: element type of every data pointer of the call */
  int32_t D;
  int32_t reserved;
  double variance;
  double lengthscales[MGP_MAX_D];
} mgp_kernel;

/* Optional collective hook (rehearsal of the N > 1 path without RCCL, e.g. gloo with host staging):
 * sum `count` elements of `buf` (device) in place over all ranks, enqueued on `stream`.  Used only
 * for the [Bt*M + 1] partial product of the row-sharded SGPR operator -- one call per
 * operator application.  The native path is `mgp_operator.comm` (RCCL, no callback). */
typedef int (*mgp_allreduce_fn)(void* ctx, void* buf, size_t count, int dtype, void* stream);

/* Linear operator handed to mgp_pcg_solve (the `matrix` argument of
 * conjugate_gradient(), conjugate_gradient.py:24-27, generalised to matrix-free forms). */
typedef struct {
  int32_t kind;  /* MGP_OP_* */
  int32_t dtype;
  int64_t n;     /* operator is n x n */
  /* MGP_OP_DENSE: explicit symmetric matrix A [n,n] */
  const void* A;
  /* MGP_OP_SGPR: S = s2 * Kmm_j + K_mn K_nm with K_nm matrix-free over the local row shard.
   * Kmm_j [M,M] dense = k(Z,Z) + jitter I (replicated).  n = M. */
  const mgp_kernel* kernel;
  const void* X; int64_t N;
  const void* Z; int64_t M;
  const void* Kmm;
  double s2;
  /* MGP_OP_KMM_LAMBDA: (k(Z,Z) + diag(lambda)) applied matrix-free; lambda [M] device */
  const void* lambda;
  mgp_allreduce_fn allreduce; void* allreduce_ctx;
  /* MGP_OP_SGPR with the `allreduce` hook, optional: caller-owned device buffer of Bt_max*M + 1
   * elements that receives the local partial K_mn(K_nm p) and the agreement word (below) before
   * the collective, so a host-side collective can address it as its own tensor; NULL = library
   * scratch. */
  void* partial_buf;
  /* MGP_OP_SGPR, multi-rank: rows [kmm_row_begin, kmm_row_end) of Kmm whose s2*Kmm.p contribution
   * THIS rank adds to its partial (the slabs of all ranks must tile [0,M)); 0,0 = all rows. */
  int64_t kmm_row_begin, kmm_row_end;
  /* MGP_OP_SGPR, multi-rank, native: RCCL communicator of this rank.  With `allreduce == NULL` and
   * `comm != NULL` every operator application issues ONE ncclAllReduce(sum) of Bt*M + 1 elements on
   * the handle's stream -- the [Bt,M] partial plus one agreement word: inside mgp_pcg_solve each
   * rank contributes its `active` flag there, and an iteration is carried out only if the sum says
   * every rank is active, so all ranks leave the loop on the same iteration by construction. */
  mgp_comm* comm;
  /* number of ranks the `allreduce` hook sums over (the agreement word is compared with it);
   * ignored with `comm`, which knows its own size */
  int32_t world_size; int32_t reserved;
} mgp_operator;

/* MGP_PRE_CALLBACK: the caller's own preconditioner, the protocol of CGPreconditioner.__call__(vec, mat)
 * (conjugate_gradient.py:125-128, used at :77,:89).  Once per CG step, between the two halves of the
 * update, the library copies the residual batch r [Bt, n] into `cb_r` and calls `apply`, which must
 * ENQUEUE on `stream` work that leaves z = M^-1 r in `cb_z` (both buffers caller-owned device memory);
 * rz = sum(z * r) is formed by the library (the reference recomputes it the same way).  The loop stays
 * device resident: steps enqueued past convergence are gated off and their z is ignored.  Non-zero
 * return aborts the solve with MGP_E_BADARG. */
typedef int (*mgp_precond_fn)(void* ctx, const void* r, void* z, int64_t Bt, int64_t n, void* stream);

typedef struct {
  int32_t kind;               /* MGP_PRE_* */
  int32_t block_size;         /* MGP_PRE_BLOCK: bs */
  int64_t num_blocks;         /* MGP_PRE_BLOCK: nb */
  const void* diag_inv;       /* MGP_PRE_JACOBI: 1/diag(A) [n] device */
  const int64_t* block_index; /* MGP_PRE_BLOCK: [nb, bs] int64 device */
  const void* block_inv;      /* MGP_PRE_BLOCK: inverse of A[idx,idx], [nb, bs, bs] device */
  const void* dense_inv;      /* MGP_PRE_DENSE: symmetric P^-1 [n, n] device; z = r @ P^-1 */
  mgp_precond_fn apply;       /* MGP_PRE_CALLBACK */
  void* apply_ctx;
  void* cb_r;                 /* MGP_PRE_CALLBACK: [Bt, n] device, receives r before every call */
  void* cb_z;                 /* MGP_PRE_CALLBACK: [Bt, n] device, z left there by `apply` */
} mgp_precond;

typedef struct {
  int32_t iterations; /* CG steps taken == reference stats_steps (conjugate_gradient.py:96) */
  int32_t converged;  /* 1 iff all_b(0.5*||r_b||^2 <= thr) at exit */
  double seconds;     /* wall time of the solve (host clock around the enqueue + poll) */
} mgp_cg_stats;

/* ---- handle ------------------------------------------------------------------------- */
int mgp_version(void);
int mgp_create(mgp_handle** out, int device);
/* As mgp_create, with the device workspace fixed up front: one allocation of `workspace_bytes`
 * serves every internal arena (sweep partials, CG state, operator scratch) and the library never
 * calls hipMalloc/hipFree afterwards -- a request that does not fit fails with MGP_E_NOMEM
 * (mgp_last_error names the shortfall).  workspace_bytes == 0 behaves as mgp_create (arenas grow
 * on demand).  mgp_workspace_bytes reports what a handle holds now: run the workload once on a
 * growing handle, read it, and create the production handle with that figure. */
int mgp_create_ex(mgp_handle** out, int device, size_t workspace_bytes);
size_t mgp_workspace_bytes(const mgp_handle* h);
int mgp_destroy(mgp_handle* h);
int mgp_set_stream(mgp_handle* h, void* hip_stream);
const char* mgp_last_error(mgp_handle* h);
/* name of the gfx arch the device code was built for, e.g. "gfx950" */
const char* mgp_build_arch(void);

/* ---- matrix-free kernel products --------------------------
 * out[N,R] = k(X,Z) V      replaces Kuf(...)^T @ a, cggp/models.py:334,351 (and :273,:157)
 * V, out layouts given by v_layout / out_layout (MGP_COLS: [M,R]/[N,R]; MGP_ROWS: [R,M]/[R,N]).
 * An empty contraction set (M = 0 here, N = 0 in mgp_kmn_matvec: no inducing points / a rank without rows) is not an
 * error: the output is written as zeros, what the reference's dense [B,0].[0,R] product gives. */
int mgp_knm_matvec(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                   int64_t M, const void* V, int32_t R, int v_layout, void* out, int out_layout);
/* out[M,R] = k(Z,X) W = K_nm^T W   (W [N,R]); deterministic two-stage reduction over row
 * blocks (no float atomics).  The transpose product of models.py:343 / the SGPR A A^T term. */
int mgp_kmn_matvec(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                   int64_t M, const void* W, int32_t R, int w_layout, void* out, int out_layout);
/* out[na, ld>=nb] = k(A,B) (+ jitter on the diagonal, + diag_add[i] on the diagonal when
 * non-NULL).  Replaces gpflow Kuu/Kuf + add_diagonal: models.py:300-301,333-337, utils.py:11-17. */
int mgp_k_dense(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B,
                int64_t nb, void* out, int64_t ld, double jitter, const void* diag_add);
/* out[M,M] = K_mn K_nm = k(Z,X) k(X,Z): fp64 MFMA contraction, K row panels generated on the
 * fly (row S1: GPflow SGPR's A A^T, 2 N M^2 flop). */
int mgp_kmn_knm(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                int64_t M, void* out);

/* out[M] = sum_i k(x_i, z_m)^2 = diag(K_mn K_nm): the N-sized part of diag(S) for the Jacobi
 * preconditioner of the SGPR normal equations (build-side addition, no reference counterpart). */
int mgp_kmn_sq_colsum(mgp_handle* h, const mgp_kernel* k, const void* X, int64_t N, const void* Z,
                      int64_t M, void* out);

/* ---- dense symmetric product (row M2: `state.p @ A`, conjugate_gradient.py:65) ---------
 * out[Bt,n] = P[Bt,n] @ A[n,n] for SYMMETRIC A (CG requires it; computed as rows of A dotted
 * with p_b, i.e. P @ A^T).  For Bt = 1 and n >= 1024 only the upper triangle of A (in 64x64
 * tiles, diagonal tiles whole) is read: each tile serves both A_IJ p_J and A_IJ^T p_I. */
int mgp_symm_matmul(mgp_handle* h, int dtype, const void* A, int64_t n, const void* P,
                    int64_t Bt, void* out);

/* ---- preconditioned batched CG (rows CG1, CG3-CG5) -------------------------------------
 * Solves V A = B for row batches B [Bt,n] starting from V0 (NULL = zeros), exactly the
 * recurrence of conjugate_gradient.py:59-98: stop when all_b(0.5||r_b||^2 <= thr) or
 * i >= max_iterations; gamma = 0 where p.Ap <= min_float; beta-term = 0 where rz <= min_float;
 * residual refresh + direction restart when i % max_steps_cycle == max_steps_cycle-1.
 * err_out [Bt] receives 0.5*rz_final (stats_error, :97).  check_every = iterations enqueued
 * between convergence polls (device-side gating keeps the step count exact). */
int mgp_pcg_solve(mgp_handle* h, const mgp_operator* op, const mgp_precond* pre, const void* B,
                  const void* V0, int64_t Bt, double error_threshold, int64_t max_iterations,
                  int64_t max_steps_cycle, double min_float, int32_t check_every, void* V_out,
                  void* err_out, mgp_cg_stats* stats);
/* one application of an operator: out[Bt,n] = P[Bt,n] @ Op (used by tests and the bench) */
int mgp_operator_apply(mgp_handle* h, const mgp_operator* op, const void* P, int64_t Bt, void* out);
/* the matrix-free (Kmm + Lambda) product by name (row M2, `p @ A` with A = add_diagonal(Kuu, lambda),
 * models.py:301,337): out[R,M] = V[R,M] @ (k(Z,Z) + diag(lambda)); lambda [M] on the device */
int mgp_kmm_lambda_matvec(mgp_handle* h, const mgp_kernel* k, const void* Z, int64_t M, const void* lambda,
                          const void* V, int64_t R, void* out);

/* ---- reductions used by the model surface ----------------------------------------------
 * out[c] = sum_r A[r,c]*B[r,c] (axis 0, like tf.reduce_sum(Kmn * W, axis=0), models.py:343) */
int mgp_colwise_dot(mgp_handle* h, int dtype, const void* A, const void* B, int64_t rows,
                    int64_t cols, void* out);
/* *out = sum(A*B) over count elements (host double); Hutchinson trace, models.py:313 */
int mgp_dot_all(mgp_handle* h, int dtype, const void* A, const void* B, int64_t count, double* out);

/* ---- collectives: RCCL over xGMI, one communicator rank per GPU ----------------
 * The reference has no collective; these are the build's own multi-GPU boundary.
 * Two ways to form the communicator:
 *  - one process per GPU (torch.distributed.run): rank 0 calls mgp_comm_unique_id, ships the
 *    MGP_COMM_ID_BYTES to the other ranks by any channel, every rank calls mgp_comm_init_rank;
 *  - one process driving ndev devices: mgp_comm_init_all(ndev, devs, comms) (ncclCommInitAll).
 * mgp_allreduce_sum: in-place sum of `count` elements of `buf` (device) over the ranks, enqueued on
 * `stream`; calls for several communicators of one process must be bracketed by
 * mgp_comm_group_begin/end.  Errors: negative code, text via mgp_comm_last_error (thread local). */
#define MGP_COMM_ID_BYTES 128
int mgp_comm_unique_id(void* id_out);
int mgp_comm_init_rank(mgp_comm** out, int device, int nranks, int rank, const void* id);
int mgp_comm_init_all(int ndev, const int* devs, mgp_comm** comms);
int mgp_comm_destroy(mgp_comm* c);
int mgp_comm_size(const mgp_comm* c);
int mgp_comm_rank(const mgp_comm* c);
int mgp_comm_group_begin(void);
int mgp_comm_group_end(void);
int mgp_allreduce_sum(void* buf, size_t count, int dtype, mgp_comm* comm, void* stream);
const char* mgp_comm_last_error(void);

/* ---- next row F1: nearest-centre assignment + cluster statistics (optimize.py:41-98) ----
 * idx[i] = argmin_m d(Z_m, X_i) (first index on ties), dist type 0 = squared euclidean on raw
 * inputs (ops.square_distance, optimize.py:50), 1 = euclidean (distance.py:9-11, same argmin),
 * 2 = covariance, 3 = correlation (distance.py:15-30; kernel required).  best[i] = the distance.
 * sums[M], counts[M] accumulate y and 1 per cluster in deterministic order. */
int mgp_nearest_center(mgp_handle* h, const mgp_kernel* k, int dist_type, const void* X, int64_t N,
                       const void* Z, int64_t M, int64_t* idx, void* best);
int mgp_cluster_stats(mgp_handle* h, int dtype, const int64_t* idx, const void* y, int64_t N,
                      int64_t M, void* sums, void* counts);
/* the same sums for C columns at once when the caller has grouped the rows: order [N] = row indices
 * sorted (stably) by cluster, offsets [M+1] = start of every cluster's run in `order`;
 * sums[M,C] = per-cluster column sums of Y [N,C], summed in a fixed order (N C work, against the
 * N M of the sweep above: the k-means centroid update, selection.py:58-63, and optimize.py:61-78) */
int mgp_segment_sums(mgp_handle* h, int dtype, const int64_t* order, const int64_t* offsets, const void* Y,
                     int64_t N, int64_t C, int64_t M, void* sums);

/* ---- next row F2: hyper-parameter gradient of a kernel block ---------------------------------
 * Given G = dL/dK for K = k(A,B) [na, nb] (leading dimension ldg), returns (host doubles)
 * dL/dvariance and dL/dlengthscales[D] -- the vector-Jacobian product autodiff takes through
 * gpflow's kernel in the reference's training step (cggp/optimize.py:198-254 over
 * cggp/models.py:125-134,293-354).  Fused reduction; synchronises the stream. */
int mgp_k_dense_vjp(mgp_handle* h, const mgp_kernel* k, const void* A, int64_t na, const void* B,
                    int64_t nb, const void* G, int64_t ldg, double* dvariance, double* dlengthscales);

/* ---- measurement (bench.py): HIP events around every launch of the fused sweep kernel ------
 * While enabled, each sweep launch is bracketed by two events on the handle's stream;
 * mgp_profile_read synchronises the stream, returns the number of bracketed launches and the
 * sum of their durations in milliseconds, and resets the counters.  Enabling reserves a 4 MB block for
 * the clock stamps below as one more arena of the workspace: counted by mgp_workspace_bytes, taken from
 * the fixed pool of mgp_create_ex (MGP_E_NOMEM if it does not fit) -- no allocation outside it. */
int mgp_profile_enable(mgp_handle* h, int on);
int mgp_profile_read(mgp_handle* h, int64_t* launches, double* total_ms);
/* as mgp_profile_read, but one duration per bracketed launch (the first `capacity` of them) so the
 * caller can report median / percentiles; *launches receives the number recorded. */
int mgp_profile_read_each(mgp_handle* h, double* ms_out, int64_t capacity, int64_t* launches);
/* Shader clock the bracketed sweep launches ran at (bench-only): while profiling is on, up to 16 workgroups of each
 * launch stamp the constant 100 MHz counter and the shader-clock counter at their start and at the end of their loop;
 * this returns one value in MHz per stamped workgroup (the first `capacity` of them) and their number. */
int mgp_profile_read_clocks(mgp_handle* h, double* mhz_out, int64_t capacity, int64_t* samples);

/* ---- next row F3: cover-tree clustering (host code, no GPU, no handle) ----------------------
 * Replaces the reference's CoverTree class (cggp/covertree.py:26-179), which
 * covertree_update_inducing_parameters (cggp/optimize.py:19-38) turns into inducing inputs,
 * cluster means and counts.  x is a HOST pointer [N, D] row-major fp64.  spatial_resolution > 0
 * derives the level count as the reference does (:54-56), otherwise num_levels is used.  The tree
 * keeps row indices into x (x is only read during the build).  Levels are read back flat:
 * level_nodes gives centres [n, D], the parent's position in the level above (-1 for the root) and
 * the number of rows held; level_rows gives CSR offsets [n+1] and the concatenated row indices
 * (rows may be NULL to get the offsets only).  Errors: mgp_host_last_error() (thread local). */
typedef struct mgp_covertree mgp_covertree;
const char* mgp_host_last_error(void);
int mgp_covertree_build(const double* x, int64_t N, int D, double spatial_resolution, int num_levels, int lloyds,
                        int voronoi, mgp_covertree** out);
/* The same construction -- the same tree, node for node and bit for bit -- with its four all-pairs-shaped passes (the
 * ball of a seed, the rows a new centre takes, the per-level Voronoi reassignment, the r-neighbour test of the new
 * centres) as device filters over `x_dev`, the fp64 device copy of the same [N, D] rows; the sequential acceptance of
 * centres and every mean stay on the host (csrc/covertree_dev.hip).  For inputs where almost every node is an
 * r-neighbour of every other (D >= ~6) the host version scans nearly all rows per centre. */
int mgp_covertree_build_device(mgp_handle* h, const double* x, const double* x_dev, int64_t N, int D,
                               double spatial_resolution, int num_levels, int lloyds, int voronoi, mgp_covertree** out);
void mgp_covertree_destroy(mgp_covertree* t);
int mgp_covertree_num_levels(const mgp_covertree* t);
int64_t mgp_covertree_level_size(const mgp_covertree* t, int level);
double mgp_covertree_level_radius(const mgp_covertree* t, int level);
int mgp_covertree_level_nodes(const mgp_covertree* t, int level, double* points, int64_t* parent, int64_t* counts);
int mgp_covertree_level_rows(const mgp_covertree* t, int level, int64_t* offsets, int64_t* rows);

#ifdef __cplusplus
}
#endif
#endif /* MGP_H */
