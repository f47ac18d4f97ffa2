// api.hip -- handle lifecycle and the small reductions of the model surface.
#include <cstdlib>

#include "mgp_common.h"

extern "C" int mgp_version(void) { return MGP_VERSION; }

extern "C" const char* mgp_build_arch(void) { return "gfx950"; }

extern "C" int mgp_create(mgp_handle** out, int device) {
  if (!out) return MGP_E_BADARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MGP_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return MGP_E_HIP;
  mgp_handle* h = new (std::nothrow) mgp_handle();
  if (!h) return MGP_E_NOMEM;
  h->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cus = prop.multiProcessorCount;
  const char* mode = getenv("MGP_SWEEP");
  if (mode && strcmp(mode, "mfma") == 0) h->sweep_mode = 1;
  const char* sf = getenv("MGP_SWEEP_FAST");
  if (sf && (strcmp(sf, "0") == 0 || strcmp(sf, "1") == 0 || strcmp(sf, "2") == 0)) h->sweep_fast = atoi(sf);
  const char* pt = getenv("MGP_PF_TRIPS");
  if (pt && atoi(pt) >= 1 && (atoi(pt) & (atoi(pt) - 1)) == 0) h->pf_trips = atoi(pt);
  const char* pa = getenv("MGP_PF_AHEAD");
  if (pa && atoi(pa) >= 0) h->pf_ahead = atoi(pa);
  const char* stg = getenv("MGP_SWEEP_TARGET");
  if (stg && atoi(stg) >= 1 && atoi(stg) <= 64) h->sweep_target_per_cu = atoi(stg);
  const char* sgr = getenv("MGP_SWEEP_GRAN");
  if (sgr && (atoi(sgr) == 64 || atoi(sgr) == 128 || atoi(sgr) == 256)) h->sweep_chunk_gran = atoi(sgr);
  const char* f32r = getenv("MGP_SWEEP_RPT32");
  if (f32r && (atoi(f32r) == 1 || atoi(f32r) == 2)) h->sweep_fast_rpt32 = atoi(f32r);
  const char* frc = getenv("MGP_SWEEP_RPT_RC");
  if (frc && (atoi(frc) == 2 || atoi(frc) == 3)) h->sweep_fast_rpt_rc = atoi(frc);
  const char* fr = getenv("MGP_SWEEP_RPT");
  if (fr && atoi(fr) >= 2 && atoi(fr) <= 4) h->sweep_fast_rpt = atoi(fr);
  const char* pm = getenv("MGP_CONTRACT_PANEL_MB");
  if (pm && atoi(pm) > 0) h->contract_panel_mb = (size_t)atoi(pm);
  const char* nz = getenv("MGP_CONTRACT_NZ");
  if (nz && atoi(nz) > 0 && atoi(nz) <= 64) h->contract_nz = atoi(nz);
  const char* gk = getenv("MGP_GEMM_KSPLIT");
  if (gk && strcmp(gk, "0") == 0) h->gemm_ksplit = 0;
  const char* sb = getenv("MGP_SKINNY_BPC");
  if (sb && atoi(sb) > 0 && atoi(sb) <= 8) h->skinny_blocks_per_cu = atoi(sb);
  const char* sst = getenv("MGP_SKINNY_STAGGER");
  if (sst && atoi(sst) >= 0 && atoi(sst) <= 200) h->skinny_stagger = atoi(sst);
  const char* sk = getenv("MGP_SKINNY");
  if (sk && strcmp(sk, "reg") == 0) h->skinny_mode = 0;
  const char* skd = getenv("MGP_SKINNY_DEFER");
  if (skd && strcmp(skd, "0") == 0) h->skinny_defer = 0;
  const char* skp = getenv("MGP_SKINNY_PIPE");
  if (skp && strcmp(skp, "0") == 0) h->skinny_pipe = 0;
  const char* tf = getenv("MGP_TRI_FORM");
  if (tf) h->tri_form = atoi(tf);
  const char* fa = getenv("MGP_FUSE_AGREE");
  if (fa) h->fuse_agree = atoi(fa);
  if (getenv("MGP_SGPR_KMM_ASIDE")) h->kmm_aside = atoi(getenv("MGP_SGPR_KMM_ASIDE"));
  const char* kta = getenv("MGP_KDENSE_TA");
  if (kta && (atoi(kta) == 16 || atoi(kta) == 64)) h->kdense_ta = atoi(kta);
  const char* cd1 = getenv("MGP_CG_DENSE1");
  if (cd1) h->cg_dense1 = atoi(cd1);
  if (getenv("MGP_CG_DENSE1_COLS")) {
    const int c = atoi(getenv("MGP_CG_DENSE1_COLS"));
    h->cg_dense1_cols = c < 1 ? 1 : (c > 8 ? 8 : c);
  }
  if (getenv("MGP_D1_FIRST_POLL")) h->d1_first_poll_sleep = atoi(getenv("MGP_D1_FIRST_POLL"));
  if (getenv("MGP_D1_INJECT_ABSENT")) h->d1_inject_absent = atoi(getenv("MGP_D1_INJECT_ABSENT"));
  if (getenv("MGP_D1_OWNER_SPREAD")) h->d1_owner_spread = atoi(getenv("MGP_D1_OWNER_SPREAD"));
  const char* cpp = getenv("MGP_CG_PIPELINE_POLLS");
  if (cpp) h->poll_pipeline = atoi(cpp) != 0;
  const char* tm = getenv("MGP_TRI_MIN_N");
  if (tm && atol(tm) > 0) h->tri_min_n = atol(tm);
  const char* ns = getenv("MGP_NOSPLIT_PER_CU");
  if (ns && atoi(ns) > 0) h->nosplit_per_cu = atoi(ns);
  if (hipHostMalloc((void**)&h->host_flag, 64, hipHostMallocDefault) != hipSuccess) {
    delete h;
    return MGP_E_NOMEM;
  }
  {
    struct { double d; float f; float pad; } one{1.0, 1.0f, 0.f};
    if (hipMalloc(&h->ones, 16) != hipSuccess ||
        hipMemcpy(h->ones, &one, 16, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipHostFree(h->host_flag);
      delete h;
      return MGP_E_NOMEM;
    }
  }
  if (hipMalloc((void**)&h->dparams, MGP_MAX_D * sizeof(double)) != hipSuccess) {
    (void)hipFree(h->ones);
    (void)hipHostFree(h->host_flag);
    delete h;
    return MGP_E_NOMEM;
  }
  if (mgp_build_e2tabs(h) != MGP_OK) {
    (void)hipFree(h->dparams);
    (void)hipFree(h->ones);
    (void)hipHostFree(h->host_flag);
    delete h;
    return MGP_E_NOMEM;
  }
  *out = h;
  return MGP_OK;
}

extern "C" int mgp_create_ex(mgp_handle** out, int device, size_t workspace_bytes) {
  MGP_TRY(mgp_create(out, device));
  if (workspace_bytes == 0) return MGP_OK;
  mgp_handle* h = *out;
  if (hipMalloc(&h->pool, workspace_bytes) != hipSuccess) {
    (void)mgp_destroy(h);
    *out = nullptr;
    return MGP_E_NOMEM;
  }
  h->pool_bytes = workspace_bytes;
  return MGP_OK;
}

extern "C" size_t mgp_workspace_bytes(const mgp_handle* h) {
  if (!h) return 0;
  if (h->pool) return h->pool_used;
  // 256 bytes of alignment slack per arena, as a fixed pool would spend
  return h->ws_bytes + h->cg_bytes + h->opws_bytes + h->gen_bytes + h->pack[0].bytes + h->pack[1].bytes +
         h->tri_tab_bytes + h->prof_clk_bytes + 8 * 256;
}

extern "C" int mgp_destroy(mgp_handle* h) {
  if (!h) return MGP_OK;
  (void)hipSetDevice(h->device);
  if (h->pool) {
    (void)hipFree(h->pool);  // the arenas live inside it
  } else {
    if (h->ws) (void)hipFree(h->ws);
    if (h->cg) (void)hipFree(h->cg);
    if (h->opws) (void)hipFree(h->opws);
    if (h->gen) (void)hipFree(h->gen);
    if (h->tri_tab) (void)hipFree(h->tri_tab);
    for (auto& ps : h->pack)
      if (ps.buf) (void)hipFree(ps.buf);
    if (h->prof_clk) (void)hipFree(h->prof_clk);
  }
  if (h->host_flag) (void)hipHostFree(h->host_flag);
  for (auto& e : h->poll_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->aside_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->aside_stream) (void)hipStreamDestroy(h->aside_stream);
  if (h->ones) (void)hipFree(h->ones);
  if (h->dparams) (void)hipFree(h->dparams);
  if (h->e2tabs) (void)hipFree(h->e2tabs);
  for (auto& pr : h->prof_ev) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  delete h;
  return MGP_OK;
}

extern "C" int mgp_set_stream(mgp_handle* h, void* hip_stream) {
  if (!h) return MGP_E_BADARG;
  h->stream = (hipStream_t)hip_stream;
  return MGP_OK;
}

extern "C" const char* mgp_last_error(mgp_handle* h) { return h ? h->err : "invalid handle"; }

extern "C" int mgp_profile_enable(mgp_handle* h, int on) {
  if (!h) return MGP_E_BADARG;
  if (on) {  // clock stamps of the profiled launches: one more arena of the workspace (4 MB) -- with a fixed pool
             // (mgp_create_ex) it comes out of the pool or the call returns MGP_E_NOMEM; nothing is allocated then
    const size_t bytes = (size_t)MGP_PROF_CLK_LAUNCHES * MGP_PROF_CLK_WORDS * sizeof(unsigned long long);
    MGP_TRY(mgp_reserve(h, &h->prof_clk, &h->prof_clk_bytes, bytes));
    MGP_HIP(h, hipMemsetAsync(h->prof_clk, 0, bytes, h->stream));
    h->prof_clk_launches = 0;
  }
  h->prof_on = on != 0;
  h->prof_used = 0;
  return MGP_OK;
}

extern "C" int mgp_profile_read(mgp_handle* h, int64_t* launches, double* total_ms) {
  if (!h || !launches || !total_ms) return MGP_E_BADARG;
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (size_t i = 0; i < h->prof_used; ++i) {
    float ms = 0.f;
    MGP_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[i].first, h->prof_ev[i].second));
    tot += ms;
  }
  *launches = (int64_t)h->prof_used;
  *total_ms = tot;
  h->prof_used = 0;
  return MGP_OK;
}

extern "C" int mgp_profile_read_each(mgp_handle* h, double* ms_out, int64_t capacity, int64_t* launches) {
  if (!h || !launches || (capacity > 0 && !ms_out)) return MGP_E_BADARG;
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  const int64_t n = (int64_t)h->prof_used;
  for (int64_t i = 0; i < n && i < capacity; ++i) {
    float ms = 0.f;
    MGP_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[i].first, h->prof_ev[i].second));
    ms_out[i] = ms;
  }
  *launches = n;
  h->prof_used = 0;
  return MGP_OK;
}

// ---- sustained shader clock of the profiled sweep launches (bench.py's roofline: the issue-slot fraction at the
// clock the chip actually held, not only at the 2.4 GHz of the datasheet).  While profiling is on, up to 16
// workgroups of every bracketed sweep launch stamp (s_memrealtime, s_memtime) -- the 100 MHz constant counter and
// the shader-clock counter -- when they start and when their loop has ended (mgp_prof_stamp, mgp_common.h); the
// clock a workgroup saw is the ratio of the two differences.  A resident sampling wave on a second stream was the
// first form and was measured to PERTURB the kernel it watched (one CU can then hold one 512-thread workgroup
// instead of two: a rank's 0.326 ms sweep took 0.383 ms), so the stamps ride inside the launch instead.
extern "C" int mgp_profile_read_clocks(mgp_handle* h, double* mhz_out, int64_t capacity, int64_t* samples) {
  if (!h || !samples || (capacity > 0 && !mhz_out)) return MGP_E_BADARG;
  *samples = 0;
  if (!h->prof_clk || h->prof_clk_launches == 0) return MGP_OK;
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  const size_t nl = h->prof_clk_launches < (size_t)MGP_PROF_CLK_LAUNCHES ? h->prof_clk_launches : (size_t)MGP_PROF_CLK_LAUNCHES;
  std::vector<unsigned long long> buf(nl * MGP_PROF_CLK_WORDS);
  MGP_HIP(h, hipMemcpy(buf.data(), h->prof_clk, buf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  int64_t k = 0;
  for (size_t l = 0; l < nl; ++l)
    for (int sl = 0; sl < 16; ++sl) {
      const unsigned long long* w = &buf[l * MGP_PROF_CLK_WORDS + 4 * sl];
      if (w[0] == 0 || w[2] <= w[0] || w[3] <= w[1]) continue;  // slot not used / workgroup left early
      if (k < capacity) mhz_out[k] = (double)(w[3] - w[1]) / (double)(w[2] - w[0]) * 100.0;
      ++k;
    }
  *samples = k;
  return MGP_OK;
}

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// out[c] = sum_r A[r,c]*B[r,c]: block = 64 columns x 4 row-slices, rows summed in fixed order
template <typename T>
__global__ __launch_bounds__(256) void colwise_dot_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                          long rows, long cols, T* __restrict__ out) {
  __shared__ T red[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + lane;
  T s = 0;
  if (c < cols)
    for (long r = slice; r < rows; r += 4) s = mgp_fma(A[r * cols + c], B[r * cols + c], s);
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && c < cols) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// stage 1 of sum(A*B): one partial per block
template <typename T>
__global__ __launch_bounds__(256) void dot_partial_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                          long count, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256)
    s += (double)A[i] * (double)B[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int mgp_colwise_dot(mgp_handle* h, int dtype, const void* A, const void* B, int64_t rows,
                               int64_t cols, void* out) {
  if (!h) return MGP_E_BADARG;
  if (dtype != MGP_F32 && dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", dtype);
  if (rows < 0 || cols < 0) return mgp_fail(h, MGP_E_SHAPE, "negative size");
  if (cols == 0) return MGP_OK;
  if (!out || (rows > 0 && (!A || !B))) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  dim3 grid((unsigned)((cols + 63) / 64));
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((colwise_dot_kernel<double>), grid, dim3(256), 0, h->stream, (const double*)A,
                       (const double*)B, rows, cols, (double*)out);
  else
    hipLaunchKernelGGL((colwise_dot_kernel<float>), grid, dim3(256), 0, h->stream, (const float*)A,
                       (const float*)B, rows, cols, (float*)out);
  MGP_LAUNCH_CHECK(h);
  return MGP_OK;
}

extern "C" int mgp_dot_all(mgp_handle* h, int dtype, const void* A, const void* B, int64_t count, double* out) {
  if (!h) return MGP_E_BADARG;
  if (dtype != MGP_F32 && dtype != MGP_F64) return mgp_fail(h, MGP_E_DTYPE, "bad dtype %d", dtype);
  if (!out) return mgp_fail(h, MGP_E_BADARG, "out is NULL");
  *out = 0.0;
  if (count <= 0) return count == 0 ? MGP_OK : mgp_fail(h, MGP_E_SHAPE, "negative count");
  if (!A || !B) return mgp_fail(h, MGP_E_BADARG, "NULL data pointer");
  const int nb = 512;
  MGP_TRY(mgp_reserve(h, &h->ws, &h->ws_bytes, nb * sizeof(double)));
  double* part = (double*)h->ws;
  if (dtype == MGP_F64)
    hipLaunchKernelGGL((dot_partial_kernel<double>), dim3(nb), dim3(256), 0, h->stream, (const double*)A,
                       (const double*)B, count, part);
  else
    hipLaunchKernelGGL((dot_partial_kernel<float>), dim3(nb), dim3(256), 0, h->stream, (const float*)A,
                       (const float*)B, count, part);
  MGP_LAUNCH_CHECK(h);
  double host[512];
  MGP_HIP(h, hipMemcpyAsync(host, part, nb * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  MGP_HIP(h, hipStreamSynchronize(h->stream));
  double s = 0;
  for (int i = 0; i < nb; ++i) s += host[i];
  *out = s;
  return MGP_OK;
}
