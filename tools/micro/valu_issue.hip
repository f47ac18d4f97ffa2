// Issue-cost probe for the instruction mix of the fused fp64 sweep (csrc/sweep.hip) on gfx950.
// Every body is inline asm on independent registers, so the instruction stream is exactly what is
// written; cycles come from s_memtime around the loop, wall time from HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

enum Mix {
  FMA64_16 = 0,        // 16 v_fma_f64
  ADD64_16,            // 16 v_add_f64
  MUL64_16,            // 16 v_mul_f64
  FMA64_16_AND4,       // 16 v_fma_f64 + 4 v_and_b32
  FMA64_16_AND8,       // 16 v_fma_f64 + 8 v_and_b32
  FMA64_16_LSHLADD4,   // 16 v_fma_f64 + 4 v_lshl_add_u32
  FMA64_16_LDEXP4,     // 16 v_fma_f64 + 4 v_ldexp_f64
  FMA64_12_LDEXP4,     // 12 v_fma_f64 + 4 v_ldexp_f64 (is ldexp a full-rate fp64 slot?)
  AND_16,              // 16 v_and_b32
  FMA32_16,            // 16 v_fma_f32
  FMA64_16_SGPR,       // 16 v_fma_f64 with one SGPR-pair operand
  FMA64_16_DSB128,     // 16 v_fma_f64 + 1 ds_read_b128 (uniform address)
  FMA64_16_DSGATHER4,  // 16 v_fma_f64 + 4 ds_read_b64 (per-lane pseudo-random address)
  FMA64_16_DSGATHER4_NC,  // same, conflict-free addresses (lane*8)
  PKFMA32_16,          // 16 v_pk_fma_f32
  EXP32_16,            // 16 v_exp_f32
  PKFMA32_12_EXP4,     // 12 v_pk_fma_f32 + 4 v_exp_f32 (does the transcendental overlap packed math?)
  RSQ64_4_FMA12,       // 4 v_rsq_f64 + 12 v_fma_f64
  NMIX
};
static const char* kNames[NMIX] = {"16 fma64", "16 add64", "16 mul64", "16 fma64 + 4 and32", "16 fma64 + 8 and32",
                                   "16 fma64 + 4 lshl_add", "16 fma64 + 4 ldexp64", "12 fma64 + 4 ldexp64",
                                   "16 and32", "16 fma32", "16 fma64 (sgpr operand)", "16 fma64 + 1 ds_read_b128 bcast",
                                   "16 fma64 + 4 ds_read_b64 gather (random)", "16 fma64 + 4 ds_read_b64 (conflict-free)",
                                   "16 pk_fma32", "16 exp32", "12 pk_fma32 + 4 exp32", "4 rsq64 + 12 fma64"};
static const int kInstr[NMIX] = {16, 16, 16, 20, 24, 20, 20, 16, 16, 16, 16, 17, 20, 20, 16, 16, 16, 16};

#define F64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b))
#define REP16(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11); M(12); M(13); M(14); M(15)
#define REP12(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11)

template <int MIX>
__global__ __launch_bounds__(256) void probe(double* out, long long* cyc, int iters, double a0, double b0, int c0) {
  const long long r0 = wall_clock64();
  __shared__ double lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = 1.0 + e * 1e-9;
  __syncthreads();
  double x[16];
  int y[8];
  float z[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = 1.0 + i + threadIdx.x * 1e-3; z[i] = 1.0f + i; }
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = threadIdx.x * 2654435761u + i;
  double a = a0, b = b0;
  float af = (float)a0, bf = (float)b0;
  int c = c0;
  const double as = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)0) | 0x3ff0000000000000LL);
  unsigned addr_r = ((threadIdx.x * 2654435761u) >> 7) & 0x7ff8;  // random 8-byte slots in 32 KB
  unsigned addr_l = (threadIdx.x & 63) * 8;
  double d0, d1, d2, d3;
  double q0, q1;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
   for (int rep = 0; rep < 8; ++rep) {
    if (MIX == FMA64_16) { REP16(F64); }
    if (MIX == ADD64_16) {
#define A64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(x[i]) : "v"(a))
      REP16(A64);
    }
    if (MIX == MUL64_16) {
#define M64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(x[i]) : "v"(a))
      REP16(M64);
    }
    if (MIX == FMA64_16_AND4 || MIX == FMA64_16_AND8) {
#define AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(y[i]) : "v"(c))
      F64(0); F64(1); F64(2); F64(3); AND(0); F64(4); F64(5); F64(6); F64(7); AND(1);
      F64(8); F64(9); F64(10); F64(11); AND(2); F64(12); F64(13); F64(14); F64(15); AND(3);
      if (MIX == FMA64_16_AND8) { AND(4); AND(5); AND(6); AND(7); }
    }
    if (MIX == FMA64_16_LSHLADD4) {
#define LA(i) asm volatile("v_lshl_add_u32 %0, %1, 9, %0" : "+v"(y[i]) : "v"(c))
      F64(0); F64(1); F64(2); F64(3); LA(0); F64(4); F64(5); F64(6); F64(7); LA(1);
      F64(8); F64(9); F64(10); F64(11); LA(2); F64(12); F64(13); F64(14); F64(15); LA(3);
    }
    if (MIX == FMA64_16_LDEXP4 || MIX == FMA64_12_LDEXP4) {
#define LD(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c))
      F64(0); F64(1); F64(2); F64(3); LD(12); F64(4); F64(5); F64(6); F64(7); LD(13);
      F64(8); F64(9); F64(10); F64(11); LD(14);
      if (MIX == FMA64_16_LDEXP4) { F64(12); F64(13); F64(14); F64(15); }
      LD(15);
    }
    if (MIX == AND_16) {
#define AND16(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(y[i & 7]) : "v"(c))
      REP16(AND16);
    }
    if (MIX == FMA32_16) {
#define F32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[i]) : "v"(af), "v"(bf))
      REP16(F32);
    }
    if (MIX == PKFMA32_16) {
#define PK32(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b))
      REP16(PK32);
    }
    if (MIX == EXP32_16) {
#define EX32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(z[i]))
      REP16(EX32);
    }
    if (MIX == PKFMA32_12_EXP4) {
      REP12(PK32);
      EX32(12); EX32(13); EX32(14); EX32(15);
    }
    if (MIX == RSQ64_4_FMA12) {
      REP12(F64);
#define RSQ(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[i]))
      RSQ(12); RSQ(13); RSQ(14); RSQ(15);
    }
    if (MIX == FMA64_16_SGPR) {
#define F64S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "s"(as))
      REP16(F64S);
    }
    if (MIX == FMA64_16_DSB128) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(*(__attribute__((ext_vector_type(2))) double*)&d0) : "v"(0u) : "memory");
      REP16(F64);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MIX == FMA64_16_DSGATHER4 || MIX == FMA64_16_DSGATHER4_NC) {
      const unsigned ad = MIX == FMA64_16_DSGATHER4 ? addr_r : addr_l;
      asm volatile("ds_read_b64 %0, %1" : "=v"(d0) : "v"(ad) : "memory");
      F64(0); F64(1); F64(2); F64(3);
      asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(d1) : "v"(ad) : "memory");
      F64(4); F64(5); F64(6); F64(7);
      asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(d2) : "v"(ad) : "memory");
      F64(8); F64(9); F64(10); F64(11);
      asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(d3) : "v"(ad) : "memory");
      F64(12); F64(13); F64(14); F64(15);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3));
    }
   }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + z[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0) {
    cyc[4 * blockIdx.x] = t1 - t0;
    cyc[4 * blockIdx.x + 1] = r0;
    cyc[4 * blockIdx.x + 2] = wall_clock64();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    cyc[4 * blockIdx.x + 3] = hw;
  }
}

template <int MIX>
void run(int blocks, int iters) {
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MIX>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 0.999, 1e-3, 0x7ffff);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MIX>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999, 1e-3, 0x7ffff);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
  long long rmin = h[1], rmax = h[2];
  double busy = 0;
  std::vector<double> cps;
  for (int b = 0; b < blocks; ++b) {
    rmin = std::min(rmin, h[4 * b + 1]);
    rmax = std::max(rmax, h[4 * b + 2]);
    busy += (double)(h[4 * b + 2] - h[4 * b + 1]);
    cps.push_back((double)h[4 * b] / (double)(h[4 * b + 2] - h[4 * b + 1]) * 0.1);  // GHz (realtime = 100 MHz)
  }
  std::sort(cps.begin(), cps.end());
  printf("   span %.3f ms (100 MHz ticks), mean concurrent blocks per CU %.2f, in-kernel clock median %.2f GHz (min %.2f max %.2f)\n",
         (rmax - rmin) * 1e-5, busy / (double)(rmax - rmin) / 256.0, cps[cps.size() / 2], cps.front(), cps.back());
  // waves per SIMD = blocks / 256 (256 CUs, one wave of each block per SIMD)
  double clk = 0; for (double c : cps) clk += c; clk /= cps.size();
  const double instr_per_simd = (double)blocks / 256.0 * iters * 8.0 * kInstr[MIX];
  printf("   => %.2f cycles per instruction per SIMD at the in-kernel clock (%.2f GHz mean)\n", ms * 1e-3 * clk * 1e9 / instr_per_simd, clk);
  const double wps = blocks / 256.0;
  const double ticks_per_iter = (double)h[0] / iters;  // one wave's view: includes the other waves' issue
  printf("%-44s waves/SIMD %.0f: %.3f ms, s_memtime ticks per loop trip per wave %.1f -> per SIMD per trip %.1f "
         "(%.2f per instruction), wall-clock cycles at 2.4 GHz per SIMD trip %.1f\n",
         kNames[MIX], wps, ms, ticks_per_iter, ticks_per_iter / wps, ticks_per_iter / wps / kInstr[MIX],
         ms * 1e-3 * 2.4e9 / iters / wps);
  hipFree(out);
  hipFree(cyc);
}

template <int MIX>
void run_all() {
  run<MIX>(256, 20000);
  run<MIX>(4096, 5000);
}

int main() {
  run_all<FMA64_16>();
  run_all<ADD64_16>();
  run_all<MUL64_16>();
  run_all<FMA64_16_AND4>();
  run_all<FMA64_16_AND8>();
  run_all<FMA64_16_LSHLADD4>();
  run_all<FMA64_16_LDEXP4>();
  run_all<FMA64_12_LDEXP4>();
  run_all<AND_16>();
  run_all<FMA32_16>();
  run_all<FMA64_16_SGPR>();
  run_all<FMA64_16_DSB128>();
  run_all<FMA64_16_DSGATHER4>();
  run_all<FMA64_16_DSGATHER4_NC>();
  run_all<PKFMA32_16>();
  run_all<EXP32_16>();
  run_all<PKFMA32_12_EXP4>();
  run_all<RSQ64_4_FMA12>();
  return 0;
}
