// Issue-rate probe for v_mfma_f64_16x16x4_f64: NACC independent accumulators, back-to-back, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using Acc = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(256) void probe(double* out, long long* cyc, int iters, double a0, double b0) {
  Acc acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      // inline asm: accumulator pinned in arch VGPRs, no compiler-inserted AGPR copies or s_nops
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks, int threads, int iters) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10, 1.0, 2.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  const double mfmas_per_wave = (double)iters * NACC;
  const double waves = (double)blocks * threads / 64;
  const double tf = waves * mfmas_per_wave * 2048 / (ms * 1e-3) / 1e12;
  printf("NACC=%d blocks=%d threads=%d: %.3f ms, %.1f TFLOP/s, s_memtime ticks per MFMA (block 0) %.1f\n", NACC, blocks,
         threads, ms, tf, (double)h[0] / mfmas_per_wave);
  hipFree(out); hipFree(cyc);
}
int main() {
  // 256 CUs; threads=256 -> 1 wave per SIMD; 512 -> 2 waves per SIMD (two blocks of 256 also possible)
  run<4>(256, 256, 4000);
  run<16>(256, 256, 1000);
  run<16>(512, 256, 1000);
  run<16>(1024, 256, 1000);
  run<8>(512, 256, 2000);
  return 0;
}
