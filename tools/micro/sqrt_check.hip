// Accuracy of mgp_sqrt_pos (csrc/mgp_math.h: v_rsq_f64 + Goldschmidt step + one residual correction) against the
// correctly rounded square root, over 2^26 positive doubles spread across 200 binades: maximum and histogram of the
// error in ulps.   hipcc -O3 --offload-arch=gfx950 -o sqrt_check tools/micro/sqrt_check.hip && ./sqrt_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../conjugate-gradient-sparse-gp_amd/csrc/mgp_math.h"

template <int VARIANT>
__global__ void check(unsigned long long* hist, unsigned long long seed) {
  unsigned long long x = seed + (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  for (int it = 0; it < 256; ++it) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;  // xorshift64
    // exponent in [923, 1123] (2^-100 .. 2^100), random mantissa
    const unsigned long long bits = ((923ull + (x >> 52) % 201ull) << 52) | (x & 0xFFFFFFFFFFFFFull);
    const double v = __builtin_bit_cast(double, bits);
    double a;
    if (VARIANT == 0) {
      a = mgp_sqrt_pos(v);
    } else if (VARIANT == 1) {  // Newton on the seed, no Goldschmidt step
      const double y = __builtin_amdgcn_rsq(v);
      const double g = v * y, hh = 0.5 * y;
      a = __builtin_fma(__builtin_fma(-g, g, v), hh, g);
    } else {  // seed accuracy alone: x * rsq(x)
      a = v * __builtin_amdgcn_rsq(v);
    }
    const double b = __builtin_sqrt(v);
    long long d = (long long)__builtin_bit_cast(unsigned long long, a) - (long long)__builtin_bit_cast(unsigned long long, b);
    if (d < 0) d = -d;
    atomicAdd(&hist[d > 7 ? 7 : d], 1ull);
  }
}

int main() {
  unsigned long long* h;
  hipMalloc(&h, 8 * sizeof(unsigned long long));
  hipMemset(h, 0, 8 * sizeof(unsigned long long));
  const char* names[3] = {"mgp_sqrt_pos (seed + Goldschmidt step + one correction)", "seed + one Newton correction", "seed alone: x * v_rsq_f64(x)"};
  for (int variant = 0; variant < 3; ++variant) {
    hipMemset(h, 0, 8 * sizeof(unsigned long long));
    if (variant == 0) hipLaunchKernelGGL(check<0>, dim3(1024), dim3(256), 0, 0, h, 12345ull);
    if (variant == 1) hipLaunchKernelGGL(check<1>, dim3(1024), dim3(256), 0, 0, h, 12345ull);
    if (variant == 2) hipLaunchKernelGGL(check<2>, dim3(1024), dim3(256), 0, 0, h, 12345ull);
    unsigned long long r[8];
    hipMemcpy(r, h, sizeof(r), hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += r[i];
    printf("%s vs correctly rounded sqrt over %llu doubles in [2^-100, 2^100]:\n ", names[variant], tot);
    for (int i = 0; i < 8; ++i) printf(" %s%d ulp: %llu;", i == 7 ? ">= " : "", i, r[i]);
    printf("\n");
  }
  return 0;
}
