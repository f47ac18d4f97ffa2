// mgp_math.h -- scalar math shared by every kernel: exp2 on a reduced argument and the
// stationary-kernel profiles of SURVEY §8a row K2 (GPflow SquaredExponential / Matern12/32/52).
//
// The header also compiles with plain g++ (MGP_HD expands to nothing) so that
// tests/test_host_math.py can check the polynomial against libm on the CPU.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define MGP_HD __host__ __device__ __forceinline__
#else
#define MGP_HD inline
#endif

#define MGP_LOG2E 1.4426950408889634074
#define MGP_LN2 0.69314718055994530942

// 2^t for double.  n = rint(t), f = t - n in [-0.5, 0.5] (exact), 2^f = 1 + f*q(f) with q the
// degree-10 near-minimax fit of (2^f - 1)/f (fit error 1.3e-17, see csrc/gen_exp2_coeffs.py),
// so 2^0 == 1 exactly and k(x,x) == variance exactly.  13 fp64 VALU instructions + cvt + ldexp.
MGP_HD double mgp_exp2(double t) {
  const double n = __builtin_rint(t);
  const double f = t - n;
  double q = 0x1.e9d3fe3952179p-32;
  q = __builtin_fma(q, f, 0x1.e6063f7217bc6p-28);
  q = __builtin_fma(q, f, 0x1.b524fae627834p-24);
  q = __builtin_fma(q, f, 0x1.62bfd47773353p-20);
  q = __builtin_fma(q, f, 0x1.ffcbfc670dcd4p-17);
  q = __builtin_fma(q, f, 0x1.430913096fd9fp-13);
  q = __builtin_fma(q, f, 0x1.5d87fe78a5276p-10);
  q = __builtin_fma(q, f, 0x1.3b2ab6fba1ddap-7);
  q = __builtin_fma(q, f, 0x1.c6b08d704a0c2p-5);
  q = __builtin_fma(q, f, 0x1.ebfbdff82c598p-3);
  q = __builtin_fma(q, f, 0x1.62e42fefa39efp-1);
  const double p = __builtin_fma(q, f, 1.0);
  // t is <= ~0 on every call path; clamp keeps the int conversion defined for huge |t|
  const double nc = n < -2000.0 ? -2000.0 : n;
  return __builtin_ldexp(p, (int)nc);
}

// Table form used by the fused sweeps (the polynomial above costs 11 dependent fp64 fmas per
// pair, and the sweeps are fp64-VALU bound).  t = n + i/2048 + g with |g| <= 2^-12:
//   u = t + 1.5*2^41 leaves m = round(2048 t) in the low mantissa word (two's complement), so
//   i = m & 2047, n = m >> 11 (floor) and g = t - (u - 1.5*2^41) exactly; then
//   2^t = ldexp(T[i] * (1 + g(c1 + g(c2 + g c3))), n),  T[i] = 2^(i/2048), c_k = ln2^k/k!.
// Truncation error (g ln2)^4/24 <= 3.4e-17.  8 fp64 instructions + 3 integer ones.
// `tab` is a 2048-entry table (LDS in the kernels) filled by mgp_exp2_tab_entry().
#define MGP_EXP2_TAB_BITS 11
#define MGP_EXP2_TAB_SIZE 2048

MGP_HD double mgp_exp2_tab_entry(int i) { return mgp_exp2((double)i * (1.0 / MGP_EXP2_TAB_SIZE)); }

// CLAMP = false is only legal when the caller has bounded |t| < 2^20 (the sweeps check
// 2(max|a|^2 + max|b|^2) per tile and fall back to the clamped loop otherwise).
template <bool CLAMP = true>
MGP_HD double mgp_exp2_tab(double t, const double* tab) {
  // keeps m inside int32 for absurdly distant points (2^-2000 == 0); a comparison, not fmax, so that a
  // NaN distance stays NaN as tf.exp would leave it (fmax(NaN, x) == x)
  if (CLAMP) t = t < -2000.0 ? -2000.0 : t;
  const double C = 0x1.8p+41;
  const double u = t + C;
  long long bits;
  __builtin_memcpy(&bits, &u, sizeof(bits));
  const int m = (int)bits;
  const double g = t - (u - C);
  double q = __builtin_fma(g, 0x1.c6b08d704a0c0p-5, 0x1.ebfbdff82c58fp-3);  // ln2^3/6, ln2^2/2
  q = __builtin_fma(q, g, 0x1.62e42fefa39efp-1);                              // ln2
  q = q * g;
  const double T = tab[m & (MGP_EXP2_TAB_SIZE - 1)];
  return __builtin_ldexp(__builtin_fma(T, q, T), m >> MGP_EXP2_TAB_BITS);
}

// Shifted form for the SE sweep: the exponent is t = s - a2 with a2 constant per owned point.
// Folding a2 into the magic constant saves the subtraction: Cq = fl(1.5*2^41 - a2) (a multiple
// of 2^-11), u = s + Cq, and with rho = (1.5*2^41 - Cq) - a2 (|rho| <= 2^-12, per owned point)
//     2^(s - a2) = 2^rho * ldexp(T[m & 2047] * 2^g, m >> 11),   g = s - (u - Cq)  (exact),
// m = low word of u.  The caller multiplies the accumulated sum by 2^rho once per owned point.
// Only legal when |s - a2| < 2^19 (same bound as the unclamped table form).
MGP_HD double mgp_exp2_tab_shifted(double s, double Cq, const double* tab) {
  const double u = s + Cq;
  long long bits;
  __builtin_memcpy(&bits, &u, sizeof(bits));
  const int m = (int)bits;
  const double g = s - (u - Cq);
  double q = __builtin_fma(g, 0x1.c6b08d704a0c0p-5, 0x1.ebfbdff82c58fp-3);
  q = __builtin_fma(q, g, 0x1.62e42fefa39efp-1);
  q = q * g;
  const double T = tab[m & (MGP_EXP2_TAB_SIZE - 1)];
  return __builtin_ldexp(__builtin_fma(T, q, T), m >> MGP_EXP2_TAB_BITS);
}
#define MGP_EXP2_MAGIC 0x1.8p+41

MGP_HD float mgp_exp2(float t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(t);  // v_exp_f32, 1 ulp
#else
  return exp2f(t);
#endif
}

MGP_HD double mgp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MGP_HD float mgp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

MGP_HD double mgp_sqrt(double x) { return __builtin_sqrt(x); }

// sqrt for x known to be a positive normal number well inside the exponent range (the Matern
// profiles clamp their argument to >= c^2 * 1e-36): v_rsq_f64 seed + Goldschmidt step + ONE
// residual correction, without the range scaling and special-case selects of the generic
// lowering (8 instructions instead of ~17 in an fp64-VALU-bound loop).  The generic lowering's second
// correction only buys correct rounding: after the Goldschmidt step the error is the square of the seed's
// (2^-46 from a 2^-23 seed), after one correction its square again -- the result is within an ulp, and a
// Matern kernel value moves by q * 2^-53 of itself (round 2: 57.3 -> 55.3 instructions per pair at C5).
MGP_HD double mgp_sqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
#else
  return __builtin_sqrt(x);
#endif
}
MGP_HD float mgp_sqrt_pos(float x) { return __builtin_sqrtf(x); }
MGP_HD float mgp_sqrt(float x) { return __builtin_sqrtf(x); }

// Input scaling c_kind such that, with a = x*c/l and b = z*c/l,
//   s := |a-b|^2 = c^2 r^2   feeds the profile directly in base 2:
//   SE        k = var * 2^(-s)                      c^2 = log2(e)/2
//   Matern12  k = var * 2^(-q),           q = sqrt(s) = r log2e        c = log2e
//   Matern32  k = var * (1 + q ln2) 2^(-q),        q = sqrt3 r log2e   c = sqrt3 log2e
//   Matern52  k = var * (1 + q ln2 + q^2 ln2^2/3) 2^(-q), q = sqrt5 r log2e
// (GPflow: K_r2 / K_r of gpflow/kernels/stationaries.py; r = sqrt(max(r2, 1e-36)).)
inline double mgp_profile_scale(int kind) {
  switch (kind) {
    case 0: return std::sqrt(0.5 * MGP_LOG2E);
    case 1: return MGP_LOG2E;
    case 2: return std::sqrt(3.0) * MGP_LOG2E;
    default: return std::sqrt(5.0) * MGP_LOG2E;
  }
}

// neg_s = 2 a.b - |a|^2 - |b|^2 = -s (the expansion form GPflow's square_distance uses).
// clamp = c^2 * 1e-36 (GPflow's floor under the sqrt, in scaled units).
// exp2 provider: E2Poly evaluates the polynomial (or v_exp_f32), E2Tab<double> reads the table
struct E2Poly {
  MGP_HD double operator()(double t) const { return mgp_exp2(t); }
  MGP_HD float operator()(float t) const { return mgp_exp2(t); }
};
template <bool CLAMP = true>
struct E2Tab {
  const double* tab;
  MGP_HD double operator()(double t) const { return mgp_exp2_tab<CLAMP>(t, tab); }
  MGP_HD float operator()(float t) const { return mgp_exp2(t); }
};

template <int KIND, typename T, typename E2 = E2Poly>
MGP_HD T mgp_profile(T neg_s, T clamp, E2 e2 = E2()) {
  if (KIND == 0) {
    return e2(neg_s);
  } else {
    T s = -neg_s;
    s = s < clamp ? clamp : s;  // tf.maximum(r2, 1e-36) keeps NaN: so does this form (s > clamp ? s : clamp would not)
    const T q = mgp_sqrt_pos(s);
    const T e = e2(-q);
    if (KIND == 1) return e;
    if (KIND == 2) return mgp_fma(q, (T)MGP_LN2, (T)1.0) * e;
    const T c2 = (T)(MGP_LN2 * MGP_LN2 / 3.0);
    return mgp_fma(mgp_fma(q, c2, (T)MGP_LN2), q, (T)1.0) * e;
  }
}
