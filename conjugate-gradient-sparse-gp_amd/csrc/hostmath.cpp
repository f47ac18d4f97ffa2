// hostmath.cpp -- exposes mgp_math.h to CPU tests (tests/test_host_math.py).  Not part of
// the product library: it only lets the polynomial and the profiles be checked without a GPU.
#include "mgp_math.h"

extern "C" {
void mgp_host_exp2(const double* t, double* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = mgp_exp2(t[i]);
}
// k/variance for scaled squared distance s (>= 0) of the given kind
void mgp_host_profile(int kind, const double* s, double* out, long n) {
  const double c = mgp_profile_scale(kind);
  const double clamp = c * c * 1e-36;
  for (long i = 0; i < n; ++i) {
    switch (kind) {
      case 0: out[i] = mgp_profile<0, double>(-s[i], clamp); break;
      case 1: out[i] = mgp_profile<1, double>(-s[i], clamp); break;
      case 2: out[i] = mgp_profile<2, double>(-s[i], clamp); break;
      default: out[i] = mgp_profile<3, double>(-s[i], clamp); break;
    }
  }
}
double mgp_host_profile_scale(int kind) { return mgp_profile_scale(kind); }
// out[i] = 2^(s[i] - a2[i]) through the shifted table form, including the per-point 2^rho
void mgp_host_exp2_shifted(const double* s, const double* a2, double* out, long n) {
  static double tab[MGP_EXP2_TAB_SIZE];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < MGP_EXP2_TAB_SIZE; ++i) tab[i] = mgp_exp2_tab_entry(i);
    init = true;
  }
  for (long i = 0; i < n; ++i) {
    const volatile double Cq = MGP_EXP2_MAGIC - a2[i];
    const double rho = (MGP_EXP2_MAGIC - Cq) - a2[i];
    out[i] = mgp_exp2(rho) * mgp_exp2_tab_shifted(s[i], Cq, tab);
  }
}
void mgp_host_exp2_tab(const double* t, double* out, long n) {
  static double tab[MGP_EXP2_TAB_SIZE];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < MGP_EXP2_TAB_SIZE; ++i) tab[i] = mgp_exp2_tab_entry(i);
    init = true;
  }
  for (long i = 0; i < n; ++i) out[i] = mgp_exp2_tab<true>(t[i], tab);
}
}
