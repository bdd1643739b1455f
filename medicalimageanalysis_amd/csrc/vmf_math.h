// vmf_math.h -- per-row / per-element arithmetic of the m = 3 path and of the KL variants.
// Same operation order as the reference lines cited, in the tensor's dtype T.
#pragma once
#include "bessel_dev.h"
#include "common.h"

// sampler/von_mises_fisher.py:30   l = loc / (||loc|| + 1e-5)
template <typename T>
VMF_HD void v3_normalise_loc(const T r[3], T l[3], T* n_out) {
  T n = M<T>::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  T den = n + (T)1e-5;
  l[0] = r[0] / den; l[1] = r[1] / den; l[2] = r[2] / den;
  *n_out = n;
}

// sampler/von_mises_fisher.py:58-63   w = 1 + logsumexp([log u, log(1-u) - 2k]) / k
// (torch.logsumexp: log(sum(exp(x - max))) + max, an infinite max replaced by 0)
template <typename T>
VMF_HD T v3_w(T u, T k, T* L_out, T* x2_out) {
  T x1 = M<T>::log(u);
  T x2 = M<T>::log((T)1 - u) - (T)2 * k;
  T mx = x1 > x2 ? x1 : x2;
  if (x1 != x1 || x2 != x2) mx = x1 + x2;  // NaN
  T mxs = (mx == (T)INFINITY || mx == (T)-INFINITY) ? (T)0 : mx;
  T L = M<T>::log(M<T>::exp(x1 - mxs) + M<T>::exp(x2 - mxs)) + mxs;
  *L_out = L;
  *x2_out = x2;
  return (T)1 + L / k;
}

// sampler/von_mises_fisher.py:77-79
template <typename T>
VMF_HD void v3_tangent(const T eps[3], const T l[3], T t[3], T v[3], T* p_out, T* s_out) {
  T p = eps[0] * l[0] + eps[1] * l[1] + eps[2] * l[2];
  t[0] = eps[0] - p * l[0]; t[1] = eps[1] - p * l[1]; t[2] = eps[2] - p * l[2];
  T s = M<T>::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  T den = s + (T)1e-8;
  v[0] = t[0] / den; v[1] = t[1] / den; v[2] = t[2] / den;
  *p_out = p;
  *s_out = s;
}

// sampler/von_mises_fisher.py:83   sqrt(clamp(1 - w^2, 1e-10, 1))
template <typename T>
VMF_HD T v3_factor(T w, bool* inside) {
  T c = (T)1 - w * w;
  *inside = (c >= (T)1e-10) && (c <= (T)1);
  if (c < (T)1e-10) c = (T)1e-10;
  if (c > (T)1) c = (T)1;
  return M<T>::sqrt(c);
}

// Backward of one output row (what autograd derives from :30, :58-63, :77-84): given g = dL/dz row,
// returns dL/dloc row contribution (dr) and dL/dscale contribution (dk).
template <typename T>
VMF_HD void v3_backward_row(const T r[3], T k, T u, const T e[3], const T g[3], T dr[3], T* dk_out) {
  T l[3], t[3], v[3], n, p, s, L, x2;
  bool inside;
  v3_normalise_loc<T>(r, l, &n);
  T w = v3_w<T>(u, k, &L, &x2);
  v3_tangent<T>(e, l, t, v, &p, &s);
  T f = v3_factor<T>(w, &inside);
  T gl = g[0] * l[0] + g[1] * l[1] + g[2] * l[2];
  T gv = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
  T dw = gl + gv * (inside ? -w / f : (T)0);
  T sig = M<T>::exp(x2 - L);
  *dk_out = dw * (-L / (k * k) - (T)2 * sig / k);
  T dv[3], dl[3], dt[3];
  for (int c = 0; c < 3; ++c) { dv[c] = f * g[c]; dl[c] = w * g[c]; }
  T sden = s + (T)1e-8;
  T tdv = t[0] * dv[0] + t[1] * dv[1] + t[2] * dv[2];
  for (int c = 0; c < 3; ++c) dt[c] = dv[c] / sden - t[c] * tdv / (s * sden * sden);
  T dtl = dt[0] * l[0] + dt[1] * l[1] + dt[2] * l[2];
  for (int c = 0; c < 3; ++c) dl[c] += -(dtl * e[c] + p * dt[c]);
  T nden = n + (T)1e-5;
  T rdl = r[0] * dl[0] + r[1] * dl[1] + r[2] * dl[2];
  for (int c = 0; c < 3; ++c) dr[c] = dl[c] / nden - r[c] * rdl / (n * nden * nden);
}

// ---- KL variants: value and derivative per element -------------------------------------------------
struct KlBessel {  // host-prepared orders for the S^1 variants
  BesselOrder o0, o1;
};
static inline KlBessel make_kl_bessel() {
  KlBessel b;
  b.o0 = make_bessel_order(0.0, 0);
  b.o1 = make_bessel_order(1.0, 0);
  return b;
}

template <typename T>
VMF_HD T kl_value(T kappa, int variant, const KlBessel& kb, T* grad) {
  T g = 0, kl = 0;
  switch (variant) {
    case VMF_KL_SPHERICAL_PIECEWISE: {  // model/spherical_vae.py:141-184
      const T eps = (T)1e-8;
      T k = kappa < eps ? eps : kappa;  // clamp(min=eps); NaN stays NaN
      T pass = (kappa >= eps) ? (T)1 : (T)0;
      if (k < (T)1e-3) {
        kl = 0; g = 0;
      } else if (k > (T)20.0) {
        T e2 = M<T>::exp((T)-2.0 * k);
        T log_sinh = k - (T)0.69314718055994530942;
        T coth = (T)1.0 + (T)2.0 * e2;
        kl = M<T>::log(k) - log_sinh + k * coth - (T)1.0;
        g = pass * ((T)1 / k - (T)1 + coth - (T)4 * k * e2);
      } else {
        T sh = M<T>::sinh(k), S = sh + eps, C = M<T>::cosh(k);
        T coth = C / S;
        kl = M<T>::log(k) - M<T>::log(S) + k * coth - (T)1.0;
        g = pass * ((T)1 / k + k * (sh * S - C * C) / (S * S));
      }
    } break;
    case VMF_KL_TORUS_RAW: {  // model/torus_vae.py:190-194
      T sh = M<T>::sinh(kappa), C = M<T>::cosh(kappa);
      kl = M<T>::log(kappa) - M<T>::log(sh) + kappa * (C / sh) - (T)1.0;
      g = (T)1 / kappa - kappa / (sh * sh);
    } break;
    case VMF_KL_CIRCULAR_EPS: {  // CNN_experiment/circular_VAE.py:126-129
      T sh = M<T>::sinh(kappa), S = sh + (T)1e-6, C = M<T>::cosh(kappa);
      kl = M<T>::log(kappa) - M<T>::log(S) + kappa * (C / S) - (T)1.0;
      g = (T)1 / kappa + kappa * (sh * S - C * C) / (S * S);
    } break;
    case VMF_KL_S1_I0I1: {  // regular_cnn/s1_vae.py:83-88: unscaled I0, I1 overflow where exp does
      T ex = M<T>::exp(M<T>::abs(kappa));
      T i0 = (T)ive_eval(kb.o0, (double)kappa) * ex;
      T i1 = (T)ive_eval(kb.o1, (double)kappa) * ex;
      T A = i1 / i0;
      kl = kappa * A - M<T>::log(i0);
      g = kappa - A - kappa * A * A;
    } break;
    case VMF_KL_STABLE_M3: {
      T k = kappa;
      if (k < (T)0.05) {
        T k2 = k * k;
        kl = k2 * ((T)(1.0 / 6.0) + k2 * ((T)(-1.0 / 60.0) + k2 * (T)(1.0 / 567.0)));
        g = k * ((T)(1.0 / 3.0) + k2 * ((T)(-1.0 / 15.0) + k2 * (T)(6.0 / 567.0)));
      } else {
        T e2 = M<T>::exp((T)-2 * k);
        T coth = ((T)1 + e2) / ((T)1 - e2);
        T log_sinh = k + M<T>::log1p(-e2) - (T)0.69314718055994530942;
        kl = M<T>::log(k) - log_sinh + k * coth - (T)1;
        T csch2 = (T)4 * e2 / (((T)1 - e2) * ((T)1 - e2));
        g = (T)1 / k - k * csch2;
      }
    } break;
    case VMF_KL_STABLE_M2: {
      double k = (double)kappa;
      double e0 = ive_eval(kb.o0, k), e1 = ive_eval(kb.o1, k);
      double A = e1 / e0;
      kl = (T)(k * A - (::log(e0) + ::fabs(k)));
      g = (T)(k - A - k * A * A);
    } break;
    default:
      kl = (T)NAN; g = (T)NAN;
  }
  *grad = g;
  return kl;
}

// ---- general-m pieces ---------------------------------------------------------------------------
template <typename T>
VMF_HD void wood_consts(T kappa, int m, T* a, T* b, T* d) {
  T m1 = (T)(m - 1);
  T c = M<T>::sqrt((T)4 * kappa * kappa + m1 * m1);
  *b = m1 / ((T)2 * kappa + c);
  *a = (m1 + (T)2 * kappa + c) / (T)4;
  *d = (T)4 * (*a) * (*b) / ((T)1 + *b) - m1 * M<T>::log(m1);
}

// One proposal; the accept test runs in double from the T inputs (the accepted index must not depend on
// the precision of T's log).
template <typename T>
VMF_HD bool wood_try(T a, T b, T d, int m, T e, T uacc, T* w, T* omw) {
  double den = 1.0 - (1.0 - (double)b) * (double)e;
  double t = 2.0 * (double)a * (double)b / den;
  double lhs = (double)(m - 1) * ::log(t) - t + (double)d;
  T denr = (T)1 - ((T)1 - b) * e;
  *w = ((T)1 - ((T)1 + b) * e) / denr;
  *omw = (T)2 * b * e / denr;
  return lhs >= ::log((double)uacc);
}

// Counter-based proposal (hot path).  With c = 2e - 1 = sqrt(1 - U^(2/(m-2))) cos 2 pi V (Ulrich 1984: e is
// symmetric Beta((m-1)/2,(m-1)/2)) the proposal and the accept test are written without cancellation:
//   den = 1 - (1-b)(1+c)/2,  1 - w = b (1+c)/den,  w = 1 - (1-w)
//   4ab/(1+b) = m-1 identically, so  (m-1) ln t - t + d = (m-1) [log(1+s) - s],  s = t/(m-1) - 1 = (1-b) c / (2 den)
// and the test is (m-1)[log1p(s) - s] >= ln u.  Same decision as wood_try up to rounding.
template <typename T>
__device__ __forceinline__ bool wood_try_counter(T b, int m, T U, T V, T uacc, T* w, T* omw) {
  T sq = (T)1;
  if (m > 2) sq = F<T>::one_minus_exp2((T)2 / (T)(m - 2) * F<T>::log2(U));  // 1 - U^(2/(m-2))
  const T c = F<T>::sqrt(sq) * F<T>::cos2pi(V);
  const T den = (T)1 - (T)0.5 * ((T)1 - b) * ((T)1 + c);
  const T rden = F<T>::rcp(den);
  *omw = b * ((T)1 + c) * rden;
  *w = (T)1 - *omw;
  const T s = (T)0.5 * ((T)1 - b) * c * rden;
  return (T)(m - 1) * F<T>::log1p_minus_over_ln2(s) >= F<T>::log2(uacc);  // both sides divided by ln 2
}

// b(kappa) alone (what the counter-based path needs)
template <typename T>
__device__ __forceinline__ T wood_b(T kappa, int m) {
  const T m1 = (T)(m - 1);
  return m1 * F<T>::rcp((T)2 * kappa + F<T>::sqrt((T)4 * kappa * kappa + m1 * m1));
}
