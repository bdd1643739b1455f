// vmf_math.h -- per-row / per-element arithmetic of the m = 3 path and of the KL variants.
// Same operation order as the reference lines cited, in the tensor's dtype T.
#pragma once
#include "bessel_dev.h"
#include "common.h"
#include "kl_m2_coeffs.h"

// MP is the math policy of the call: M<T> (accurate library functions, IEEE division) or, for fp32 kernels, MFast
// (hardware log2 / exp2 / sqrt / reciprocal).

// sampler/von_mises_fisher.py:30   l = loc / (||loc|| + 1e-5)
template <typename T, typename MP = M<T>>
VMF_HD void v3_normalise_loc(const T r[3], T l[3], T* n_out) {
  T n = MP::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  T den = n + (T)1e-5;
  l[0] = MP::div(r[0], den); l[1] = MP::div(r[1], den); l[2] = MP::div(r[2], den);
  *n_out = n;
}

// sampler/von_mises_fisher.py:58-63   w = 1 + logsumexp([log u, log(1-u) - 2k]) / k
// (torch.logsumexp: log(sum(exp(x - max))) + max, an infinite max replaced by 0)
template <typename T, typename MP = M<T>>
VMF_HD T v3_w(T u, T k, T* L_out, T* x2_out) {
  T x1 = MP::log(u);
  T x2 = MP::log((T)1 - u) - (T)2 * k;
  T mx = x1 > x2 ? x1 : x2;
  if (x1 != x1 || x2 != x2) mx = x1 + x2;  // NaN
  T mxs = (mx == (T)INFINITY || mx == (T)-INFINITY) ? (T)0 : mx;
  T L = MP::log(MP::exp(x1 - mxs) + MP::exp(x2 - mxs)) + mxs;
  *L_out = L;
  *x2_out = x2;
  return (T)1 + MP::div(L, k);
}

// sampler/von_mises_fisher.py:77-79
template <typename T, typename MP = M<T>>
VMF_HD void v3_tangent(const T eps[3], const T l[3], T t[3], T v[3], T* p_out, T* s_out) {
  T p = eps[0] * l[0] + eps[1] * l[1] + eps[2] * l[2];
  t[0] = eps[0] - p * l[0]; t[1] = eps[1] - p * l[1]; t[2] = eps[2] - p * l[2];
  T s = MP::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  T den = s + (T)1e-8;
  v[0] = MP::div(t[0], den); v[1] = MP::div(t[1], den); v[2] = MP::div(t[2], den);
  *p_out = p;
  *s_out = s;
}

// sampler/von_mises_fisher.py:83   sqrt(clamp(1 - w^2, 1e-10, 1))
template <typename T, typename MP = M<T>>
VMF_HD T v3_factor(T w, bool* inside) {
  T c = (T)1 - w * w;
  *inside = (c >= (T)1e-10) && (c <= (T)1);
  if (c < (T)1e-10) c = (T)1e-10;
  if (c > (T)1) c = (T)1;
  return MP::sqrt(c);
}

// kappa = softplus(raw) + 1, the head of every encoder (model/spherical_vae.py:70, model/torus_vae.py:103-104,
// CNN_experiment/circular_VAE.py:50): torch's softplus with beta = 1 (log1p(exp x), the identity above its threshold
// of 20).  *sig receives d kappa / d raw.
template <typename T, typename MP = M<T>>
VMF_HD T v3_softplus_kappa(T raw, T* sig) {
  if (raw > (T)20) {
    *sig = (T)1;
    return raw + (T)1;
  }
  const T e = MP::exp(raw);
  *sig = MP::div(e, (T)1 + e);
  // log1p(e): the series where 1 + e would round the small term away
  const T l1p = e < (T)1e-4 ? e * ((T)1 - e * ((T)0.5 - e * (T)(1.0 / 3.0))) : MP::log((T)1 + e);
  return l1p + (T)1;
}
template <>
VMF_HD double v3_softplus_kappa<double, M<double>>(double raw, double* sig) {
  if (raw > 20.0) {
    *sig = 1.0;
    return raw + 1.0;
  }
  const double e = ::exp(raw);
  *sig = e / (1.0 + e);
  return ::log1p(e) + 1.0;
}

// The two halves of the backward pass (what autograd derives from :30, :58-63, :77-84).
// (1) through w: given gl = sum <g, l> and gv = sum <g, v> over the output rows that use this w,
//     dL/dw = gl + gv * d sqrt(clamp(1 - w^2)) / dw, and dL/dk through w = 1 + L(k)/k.
template <typename T, typename MP = M<T>>
VMF_HD T v3_bwd_scale(T k, T u, T gl, T gv) {
  T L, x2;
  bool inside;
  const T w = v3_w<T, MP>(u, k, &L, &x2);
  const T f = v3_factor<T, MP>(w, &inside);
  const T dw = gl + gv * (inside ? -MP::div(w, f) : (T)0);
  const T sig = MP::exp(x2 - L);
  return dw * (-MP::div(L, k * k) - (T)2 * MP::div(sig, k));
}
// (2) through the direction: given dl = sum w g and dv = sum f g over the output rows that use this (loc, eps) row,
//     back through v = t / (|t| + 1e-8), t = eps - <eps, l> l and l = r / (|r| + 1e-5).
template <typename T, typename MP = M<T>>
VMF_HD void v3_bwd_loc(const T r[3], const T e[3], const T dl_in[3], const T dv[3], T dr[3]) {
  T l[3], t[3], v[3], n, p, s, dt[3], dl[3];
  v3_normalise_loc<T, MP>(r, l, &n);
  v3_tangent<T, MP>(e, l, t, v, &p, &s);
  const T sden = s + (T)1e-8;
  const T tdv = t[0] * dv[0] + t[1] * dv[1] + t[2] * dv[2];
  const T c2 = MP::div(tdv, s * sden * sden);
  for (int c = 0; c < 3; ++c) dt[c] = MP::div(dv[c], sden) - t[c] * c2;
  const T dtl = dt[0] * l[0] + dt[1] * l[1] + dt[2] * l[2];
  for (int c = 0; c < 3; ++c) dl[c] = dl_in[c] - (dtl * e[c] + p * dt[c]);
  const T nden = n + (T)1e-5;
  const T rdl = r[0] * dl[0] + r[1] * dl[1] + r[2] * dl[2];
  const T c3 = MP::div(rdl, n * nden * nden);
  for (int c = 0; c < 3; ++c) dr[c] = MP::div(dl[c], nden) - r[c] * c3;
}

// Backward of one output row: given g = dL/dz row, returns the dL/dloc row contribution (dr) and the dL/dscale
// contribution (dk).
template <typename T, typename MP = M<T>>
VMF_HD void v3_backward_row(const T r[3], T k, T u, const T e[3], const T g[3], T dr[3], T* dk_out) {
  T l[3], t[3], v[3], n, p, s, L, x2;
  bool inside;
  v3_normalise_loc<T, MP>(r, l, &n);
  const T w = v3_w<T, MP>(u, k, &L, &x2);
  v3_tangent<T, MP>(e, l, t, v, &p, &s);
  const T f = v3_factor<T, MP>(w, &inside);
  const T gl = g[0] * l[0] + g[1] * l[1] + g[2] * l[2];
  const T gv = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
  *dk_out = v3_bwd_scale<T, MP>(k, u, gl, gv);
  T dv[3], dl[3];
  for (int c = 0; c < 3; ++c) { dv[c] = f * g[c]; dl[c] = w * g[c]; }
  v3_bwd_loc<T, MP>(r, e, dl, dv, dr);
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

// BESSEL = false leaves the two forms that need the double-precision Bessel code out (they return NaN): the fp32
// kernels use it as the per-element fall-back of the fast forms below, where those two variants only fall back for a
// NaN kappa -- and without the Bessel code the kernels need a third of the registers.
template <typename T, bool BESSEL = true>
VMF_HD T kl_value(T kappa, int variant, const KlBessel& kb, T* grad) {
  T g = 0, kl = 0;
  if (!BESSEL && (variant == VMF_KL_S1_I0I1 || variant == VMF_KL_STABLE_M2)) {
    *grad = (T)NAN;
    return (T)NAN;
  }
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
    case VMF_KL_S1_I0I1: if constexpr (BESSEL) {  // regular_cnn/s1_vae.py:83-88: unscaled I0, I1 overflow where exp does
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
        // log k - log sinh k + k coth k - 1 with log sinh k = k + log(1 - e^-2k) - ln 2 and k coth k = k + 2k e^-2k /
        // (1 - e^-2k): the two k cancel on paper, so no rounding of k (1e-16 k in double) reaches the result
        T e2 = M<T>::exp((T)-2 * k);
        T om = (T)1 - e2;
        kl = M<T>::log(k) - M<T>::log1p(-e2) + (T)(0.69314718055994530942 - 1.0) +
             (e2 > (T)0 ? (T)2 * (k * e2) / om : (T)0);  // (an infinite k meets e2 = 0)
        T csch2 = (T)4 * e2 / (om * om);
        g = (T)1 / k - k * csch2;
      }
    } break;
    case VMF_KL_STABLE_M2: if constexpr (BESSEL) {
      double k = ::fabs((double)kappa);
      if (k > 200.0) {
        // k A - log I0 = (log 2 pi k - 1) / 2 - 1/(4k) - 3/(16k^2) - 25/(96k^3) - 65/(128k^4) - 3219/(2560k^5) - ...
        // (from the large-argument expansions of I0 and I1, A&S 9.7.1); the first term left out is 3.8 / k^6 < 6e-14,
        // while k (1 - A) formed from the two scaled Bessel values loses 2 k ulp (1e-13 at 200, everything at 1e16)
        double x = 1.0 / k;
        kl = (T)(0.5 * (::log(6.283185307179586476925 * k) - 1.0) -
                 x * (0.25 + x * (3.0 / 16.0 + x * (25.0 / 96.0 + x * (65.0 / 128.0 + x * (3219.0 / 2560.0))))));
        double gg = x * (0.5 + x * (0.25 + x * (3.0 / 8.0 + x * (25.0 / 32.0 + x * (65.0 / 32.0 + x * (3219.0 / 512.0))))));
        g = (T)(kappa < 0 ? -gg : gg);
      } else {
        double e0 = ive_eval(kb.o0, k), e1 = ive_eval(kb.o1, k);
        double A = e1 / e0;
        kl = (T)(k * A - (::log(e0) + k));
        double gg = k - A - k * A * A;
        g = (T)(kappa < 0 ? -gg : gg);
      }
    } break;
    default:
      kl = (T)NAN; g = (T)NAN;
  }
  *grad = g;
  return kl;
}

VMF_HD float kl_m2_value_f32(float a);  // (defined below, next to its coefficient tables)

// ---- fp32 fast path of the KL variants ------------------------------------------------------------------------
// The accurate version above spends ~500 instructions per element in libm's sinh / cosh / log / exp (and, for the
// S^1 form, two double Chebyshev sums); the streams over kappa then run at a few percent of the memory rate.  For
// fp32 tensors the contract is 1e-4 relative, which the hardware exp2 / log2 and single-precision series meet with
// two orders of magnitude to spare.
// (fast_exp2 / fast_log2 / fast_exp / fast_rcp / fast_log live in common.h)

// sinh and cosh of k >= 0 with sinhf's overflow point (k > 89.416): Taylor below 1 (no cancellation), the two
// exponentials up to 10, exp(k - ln 2) beyond (sinh = cosh to 2e-9 there)
VMF_HD void fast_sinh_cosh(float k, float* sh, float* ch) {
  if (k > 10.0f) {
    // sinhf is finite up to the float 89.415985 (0x42b2d4fc) and overflows from the next one on; k - ln 2 and its
    // product with log2 e are rounded, so the exponential alone would put that last float on the wrong side
    const float E = fast_exp(k - 0.69314718055994530942f);
    *sh = *ch = k <= 89.415985f ? ::fminf(E, 3.4028234663852886e38f) : INFINITY;
    return;
  }
  const float e = fast_exp(k), ei = fast_rcp(e);
  *ch = 0.5f * (e + ei);
  if (k < 1.0f) {
    const float k2 = k * k;
    *sh = k * (1.0f + k2 * (1.0f / 6.0f + k2 * (1.0f / 120.0f + k2 * (1.0f / 5040.0f + k2 * (1.0f / 362880.0f)))));
  } else {
    *sh = 0.5f * (e - ei);
  }
}

// ---- CNN_experiment/von_mises_fisher.py:92-108: log_prob with the reference's sign, one row ----------------------
// log k - log 4 pi - log(sinh k + 1e-10) and (for the gradient) cosh k / (sinh k + 1e-10).  Double precision through
// the library; single precision through the hardware logarithm and fast_sinh_cosh, which keeps sinhf's overflow point
// float for float (inf beyond 89.415985: the reference's -inf normaliser and NaN gradient there).  Both kernels that
// evaluate rows (whole tiles, thread per row) call these, so a row's value does not depend on the layout.
// (every rounding fixed -- no contraction left to the compiler -- so that the inlined copies agree)
template <typename T>
VMF_HD void v3_log_norm_ref(T k, T* ln, T* coth_eps) {
#pragma clang fp contract(off)
  const T sh = M<T>::sinh(k) + (T)1e-10;
  *ln = M<T>::log(k) - (T)2.5310242469692907 /* log(4 pi) */ - M<T>::log(sh);
  if (coth_eps) *coth_eps = M<T>::cosh(k) / sh;
}
template <>
VMF_HD void v3_log_norm_ref<float>(float k, float* ln, float* coth_eps) {
#pragma clang fp contract(off)
  if (!(k > 0.0f && k < INFINITY)) {  // zero, negative, NaN, inf: whatever the library makes of them
    const float sh = ::sinhf(k) + 1e-10f;
    *ln = ::logf(k) - 2.5310242469692907f - ::logf(sh);
    if (coth_eps) *coth_eps = ::coshf(k) / sh;
    return;
  }
  float sh, ch;
  fast_sinh_cosh(k, &sh, &ch);
  sh += 1e-10f;
  *ln = fast_log(k) - 2.5310242469692907f - fast_log(sh);
  if (coth_eps) *coth_eps = ch * fast_rcp(sh);
}
template <typename T>
VMF_HD void v3_normalise_loc_fixed(const T r[3], T l[3], T* n_out) {  // v3_normalise_loc with its roundings fixed
#pragma clang fp contract(off)
  const T n = M<T>::sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
  const T den = n + (T)1e-5;
  l[0] = r[0] / den; l[1] = r[1] / den; l[2] = r[2] / den;
  *n_out = n;
}
template <typename T>
VMF_HD T v3_log_prob_ref_row(const T r[3], T k, const T x[3], T* ln_out) {
#pragma clang fp contract(off)
  T l[3], nn, ln;
  v3_normalise_loc_fixed<T>(r, l, &nn);
  v3_log_norm_ref<T>(k, &ln, (T*)nullptr);
  if (ln_out) *ln_out = ln;
  const T dot = (l[0] * x[0] + l[1] * x[1]) + l[2] * x[2];
  return k * dot - ln;  // :92-95 (the reference subtracts)
}
// gradients of g * log_prob with respect to the raw loc row, kappa and x (what autograd derives from :92-108)
template <typename T>
VMF_HD void v3_log_prob_ref_bwd_row(const T r[3], T k, const T xv[3], T g, T dloc[3], T* dscale, T dx[3]) {
#pragma clang fp contract(off)
  T l[3], nn;
  v3_normalise_loc_fixed<T>(r, l, &nn);
  const T dot = (l[0] * xv[0] + l[1] * xv[1]) + l[2] * xv[2];
  T ln, ce;
  v3_log_norm_ref<T>(k, &ln, &ce);
  // d/dk [k dot - (log k - log 4pi - log(sinh k + 1e-10))] = dot - 1/k + cosh k / (sinh k + 1e-10)
  *dscale = g * (dot - (T)1 / k + ce);
  dx[0] = g * k * l[0]; dx[1] = g * k * l[1]; dx[2] = g * k * l[2];
  const T dl[3] = {g * k * xv[0], g * k * xv[1], g * k * xv[2]};
  const T nden = nn + (T)1e-5;
  const T rdl = (r[0] * dl[0] + r[1] * dl[1]) + r[2] * dl[2];
  for (int c = 0; c < 3; ++c) dloc[c] = dl[c] / nden - r[c] * rdl / (nn * nden * nden);
}

// sampler/von_mises_fisher.py:42 on one row: loc * (ive(3/2, k) / ive(1/2, k)), each ive rounded to T as the reference's
// two calls round theirs
template <typename T>
VMF_HD void v3_mean_row(const T r[3], T k, const BesselOrder& o32, const BesselOrder& o12, T out[3]) {
#pragma clang fp contract(off)
  T l[3], nn;
  v3_normalise_loc_fixed<T>(r, l, &nn);
  double i12, i32;
  const double kd = (double)k;
  if (kd > 0.0 && kd < 1.0e300) {
    ive_half_pair(kd, &i12, &i32);
  } else {  // zero, negative, infinite, NaN: what the two calls return there
    i32 = ive_eval(o32, kd);
    i12 = ive_eval(o12, kd);
  }
  const T a = (T)i32, b = (T)i12;
  const T A = a / b;
  out[0] = l[0] * A; out[1] = l[1] * A; out[2] = l[2] * A;
}

// Unrolled on purpose (unlike the double-precision sums): single-precision coefficients are free 32-bit literals of
// the fma, so a term costs two instructions instead of a scalar load and the loop bookkeeping.
template <int n>
VMF_HD float cheb_sum_f32(const float (&c)[n], float t) {
  float b0 = 0.0f, b1 = 0.0f;
  const float t2 = 2.0f * t;
#pragma unroll
  for (int k = n - 1; k >= 1; --k) {
    const float b = ::fmaf(t2, b0, c[k]) - b1;
    b1 = b0;
    b0 = b;
  }
  return ::fmaf(t, b0, c[0]) - b1;
}
// log k - log(sinh k + eps) + k cosh k / (sinh k + eps) - 1 and its derivative for k > 0 (eps >= 0): the common
// form of model/spherical_vae.py:161-169, model/torus_vae.py:190-194, CNN_experiment/circular_VAE.py:126-129.
// Below 1 everything is expressed through the small quantities p = sinh k / k - 1, c1 = cosh k - 1 and
// q = (sinh k + eps) / k - 1, so nothing cancels (the textbook order loses all digits of the O(k^2) result to
// log k - log sinh k and 1/k - k / sinh^2 k).  The derivative uses cosh^2 - sinh^2 = 1:
// d/dk = 1/k + k (eps sinh k - 1) / (sinh k + eps)^2.
// The straight-line m = 3 form in em = e^-k (derivation at its use below), value alone; k >= 1.  Three transcendental
// instructions: log k - log D is taken as one logarithm.
VMF_HD float kl_m3_unified_value(float k, float eps) {
  const float em = fast_exp2(-1.44269504088896340736f * k), e2 = em * em;
  const float D = 1.0f - e2 + 2.0f * eps * em;
  const float rD = fast_rcp(D);
  return fast_log(k * rD) + (0.69314718055994530942f - 1.0f) + 2.0f * k * (e2 - eps * em) * rD;
}
// the overflow-free m = 3 form for k >= 1: log k - log sinh k + k coth k - 1 with the two O(k) terms cancelled by hand,
// k coth k - k = 2 k e2 / (1 - e2), e2 = e^-2k
VMF_HD float kl_m3_stable_value(float k) {
  const float e2 = fast_exp(-2.0f * k);
  const float rom = fast_rcp(1.0f - e2);
  return fast_log(k * rom) + (0.69314718055994530942f - 1.0f) + 2.0f * (k * e2) * rom;  // (k e2 first: 2k may overflow)
}
// The reference's forms hold sinh k: finite up to sinhf's overflow point 89.416, inf / inf = NaN beyond.  The straight-line
// form serves everything up to KL_M3_UNIFIED_MAX, the value is NaN from KL_M3_NAN_MIN on, and only the window between
// them is left to the form that builds sinh k and so decides the overflow itself.
constexpr float KL_M3_UNIFIED_MAX = 89.40f, KL_M3_NAN_MIN = 89.42f;
VMF_HD void kl_m3_eps_fast(float k, float eps, float* kl, float* g, bool finite_beyond = false) {
  if (k < 1.0f) {
    const float k2 = k * k;
    const float p = k2 * (1.0f / 6.0f + k2 * (1.0f / 120.0f + k2 * (1.0f / 5040.0f + k2 * (1.0f / 362880.0f))));
    const float c1 = k2 * (0.5f + k2 * (1.0f / 24.0f + k2 * (1.0f / 720.0f + k2 * (1.0f / 40320.0f))));
    const float rk = fast_rcp(k);
    const float q = p + eps * rk;
    const float l1p = q < 0.01f ? q * (1.0f - q * (0.5f - q * (1.0f / 3.0f))) : fast_log(1.0f + q);
    const float opq = 1.0f + q;
    const float ropq = fast_rcp(opq);
    *kl = (c1 - q) * ropq - l1p;
    const float sh = k * (1.0f + p);
    // (q / (1+q)) ((2+q) / (1+q)) / k: factors of at most 1 and 2, so a tiny kappa (q = eps / k up to 1e32) neither
    // overflows q (2 + q) nor underflows 1 / (1+q)^2 on the way to the reference's 1 / kappa
    *g = (q * ropq) * ((2.0f + q) * ropq) * rk + eps * sh * ropq * ropq * rk;
    return;
  }
  // (finite_beyond: the caller's form has no overflow to reproduce -- e^-k underflows to 0 and leaves log k + ln 2 - 1)
  if (k <= KL_M3_UNIFIED_MAX || finite_beyond) {
    // 1 <= k <= 89.40, just below sinhf's overflow point 89.416: everything through em = e^-k (one exponential, nothing overflows, nothing cancels).  With
    // D = 1 - e^-2k + 2 eps e^-k:  sinh k + eps = D e^k / 2,  cosh k / (sinh k + eps) = (1 + e^-2k) / D, so
    //   kl = log k + (ln 2 - 1) - log D + 2 k (e^-2k - eps e^-k) / D
    //   d/dk = 1/k + k (2 eps e^-k (1 - e^-2k) - 4 e^-2k) / D^2
    // -- the reference's expression term for term, eps included.  One straight-line path for every kappa an encoder
    // produces (softplus + 1 >= 1), so a wave does not pay for several regions.
    const float em = fast_exp2(-1.44269504088896340736f * k), e2 = em * em;
    const float D = 1.0f - e2 + 2.0f * eps * em;
    const float rD = fast_rcp(D);
    *kl = kl_m3_unified_value(k, eps);
    *g = fast_rcp(k) + k * (2.0f * eps * em * (1.0f - e2) - 4.0f * e2) * (rD * rD);
    return;
  }
  float sh, C;
  fast_sinh_cosh(k, &sh, &C);
  const float S = sh + eps;
  const float rS = fast_rcp(S);  // flushes to 0 once S > 2^126; every use below tolerates that
  // cosh / (sinh + eps): beyond 10 the two are one number E, and E / (E + eps) = 1 - eps / (E + eps) stays exact
  // when 1 / (E + eps) underflows; an overflowed E gives the reference's inf / inf
  const float ratio = k > 10.0f ? (sh > 3.4028234e38f ? NAN : 1.0f - eps * rS) : C * rS;
  *kl = fast_log(k) - fast_log(S) + k * ratio - 1.0f;
  *g = fast_rcp(k) + k * (eps * sh - 1.0f) * (rS * rS);
}

// Same values and NaN / inf behaviour as kl_value<float> (to the tolerance above); returns false when there is no
// fast form (non-positive or NaN kappa in the unguarded variants, which are the accurate form's business).
VMF_HD bool kl_value_fast(float kappa, int variant, float* kl_out, float* grad) {
  float kl = 0.0f, g = 0.0f;
  switch (variant) {
    case VMF_KL_SPHERICAL_PIECEWISE: {
      const float eps = 1e-8f;
      if (kappa != kappa) return false;
      const float k = kappa < eps ? eps : kappa;
      const float pass = (kappa >= eps) ? 1.0f : 0.0f;
      if (k < 1e-3f) {
        kl = 0.0f; g = 0.0f;
      } else if (k >= 1.0f && k <= 1000.0f) {
        // the mid form; beyond 20 the reference switches to log k - (k - ln 2) + k (1 + 2 e^-2k) - 1, which differs
        // from it by O(e^-2k) < 1e-17: the same float.  (Beyond 1000 the reference's own fp32 result carries the
        // rounding of k - k: its order of operations is kept there, below.)
        kl_m3_eps_fast(k, eps, &kl, &g, true);
        g *= pass;
      } else if (k > 20.0f) {
        const float e2 = fast_exp(-2.0f * k);
        const float coth = 1.0f + 2.0f * e2;
        kl = fast_log(k) - (k - 0.69314718055994530942f) + k * coth - 1.0f;
        g = pass * (fast_rcp(k) - 1.0f + coth - 4.0f * k * e2);
      } else {
        kl_m3_eps_fast(k, eps, &kl, &g);
        g *= pass;
      }
    } break;
    case VMF_KL_TORUS_RAW: {
      if (!(kappa > 0.0f)) return false;
      kl_m3_eps_fast(kappa, 0.0f, &kl, &g);
    } break;
    case VMF_KL_CIRCULAR_EPS: {
      if (!(kappa > 0.0f)) return false;
      kl_m3_eps_fast(kappa, 1e-6f, &kl, &g);
    } break;
    case VMF_KL_S1_I0I1: {
      // regular_cnn/s1_vae.py:83-88 with unscaled I0, I1: NaN once e^|k| overflows, as there.  1 - I1/I0 comes
      // from its own series on both intervals (for large k the two scaled functions agree to 1/(2k)), which
      // keeps the derivative k (1 - A)(1 + A) - A accurate where k - A - k A^2 would cancel.
      if (kappa != kappa) return false;
      const float a = ::fabsf(kappa);
      float e0, A, omA;
      if (a <= 8.0f) {
        const float t = 0.25f * a - 1.0f;
        e0 = cheb_sum_f32(I0E_A_F32, t);
        omA = cheb_sum_f32(I01R_A_F32, t);
        A = 1.0f - omA;
      } else {
        const float ra = fast_rcp(a);
        const float t = 16.0f * ra - 1.0f;
        const float B0 = cheb_sum_f32(I0E_B_F32, t);
        e0 = B0 * ::sqrtf(ra);
        omA = cheb_sum_f32(I01D_B_F32, t) * ra * fast_rcp(B0);
        A = 1.0f - omA;
      }
      const float i0 = e0 * fast_exp(a);
      if (i0 > 3.4028234e38f) A = omA = NAN;  // inf / inf in the reference
      kl = a * A - fast_log(i0);
      g = a * omA * (1.0f + A) - A;
      if (kappa < 0.0f) g = -g;
    } break;
    case VMF_KL_STABLE_M3: {
      const float k = kappa;
      if (k < 1.0f) {
        if (!(k > 0.0f)) return false;
        kl_m3_eps_fast(k, 0.0f, &kl, &g);  // the cancellation-free small-k form
      } else if (k > 3.0e38f) {  // +inf: the limit of log k + ln 2 - 1
        kl = k;
        g = 0.0f;
      } else {
        const float e2 = fast_exp(-2.0f * k);
        const float rom = fast_rcp(1.0f - e2);
        kl = kl_m3_stable_value(k);
        g = fast_rcp(k) - (k * e2) * (4.0f * rom * rom);
      }
    } break;
    case VMF_KL_STABLE_M2: {
      // k I1/I0 - log I0 without the exponential: with e0 = I0 e^-k and 1 - A from its own series (as above),
      // k A - log I0 = -k (1 - A) - log e0, finite for every k.  Below 1/4 the O(k) terms of that form cancel, and the
      // series k^2/4 - 3 k^4/64 + 5 k^6/576 - 77 k^8/49152 is exact to 2e-9 relative.
      if (kappa != kappa) return false;
      const float a = ::fabsf(kappa);
      if (a > 1e19f) {  // 1 / a leaves the normal range on the way to the two series below: the leading terms of
        kl = a > 3.0e38f ? a : kl_m2_value_f32(a);  // (log 2 pi k - 1) / 2 and of its slope 1 / 2k are exact to an ulp
        g = 0.5f * fast_rcp(a);
      } else if (a < 0.25f) {
        const float k2 = a * a;
        kl = k2 * (0.25f + k2 * (-3.0f / 64.0f + k2 * (5.0f / 576.0f + k2 * (-77.0f / 49152.0f))));
        g = a * (0.5f + k2 * (-3.0f / 16.0f + k2 * (5.0f / 96.0f + k2 * (-77.0f / 6144.0f))));
      } else {
        float e0, omA;
        if (a <= 8.0f) {
          const float t = 0.25f * a - 1.0f;
          e0 = cheb_sum_f32(I0E_A_F32, t);
          omA = cheb_sum_f32(I01R_A_F32, t);
        } else {
          const float ra = fast_rcp(a);
          const float t = 16.0f * ra - 1.0f;
          const float B0 = cheb_sum_f32(I0E_B_F32, t);
          e0 = B0 * ::sqrtf(ra);
          omA = cheb_sum_f32(I01D_B_F32, t) * ra * fast_rcp(B0);
        }
        const float A = 1.0f - omA;
        kl = -a * omA - fast_log(e0);
        g = a * omA * (1.0f + A) - A;
      }
      if (kappa < 0.0f) g = -g;
    } break;
    default:
      return false;
  }
  *kl_out = kl;
  *grad = g;
  return true;
}

// KL(vM(kappa) || U(S^1)) = kappa I1/I0 - log I0, the VALUE alone (what the fused sums and the per-element forward
// need), from polynomial fits of the function itself: one Horner chain of 12 (a <= 3) or 17 (a > 3) terms per element
// instead of the two Chebyshev series (20 + 29 or 8 + 11 terms at two instructions each) that the value-and-derivative
// form evaluates.  Two intervals, not more: a wave of mixed concentrations runs every chain, so the sum of the chain
// lengths is the cost (packed v_pk_fma_f32 pairs were tried for a branch-free form and issue at half rate here: slower).
template <int n>
VMF_HD float horner_f32(const float (&c)[n], float t) {
  float acc = c[n - 1];
#pragma unroll
  for (int k = n - 2; k >= 0; --k) acc = ::fmaf(acc, t, c[k]);
  return acc;
}
VMF_HD float kl_m2_value_f32(float a) {  // a = |kappa|, finite
  if (a <= KLM2_A0) {
    const float y = 0.25f * a * a;
    return y * horner_f32(KLM2_S, (8.0f / (KLM2_A0 * KLM2_A0)) * y - 1.0f);
  }
  const float ra = fast_rcp(a);
  return ::fmaf(0.5f * 0.69314718055994530942f, fast_log2(a), KLM2_C0) + horner_f32(KLM2_L, (2.0f * KLM2_A0) * ra - 1.0f);
}

// log C_2(k) = -log 2pi - log(I0(k) e^-k) - k for the circle, single precision, k > 0: the Chebyshev series of i0e
// (about twenty fused multiply-adds on [0,8], eight beyond) instead of the general-order power series / Debye sum,
// whose data-dependent trip count made the fused log_prob the larger half of the circle sampler's time.
VMF_HD float vmf_log_norm_circle_f32(float k) {
  float e0;
  if (k <= 8.0f) {
    e0 = cheb_sum_f32(I0E_A_F32, 0.25f * k - 1.0f);
  } else {
    const float rk = fast_rcp(k);
    e0 = cheb_sum_f32(I0E_B_F32, 16.0f * rk - 1.0f) * ::sqrtf(rk);
  }
  return -1.83787706640934548356f - fast_log(e0) - k;
}

// value + derivative for the kernels: the fast form for fp32 where there is one, the accurate form otherwise
template <typename T, bool BESSEL = true>
VMF_HD T kl_eval(T kappa, int variant, const KlBessel& kb, T* grad) {
  return kl_value<T, true>(kappa, variant, kb, grad);
}
template <>
VMF_HD float kl_eval<float, true>(float kappa, int variant, const KlBessel& kb, float* grad) {
  float kl;
  if (kl_value_fast(kappa, variant, &kl, grad)) return kl;
  return kl_value<float, true>(kappa, variant, kb, grad);
}
// fp32 tensors with 128-bit accesses (the launch code picks the instantiation)
template <>
VMF_HD float kl_eval<float, false>(float kappa, int variant, const KlBessel& kb, float* grad) {
  float kl;
  if (kl_value_fast(kappa, variant, &kl, grad)) return kl;
  return kl_value<float, false>(kappa, variant, kb, grad);
}

// value alone (forward kernels): the compiler drops the derivative of the forms above by itself; the two S^1 variants
// take their own value-only fits.  VMF_KL_S1_I0I1 keeps the reference's overflow behaviour (NaN once e^kappa leaves the
// float range, kappa > 88.7228): the fit below 88.70, NaN from 88.75 on, and the value-and-derivative form -- which
// forms e^kappa and so decides the overflow itself -- only in the window between them.  (With that form for everything
// beyond 85, a batch with a third of its concentrations past the overflow point ran every wave through all three
// chains: 0.37 ms per 2^27 against 0.17 for concentrations below 80.)
template <typename T, bool BESSEL>
VMF_HD T kl_eval_value(T kappa, int variant, const KlBessel& kb) {
  T g;
  return kl_eval<T, BESSEL>(kappa, variant, kb, &g);
}
#define VMF_KL_VALUE_F32(BESSEL)                                                                              \
  template <>                                                                                                 \
  VMF_HD float kl_eval_value<float, BESSEL>(float kappa, int variant, const KlBessel& kb) {                   \
    const float a = ::fabsf(kappa);                                                                           \
    if ((variant == VMF_KL_STABLE_M2 && a <= 3.0e38f) || (variant == VMF_KL_S1_I0I1 && a <= 88.70f))          \
      return kl_m2_value_f32(a);                                                                              \
    if (variant == VMF_KL_S1_I0I1 && a >= 88.75f) return NAN; /* e^a = inf: inf / inf in the reference */     \
    float g;                                                                                                  \
    return kl_eval<float, BESSEL>(kappa, variant, kb, &g);                                                    \
  }
VMF_KL_VALUE_F32(true)
VMF_KL_VALUE_F32(false)
#undef VMF_KL_VALUE_F32

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
// The composition needs 1 - w^2 = (1-w)(1+w), and 1 + w = (1-c)/den loses its digits when formed from the rounded w
// near w = -1 (m = 3 has density there: one row in ~10^6 at kappa = 1 came out 1e-4 off in fp32).  With q = U^(2/(m-2)),
// r = sqrt(1-q) and V folded to its distance from the nearest integer,
//   1 - c = (1 - r) + r (1 - cos 2 pi V) = q / (1 + r) + 2 r sin^2(pi V)
// is a sum of two non-negative terms, so *c2 = (1-w)(1+w) keeps its relative accuracy everywhere (used for m <= 4:
// the density at w = -1 is ~ (1+w)^((m-3)/2), so wider rows never draw a w that close and take the plain 1 - c).
template <typename T>
__device__ __forceinline__ bool wood_try_counter(T b, int m, T U, T V, T uacc, T* w, T* omw, T* c2) {
  T sq = (T)1, q = (T)0;
  if (m == 4) {  // U^(2/(m-2)) = U: no logarithm, and 1 - U is exact
    q = U;
    sq = (T)1 - U;
  } else if (m == 3) {  // U^2
    q = U * U;
    sq = ((T)1 - U) * ((T)1 + U);
  } else if (m > 2) {
    const T t = (T)2 / (T)(m - 2) * F<T>::log2(U);
    sq = F<T>::one_minus_exp2(t);  // 1 - U^(2/(m-2))
    q = F<T>::exp2(t);
  }
  const T rt = F<T>::sqrt(sq);
  const T c = rt * F<T>::cos2pi(V);
  const T opc = (T)1 + c;
  T omc = (T)1 - c, opa = opc;
  if (m <= 4) {  // from m = 5 on the density vanishes at w = -1 and at w = 1 fast enough that no draw comes near them
    const T Vf = V > (T)0.5 ? (T)1 - V : V;  // sin^2 pi V = sin^2 pi Vf, from a small argument at both ends
    const T sh = F<T>::sin2pi((T)0.5 * Vf);
    const T r1 = q * F<T>::rcp((T)1 + rt);  // 1 - r
    omc = r1 + (T)2 * rt * sh * sh;
    // and the mirror image for the other pole: 1 + c = (1 - r) + 2 r cos^2(pi V), the cosine as the sine of the exact
    // difference 1/4 - Vf/2 (revolutions).  It feeds the outputs only (1 - w, 1 - w^2: a w within 1e-7 of +1 keeps the
    // digits of its tangent component); the accept test below takes the plain 1 + c, whose absolute error is all it sees.
    const T chh = F<T>::sin2pi((T)0.25 - (T)0.5 * Vf);
    opa = r1 + (T)2 * rt * chh * chh;
  }
  const T den = (T)1 - (T)0.5 * ((T)1 - b) * opc;
  const T rden = F<T>::rcp(den);
  *omw = b * opa * rden;
  *w = (T)1 - *omw;
  *c2 = *omw * (omc * rden);
  const T s = (T)0.5 * ((T)1 - b) * c * rden;
  return (T)(m - 1) * F<T>::log1p_minus_over_ln2(s) >= F<T>::log2(uacc);  // both sides divided by ln 2
}

// ---- circle (m = 2), counter noise ------------------------------------------------------------------------------
// Noise layout (shared with the CPU checker): the FIRST proposal of rows 2p and 2p+1 comes from ONE Philox call
//   Philox(index = p = row >> 1, slot 0): words (x, y) serve the even row, (z, w) the odd one -- the first word gives
//   the proposal's angle V = u01(word) and, in its bit 0 (below the 23 bits u01 uses), the sign of the tangent; the
//   second word is the accept draw.
// A row whose first proposal is rejected (about a fifth of them for kappa >= 10) continues on calls of its own:
//   Philox(index = row, slot 1 + t), t = 0, 1, ...: proposals 1 + 2t from (x, y) and 2 + 2t from (z, w).
// So a pair of rows costs one call, plus one per further two proposals of a rejected row.
//
// The proposal cosine c = cos 2 pi V follows the arcsine law, half its mass near -1 and +1, where 1 + c, 1 - c and
// the denominator cancel in turn.  The half angle gives every factor without cancellation:
//   1 + c = 2 cos^2(pi V),  1 - c = 2 sin^2(pi V),  den = 1 - (1-b)(1+c)/2 = b cos^2 + sin^2,
//   1 - w = 2 b cos^2 / den,  1 + w = 2 sin^2 / den
// and sqrt(1 - w^2) is taken from the product of the two (an accepted w within 1e-6 of -1 is rare but happens at
// kappa ~ 1, and 1 + w formed from the rounded w loses its digits there: 1e-4 in z).
//
// Two kernels draw circle rows (two rows per lane for dense aligned tensors, one row per lane for every other layout)
// and a shard must reproduce the unsharded rows bit for bit whichever of them it lands in: the functions below fix
// every rounding (no contraction left to the compiler) and both kernels call them.
__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double vfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// u01(x) / 2, exact: the angle pi V as the argument, in revolutions, of the hardware sine / cosine
template <typename T>
__device__ __forceinline__ T u01_half(uint32_t x) { return (T)0.5 * u01<T>(x); }
template <>
__device__ __forceinline__ float u01_half<float>(uint32_t x) {
  // the 23 bits under the exponent of 1/2 give 1/2 + (x >> 9) 2^-24 in [1/2, 1); subtracting 1/2 - 2^-25 is exact
  return __uint_as_float(__builtin_amdgcn_alignbit(0x7Eu, x, 9u)) - 0.49999997f;
}

// b(kappa) = 1 / (2 kappa + sqrt(4 kappa^2 + 1))
template <typename T>
__device__ __forceinline__ T circle_b(T kappa) {
  return F<T>::rcp(vfma((T)2, kappa, F<T>::sqrt(vfma((T)4 * kappa, kappa, (T)1))));
}

// One proposal from the words (xv, xu); hb = (1 - b) / 2.  Returns cos^2 / den and sin^2 / den; the caller forms
// 1 - w = 2 b pc and 1 + w = 2 ps for the proposal it keeps.
template <typename T>
__device__ __forceinline__ bool circle_try(T b, T hb, uint32_t xv, uint32_t xu, T* pc, T* ps) {
#pragma clang fp contract(off)
  T sh, ch;
  F<T>::sincos2pi(u01_half<T>(xv), &sh, &ch);
  const T ch2 = ch * ch, sh2 = sh * sh;
  const T rden = F<T>::rcp(vfma(b, ch2, sh2));
  *pc = ch2 * rden;
  *ps = sh2 * rden;
  const T s = hb * (*pc - *ps);  // (1-b) c / (2 den)
  // m - 1 = 1, so log(1+s) - s >= ln u can be tested as 1 + s >= u e^s: one exponential instead of two logarithms
  return (T)1 + s >= u01<T>(xu) * F<T>::exp2((T)1.44269504088896340736 * s);
}
template <typename T>
__device__ __forceinline__ T circle_hb(T b) { return vfma((T)-0.5, b, (T)0.5); }

// The accepted proposal as the composition needs it: w, 1 - w (exact where w is close to 1) and the signed
// sqrt(1 - w^2) = the tangent component of z' (the tangent "direction" on a circle is one sign).
template <typename T>
__device__ __forceinline__ void circle_finish(T b, T pc, T ps, bool neg, T* w, T* omw, T* fs) {
#pragma clang fp contract(off)
  const T om = ((T)2 * b) * pc, op = (T)2 * ps;
  T c2 = om * op;
  c2 = c2 < (T)1e-10 ? (T)1e-10 : c2;
  const T sc = F<T>::sqrt(c2);
  const bool hi = om > (T)1;  // w < 0: 1 + w is the small, exactly known one
  *w = hi ? op - (T)1 : (T)1 - om;
  *omw = hi ? (T)2 - op : om;
  *fs = neg ? -sc : sc;
}

// z = H (w, fs) with H the Householder reflection e1 -> mu (mu / ||mu|| when `normalize`), and <mu, z> for log p:
//   z = z' - 2 <z', d> d / <d, d>,  d = e1 - mu
// <mu, z> = <mu, z'> - coef <mu, d> with <mu, d> = mu0 (1 - mu0) - mu1^2: written through the two small quantities,
// because coef ~ 1 / |d|^2 multiplies it and mu0 - |mu|^2 formed from rounded values near 1 loses everything when mu is
// close to the pole (for one row in a few hundred it is).
template <typename T>
__device__ __forceinline__ void circle_compose(T mu0, T mu1, T w, T fs, bool normalize, T* z0, T* z1, T* mz) {
#pragma clang fp contract(off)
  T inv = (T)1;
  if (normalize) inv = F<T>::rsqrt(vfma(mu1, mu1, mu0 * mu0));
  const T m0 = mu0 * inv, m1 = mu1 * inv;
  const T d0 = (T)1 - m0;
  const T tail = m1 * m1;
  const T dd = vfma(d0, d0, tail);  // |e1 - mu|^2
  const T fm = fs * m1;
  const T zd = vfma(w, d0, -fm);    // <z', d>
  const T coef = dd > (T)1e-30 ? ((T)2 * zd) * F<T>::rcp(dd) : (T)0;
  *z0 = vfma(-coef, d0, w);
  *z1 = vfma(coef, m1, fs);
  *mz = vfma(w, m0, fm) - coef * vfma(m0, d0, -tail);
}

// Backward of one circle row: (d0, d1) = J^T G with respect to mu (raw mu when `normalize`) and *dk = dz/dkappa . G
// through w(kappa).  mu = (m0, m1), G = grad z, neg = the row's tangent sign, (w, omw = 1 - w) as the forward left them.
// The general backward kernel specialised to m = 2 (tangent noise g = (0, +-1), |g| = 1); both kernels that take
// circle rows call it, so a row's gradient does not depend on where a shard's rows start.
template <typename T>
__device__ __forceinline__ void circle_bwd_row(T mu0, T mu1, T G0, T G1, bool neg, T w, T omw, T kap, bool normalize,
                                               T* d0_out, T* d1_out, T* dk) {
#pragma clang fp contract(off)
  T inv = (T)1;
  if (normalize) inv = F<T>::rsqrt(vfma(mu1, mu1, mu0 * mu0));
  const T m0 = mu0 * inv, mh = mu1 * inv;
  const T sg = neg ? (T)-1 : (T)1;
  const T gm = sg * mh;                        // <g, mu_hat>
  const T d0 = (T)1 - m0;
  const T Gd = vfma(G0, d0, -(G1 * mh));       // <G, e1 - mu_hat>
  const T dd = vfma(d0, d0, mh * mh);          // |e1 - mu_hat|^2
  T c2 = omw * ((T)1 + w);
  const bool inside = c2 >= (T)1e-10;
  c2 = inside ? c2 : (T)1e-10;
  const T sq = F<T>::sqrt(c2);                 // f = sqrt(1 - w^2) / |g|
  const T zd = vfma(w, d0, -(sq * gm));        // <z', d>
  const T inv_dd = dd > (T)1e-30 ? F<T>::rcp(dd) : (T)0;
  const T t = ((T)2 * zd) * (Gd * inv_dd);     // 2 zd Gd / |d|^2
  T o0 = (T)2 * (vfma(w, Gd, vfma(zd, G0, -(t * d0)))) * inv_dd;
  T o1 = (T)2 * (vfma(sq * sg, Gd, vfma(zd, G1, t * mh))) * inv_dd;
  if (normalize) {  // d mu = (d mu_hat - mu_hat <mu_hat, d mu_hat>) / ||mu||
    const T dot = vfma(o0, m0, o1 * mh);
    o0 = vfma(-m0, dot, o0) * inv;
    o1 = vfma(-mh, dot, o1) * inv;
  }
  *d0_out = o0;
  *d1_out = o1;
  const T g2 = ((T)2 * Gd) * inv_dd;
  const T Gp0 = vfma(-g2, d0, G0);
  const T S = vfma(g2, gm, G1 * sg);
  const T dw = Gp0 + (inside ? S * (-w * F<T>::rcp(sq)) : (T)0);
  const T c = F<T>::sqrt(vfma((T)4 * kap, kap, (T)1));
  const T b = F<T>::rcp(vfma((T)2, kap, c));
  const T e = omw * F<T>::rcp(((T)1 + b) - w * ((T)1 - b));
  const T den = (T)1 - ((T)1 - b) * e;
  const T dw_db = ((T)-2 * e) * ((T)1 - e) * F<T>::rcp(den * den);
  *dk = dw * dw_db * (((T)-2 * b) * F<T>::rcp(c));
}

// b(kappa) alone (what the counter-based path needs)
template <typename T>
__device__ __forceinline__ T wood_b(T kappa, int m) {
  const T m1 = (T)(m - 1);
  return m1 * F<T>::rcp((T)2 * kappa + F<T>::sqrt((T)4 * kappa * kappa + m1 * m1));
}
