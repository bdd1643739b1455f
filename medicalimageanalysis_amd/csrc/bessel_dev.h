// bessel_dev.h -- exponentially scaled modified Bessel function I_v on the device (double arithmetic).
//
// Three regimes, chosen per element:
//   power series            v^2 + x^2 <  DEBYE_R2   (x < 30: at most ~80 positive terms)
//   uniform (Debye) expansion v^2 + x^2 >= DEBYE_R2  (13 terms; truncation error ~1e-15 at the switch)
//   continued fraction      for the ratio I_v/I_{v-1} in the series region (modified Lentz)
// The order v is one scalar per launch, so everything that depends on v alone (lgamma) is computed
// on the host and carried in BesselOrder.
#pragma once
#include <math.h>

#include "common.h"
#include "debye_coeffs.h"
#include "i0e_coeffs.h"

#define VMF_DEBYE_R2 900.0
#define VMF_AMOS_LIMIT 1073741823.5

struct BesselOrder {
  double v;            // order as given
  double av;           // |v| used for the large-x regime
  double lgam_v1;      // lgamma(v + 1) (for v > -1)
  double lgam_av1;     // lgamma(|v| + 1)
  int is_int;          // v is an integer
  int odd;             // integer and odd
  int neg_frac;        // -1 < v < 0 (non integer)
  int unsupported;     // non-integer v <= -1
  int amos;            // apply AMOS's |z| <= 2^30 limit (scipy.special.ive); 0 for i0e / i1e
  int half;            // closed form available: 1 -> v = 1/2, 2 -> v = -1/2, 3 -> v = 3/2; 0 otherwise
};

static inline BesselOrder make_bessel_order(double v, int amos) {
  BesselOrder o;
  o.v = v;
  o.amos = amos;
  o.is_int = (v == floor(v));
  o.neg_frac = (!o.is_int && v < 0 && v > -1.0);
  o.unsupported = (!o.is_int && v <= -1.0);
  if (o.is_int && v < 0) v = -v;  // I_{-n} = I_n
  o.av = fabs(v);
  o.v = o.is_int ? v : o.v;
  o.odd = o.is_int && fmod(v, 2.0) != 0.0;
  o.lgam_v1 = (o.v > -1.0) ? lgamma(o.v + 1.0) : 0.0;
  o.lgam_av1 = lgamma(o.av + 1.0);
  o.half = (o.v == 0.5) ? 1 : (o.v == -0.5) ? 2 : (o.v == 1.5) ? 3 : 0;
  return o;
}

// log of  sum_k (x^2/4)^k / (k! (v+1)_k)
VMF_HD double bessel_series_logsum(double v, double x) {
  double q = 0.25 * x * x, term = 1.0, sum = 1.0;
  for (int k = 1; k < 400; ++k) {
    term *= q / ((double)k * (v + (double)k));
    sum += term;
    if (term < 1e-17 * sum) break;
  }
  return ::log(sum);
}

// log ive(v,x) by the uniform expansion; v >= 0 (for v = 0 it is the Hankel series in 1/x)
VMF_HD double log_ive_debye(double v, double x) {
  double r = x > 1e150 ? x : ::sqrt(v * v + x * x);  // x*x would overflow
  double p = v / r;
  double rmx = v * v / (r + x);                 // r - x without cancellation
  double expo = rmx - v * ::log1p((v + rmx) / x);  // v*eta - x
  double p2 = p * p, pv = 1.0 / r, pk = 1.0, sum = 0.0;  // p/v = 1/r
  // rolled loops on purpose: the indices are uniform, so the 91 coefficients arrive through scalar loads; unrolled,
  // they become 182 vector registers of immediates and the elementwise kernels drop to one wave per SIMD
  // Term k is bounded by max|u_k| / r^k (max|u_k| = 1, 0.125, 0.07, 0.073, 0.11, 0.23, 0.57, 1.7, 6.1, 24, 110, 551,
  // 3040): with nk = ceil(57 / log2(r/2)) terms everything left out adds up to < 5e-18 of the sum, for every r
  // (checked against the coefficient table on a dense grid of r).  At r ~ 1e4 that is 5 terms instead of 13, and the
  // late terms are the long polynomials.
  int nk = DEBYE_K;
#ifdef __HIP_DEVICE_COMPILE__
  // One trip count for the wave -- the largest any active lane needs -- so that the loop below stays a scalar loop.
  // A wave with any lane at r <= 64 runs all terms and skips this bookkeeping (one compare); otherwise the wave
  // maximum is found by bisection on ballots.
  if (__ballot(r <= 64.0) == 0ull) {
    const int need = (int)::ceilf(57.0f * fast_rcp(fast_log2(0.5f * (float)(r < 1e30 ? r : 1e30))));
    int lo = 1, hi = DEBYE_K;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (__ballot(need > mid) != 0ull) lo = mid + 1;
      else hi = mid;
    }
    nk = lo;
  }
#else
  if (r > 64.0) {
    const int need = (int)::ceilf(57.0f / ::log2f(0.5f * (float)(r < 1e30 ? r : 1e30)));
    nk = need < 1 ? 1 : (need < DEBYE_K ? need : DEBYE_K);
  }
#endif
#pragma unroll 1
  for (int k = 0; k < nk; ++k) {
    double poly = 0.0;
#pragma unroll 1
    for (int j = k; j >= 0; --j) poly = poly * p2 + DEBYE_U[k][j];
    sum += pk * poly;
    pk *= pv;
  }
  // (2 pi v)^-1/2 (1+z^2)^-1/4 = (2 pi r)^-1/2
  return expo - 0.5 * ::log(2.0 * M_PI * r) + ::log(sum);
}

// log ive(v, x), x > 0, v > -1
VMF_HD double log_ive_pos(double v, double lgam_v1, double x) {
  if (v * v + x * x < VMF_DEBYE_R2) return v * ::log(0.5 * x) - lgam_v1 - x + bessel_series_logsum(v, x);
  return log_ive_debye(v, x);
}

// ive(v, z) with scipy's edge conventions (see include/vmf_hip.h)
// Orders 0 and 1: Chebyshev expansions on [0, 8] and [8, inf) (tables in i0e_coeffs.h), ~30 fused multiply-adds each
// instead of a power series of up to 60 terms.  Clenshaw recurrence; x >= 0.
VMF_HD double cheb_sum(const double* c, int n, double t) {
  double b0 = 0.0, b1 = 0.0;
  const double t2 = 2.0 * t;
  // eight terms per trip: the coefficients of a trip arrive in one batch of scalar loads instead of one dependent load
  // per term (ive(0) / ive(1): 0.22 -> 0.17 ms per 2^24; fully unrolled they would sit in vector registers)
#pragma unroll 8
  for (int k = n - 1; k >= 1; --k) {
    const double b = c[k] + t2 * b0 - b1;
    b1 = b0;
    b0 = b;
  }
  return c[0] + t * b0 - b1;
}
VMF_HD double i0e_pos(double x) {
  if (x <= 8.0) return cheb_sum(I0E_A, I0E_A_N, 0.25 * x - 1.0);
  return cheb_sum(I0E_B, I0E_B_N, 16.0 / x - 1.0) / ::sqrt(x);
}
VMF_HD double i1e_pos(double x) {
  if (x <= 8.0) return x * cheb_sum(I1E_A, I1E_A_N, 0.25 * x - 1.0);
  return cheb_sum(I1E_B, I1E_B_N, 16.0 / x - 1.0) / ::sqrt(x);
}

// Half-integer orders the reference's mean() and ive's backward use (sampler/von_mises_fisher.py:38-43 calls
// ive(3/2, k) / ive(1/2, k); the derivative of ive(1/2, .) needs ive(-1/2, .)): elementary closed forms,
//   ive(1/2, x) = sqrt(2/(pi x)) (1 - e^{-2x}) / 2,   ive(-1/2, x) = sqrt(2/(pi x)) (1 + e^{-2x}) / 2,
//   ive(3/2, x) = sqrt(2/(pi x)) [(1 + e^{-2x}) / 2 - (1 - e^{-2x}) / (2x)]   (x >= 1; below, the two terms cancel
//   and the power series, a handful of terms there, is used instead).
VMF_HD double ive_half_closed(int half, double x) {
  const double pref = ::sqrt(0.63661977236758134308 / x);  // sqrt(2 / (pi x))
  const double em = -::expm1(-2.0 * x);                     // 1 - e^{-2x}
  if (half == 1) return pref * 0.5 * em;
  const double ep = 2.0 - em;                               // 1 + e^{-2x}
  if (half == 2) return pref * 0.5 * ep;
  return pref * (0.5 * ep - 0.5 * em / x);
}

// ive(1/2, x) and ive(3/2, x) together for finite x > 0: one square root and one exponential between them.  Below
// x = 1, where the two terms of the order 3/2 cancel, cosh x - sinh(x)/x = sum_{k>=1} 2k x^(2k) / (2k+1)!  (ten terms:
// the next one is below 1e-19 of the sum).
VMF_HD void ive_half_pair(double x, double* i12, double* i32) {
  const double pref = ::sqrt(0.63661977236758134308 / x);  // sqrt(2 / (pi x))
  const double em = -::expm1(-2.0 * x);                     // 1 - e^{-2x}
  *i12 = pref * 0.5 * em;
  if (x >= 1.0) {
    *i32 = pref * (0.5 * (2.0 - em) - 0.5 * em / x);
    return;
  }
  const double q = x * x;
  double s = 1.0 / 2554547108585472000.0;
  s = s * q + 1.0 / 6758061133824000.0;
  s = s * q + 1.0 / 22230464256000.0;
  s = s * q + 1.0 / 93405312000.0;
  s = s * q + 1.0 / 518918400.0;
  s = s * q + 1.0 / 3991680.0;
  s = s * q + 1.0 / 45360.0;
  s = s * q + 1.0 / 840.0;
  s = s * q + 1.0 / 30.0;
  s = s * q + 1.0 / 3.0;
  *i32 = pref * ::exp(-x) * (q * s);
}

VMF_HD double ive_eval(const BesselOrder& o, double z) {
  if (z != z) return z;
  if (o.unsupported) return NAN;
  double x = z, sign = 1.0, v = o.v;
  if (x < 0) {
    if (!o.is_int) return NAN;
    if (o.odd) sign = -1.0;
    x = -x;
  }
  if (x == 0) return o.is_int ? (v == 0 ? 1.0 : 0.0) : (v < 0 ? NAN : 0.0);
  if (o.amos && x > VMF_AMOS_LIMIT) return NAN;
  if (isinf(x)) return 0.0;
  if (o.is_int && v <= 1.0) return sign * (v == 0.0 ? i0e_pos(x) : i1e_pos(x));
  if (o.half && (o.half != 3 || x >= 1.0)) return ive_half_closed(o.half, x);
  if (o.neg_frac && x > 20.0) {
    // I_v = I_{-v} + (2/pi) sin(-v pi) K_{-v}; the K term is < 2 e^{-2x} relative: below one ulp here
    return ::exp(log_ive_pos(o.av, o.lgam_av1, x));
  }
  return sign * ::exp(log_ive_pos(v, o.lgam_v1, x));
}

// ---- ive for a given order with per-launch coefficient tables -----------------------------------------------------
// The order is one scalar per launch, so both expansions become ONE polynomial each whose coefficients the host works
// out once per call and the kernel reads through scalar loads:
//   series   ive = exp(v log(x/2) - lgamma(v+1) - x) sum_k a_k q^k,  q = x^2/4,  a_k = 1 / (k! (v+1)_k): no division
//            per term; 48 terms cover x < 30 (the region v^2 + x^2 < 900), 24 cover x <= 8, all terms positive
//   uniform  sum_k u_k(p) / v^k with p = v/r, r = sqrt(v^2 + x^2): u_k(p) = p^k sum_j U[k][j] p^(2j), and p^2 = v^2 t^2
//            with t = 1/r, so the double sum is the polynomial sum_n d_n t^n, d_n = sum_{k+2j=n} U[k][j] v^(2j), of
//            degree 36 -- 37 fused multiply-adds instead of thirteen nested polynomial loops; and
//            ive = exp((r - x) - v log((v + r)/x)) * sum / sqrt(2 pi r) needs one logarithm and one exponential
//            (the log-domain form above takes three logarithms).
// Accuracy against scipy.special.ive (tests/fuzz_ive.py): 1e-12 relative; the error is that of v log(.) in the
// exponent, as in the log-domain form.
#define VMF_IVE_NS 48
#define VMF_IVE_NS_SHORT 24
#define VMF_IVE_ND (3 * (DEBYE_K - 1) + 1)
struct IveTab {
  double ser[VMF_IVE_NS + 1];
  double deb[VMF_IVE_ND];
};
static inline IveTab make_ive_tab(const BesselOrder& o) {
  IveTab t;
  // series coefficients for the order the series runs on (v itself; -1 < v allowed)
  const double vs = o.v;
  t.ser[0] = 1.0;
  for (int k = 1; k <= VMF_IVE_NS; ++k) t.ser[k] = (vs > -1.0) ? t.ser[k - 1] / ((double)k * (vs + (double)k)) : 0.0;
  // uniform expansion for |v| (negative fractional orders use it beyond x = 20, see ive_eval)
  const double v2 = o.av * o.av;
  for (int n = 0; n < VMF_IVE_ND; ++n) t.deb[n] = 0.0;
  for (int k = 0; k < DEBYE_K; ++k) {
    double vp = 1.0;
    for (int j = 0; j <= k; ++j) {
      t.deb[k + 2 * j] += DEBYE_U[k][j] * vp;
      vp *= v2;
    }
  }
  return t;
}
// 1/a and 1/sqrt(a) to an ulp or so: the hardware estimate and two Newton steps (a few fused multiply-adds; the IEEE
// division / square-root sequences are 15-25 instructions each)
VMF_HD double rcp_nr(double a) {
#ifdef __HIP_DEVICE_COMPILE__
  double y = __builtin_amdgcn_rcp(a);
  y = ::fma(y, ::fma(-a, y, 1.0), y);
  y = ::fma(y, ::fma(-a, y, 1.0), y);
  return y;
#else
  return 1.0 / a;
#endif
}
VMF_HD double rsqrt_nr(double a) {
#ifdef __HIP_DEVICE_COMPILE__
  double y = __builtin_amdgcn_rsq(a);
  y = ::fma(0.5 * y, ::fma(-a * y, y, 1.0), y);
  y = ::fma(0.5 * y, ::fma(-a * y, y, 1.0), y);
  return y;
#else
  return 1.0 / ::sqrt(a);
#endif
}
// sum_{k<n} c[k] t^k; eight coefficients per trip arrive in one batch of scalar loads
VMF_HD double horner_tab(const double* c, int n, double t) {
  double acc = c[n - 1];
#pragma unroll 8
  for (int k = n - 2; k >= 0; --k) acc = ::fma(acc, t, c[k]);
  return acc;
}
// Which evaluator a launch needs, decided on the host from the order alone (the kernels are instantiated per kind, so
// that an order pays for its own code path only):
//   CHEB     orders 0 and 1 (and -1 = 1): the Chebyshev expansions
//   HALF     1/2, -1/2, 3/2: closed forms (3/2 below x = 1: the tabulated series)
//   GENERIC  any other order v > -1 that is an integer or positive: tabulated series / uniform expansion
//   SLOW     everything else (negative non-integer orders): ive_eval
enum { IVE_KIND_CHEB = 0, IVE_KIND_HALF = 1, IVE_KIND_GENERIC = 2, IVE_KIND_SLOW = 3 };
static inline int ive_kind_of(const BesselOrder& o) {
  if (o.unsupported || o.neg_frac) return o.half ? IVE_KIND_HALF : IVE_KIND_SLOW;
  if (o.is_int && o.v <= 1.0) return IVE_KIND_CHEB;
  if (o.half) return IVE_KIND_HALF;
  return IVE_KIND_GENERIC;
}
// ive(v, x), x > 0 finite with x <= 2^30, v > -1 the order of `tab`: the tabulated series where v^2 + x^2 < 900, the
// uniform expansion beyond.  A wave that holds both regions runs both polynomials but ONE logarithm and ONE
// exponential: ive = exp(E) P with
//   series   P = sum_k a_k q^k,                  E = v log(x/2) - lgamma(v+1) - x
//   uniform  P = sum_n d_n t^n sqrt(t / 2 pi),   E = (r - x) - v log((v + r)/x)
// (E, P) of ive = exp(E) P: what ive itself, its logarithm and ratios of two orders are made of
VMF_HD void ive_pos_tab_parts(const IveTab& tab, double v, double lgam_v1, double x, double* E_out, double* P_out) {
  const double r2 = ::fma(x, x, v * v);
  double P, larg, e0;
  if (r2 < VMF_DEBYE_R2) {
    const double q = 0.25 * x * x;
    int n = VMF_IVE_NS + 1;
#ifdef __HIP_DEVICE_COMPILE__
    if (__ballot(x > 8.0) == 0ull) n = VMF_IVE_NS_SHORT + 1;  // one trip count for the lanes in here
#else
    if (x <= 8.0) n = VMF_IVE_NS_SHORT + 1;
#endif
    P = horner_tab(tab.ser, n, q);
    larg = 0.5 * x;
    e0 = -lgam_v1 - x;
  } else {
    const double t = rsqrt_nr(r2);  // 1 / r
    const double r = r2 * t;
    P = horner_tab(tab.deb, VMF_IVE_ND, t) * ::sqrt(0.15915494309189533577 * t);  // (2 pi r)^-1/2 = sqrt(t / 2 pi)
    larg = (v + r) * rcp_nr(x);
    e0 = v * v * rcp_nr(r + x);  // r - x without cancellation
  }
  const double L = ::log(larg);
  *E_out = r2 < VMF_DEBYE_R2 ? ::fma(v, L, e0) : ::fma(-v, L, e0);
  *P_out = P;
}
VMF_HD double ive_pos_tab(const IveTab& tab, double v, double lgam_v1, double x) {
  double E, P;
  ive_pos_tab_parts(tab, v, lgam_v1, x, &E, &P);
  return ::exp(E) * P;
}
// The log-domain quantities from the same tables (x > 0; beyond VMF_IVE_TAB_XMAX the rolled expansions further up take
// over: r^2 would leave the range the Newton steps of the reciprocal root are good for):
//   log ive(v, x) = E + log P                       two logarithms, no loop over terms
//   I_v / I_{v-1} = exp(E_v - E_{v-1}) P_v / P_{v-1}   (tables of both orders; each order picks its own region)
// against a power series of up to 400 terms / thirteen nested polynomial loops with three logarithms, and a continued
// fraction of up to 500 divisions.  Accuracy as ive's: the error of v log(.) in the exponent (1e-13 at v = 255).
#define VMF_IVE_TAB_XMAX 1.0e9
VMF_HD double log_ive_pos_tab(const IveTab& tab, double v, double lgam_v1, double x) {
  if (!(x <= VMF_IVE_TAB_XMAX)) return log_ive_pos(v, lgam_v1, x);
  double E, P;
  ive_pos_tab_parts(tab, v, lgam_v1, x, &E, &P);
  return E + ::log(P);
}
// ive(v, z) with scipy's edge conventions (as ive_eval) for a launch of the given kind
template <int KIND>
VMF_HD double ive_eval_kind(const BesselOrder& o, const IveTab& tab, double z) {
  if (KIND == IVE_KIND_SLOW) return ive_eval(o, z);
  if (z != z) return z;
  double x = z, sign = 1.0;
  const double v = o.v;
  if (x < 0) {
    if (!o.is_int) return NAN;
    if (o.odd) sign = -1.0;
    x = -x;
  }
  if (x == 0) return o.is_int ? (v == 0 ? 1.0 : 0.0) : (v < 0 ? NAN : 0.0);
  if (o.amos && x > VMF_AMOS_LIMIT) return NAN;
  if (isinf(x)) return 0.0;
  if (KIND == IVE_KIND_CHEB) return sign * (v == 0.0 ? i0e_pos(x) : i1e_pos(x));
  if (KIND == IVE_KIND_HALF) {
    if (o.half != 3 || x >= 1.0) return ive_half_closed(o.half, x);
    return ive_pos_tab(tab, v, o.lgam_v1, x);
  }
  return sign * ive_pos_tab(tab, v, o.lgam_v1, x);
}
// run-time dispatch over the kinds (host-side checks)
VMF_HD double ive_eval_tab(const BesselOrder& o, const IveTab& tab, int kind, double z) {
  switch (kind) {
    case IVE_KIND_CHEB: return ive_eval_kind<IVE_KIND_CHEB>(o, tab, z);
    case IVE_KIND_HALF: return ive_eval_kind<IVE_KIND_HALF>(o, tab, z);
    case IVE_KIND_GENERIC: return ive_eval_kind<IVE_KIND_GENERIC>(o, tab, z);
    default: return ive_eval(o, z);
  }
}

// I_v(x)/I_{v-1}(x), v >= 1/2 ... any v with v-1 > -1, x > 0.  lg_v1 = lgamma(v+1), lg_v = lgamma(v).
VMF_HD double bessel_ratio_eval(double v, double lg_v1, double lg_v, double x) {
  if (x == 0) return 0.0;
  if (v * v + x * x < VMF_DEBYE_R2 && (v - 1.0) * (v - 1.0) + x * x < VMF_DEBYE_R2) {
    // continued fraction  I_v/I_{v-1} = 1/(2v/x + 1/(2(v+1)/x + ...)), modified Lentz
    const double tiny = 1e-300;
    double f = tiny, C = f, D = 0.0;
    for (int k = 0; k < 500; ++k) {
      double bk = 2.0 * (v + k) / x;
      D = bk + D;  // a_k = 1
      if (D == 0.0) D = tiny;
      C = bk + 1.0 / C;
      if (C == 0.0) C = tiny;
      D = 1.0 / D;
      double delta = C * D;
      f *= delta;
      if (::fabs(delta - 1.0) < 1e-16) break;
    }
    return f;
  }
  return ::exp(log_ive_pos(v, lg_v1, x) - log_ive_pos(v - 1.0, lg_v, x));
}

// the same ratio from the tables of the orders v (tab1) and v - 1 (tab0)
VMF_HD double bessel_ratio_tab(const IveTab& tab1, const IveTab& tab0, double v, double lg_v1, double lg_v, double x) {
  if (x == 0) return 0.0;
  if (!(x <= VMF_IVE_TAB_XMAX)) return bessel_ratio_eval(v, lg_v1, lg_v, x);
  double E1, P1, E0, P0;
  ive_pos_tab_parts(tab1, v, lg_v1, x, &E1, &P1);
  ive_pos_tab_parts(tab0, v - 1.0, lg_v, x, &E0, &P0);
  return ::exp(E1 - E0) * P1 * rcp_nr(P0);
}

// ---- single-precision log ive for the fp32 log-normaliser (one evaluation per ROW of a 2^24-row batch) ---------
// 7 Debye terms once v^2 + x^2 >= 100 (truncation < 2e-7 relative), the power series below that (x < 10: <= 35
// terms, all positive).  Every quantity that cancels (r - x, log(x/(v+r))) is formed the same way as in double.
#define VMF_DEBYE_R2_F32 100.0f
#define VMF_DEBYE_K_F32 7
VMF_HD float log_ive_pos_f32(float v, float lgam_v1, float x) {
  if (v * v + x * x < VMF_DEBYE_R2_F32) {
    const float q = 0.25f * x * x;
    float term = 1.0f, sum = 1.0f;
    for (int k = 1; k < 64; ++k) {
      term *= q * fast_rcp((float)k * (v + (float)k));  // hardware reciprocal: 1 ulp per term, all terms positive
      sum += term;
      if (term < 1e-8f * sum) break;
    }
    return v * fast_log(0.5f * x) - lgam_v1 - x + fast_log(sum);
  }
  // Debye branch with the hardware log2 / reciprocal / rsqrt: this runs once per row inside the streaming sampler
  // (fused log_prob), where the libm versions cost 13 % more vector instructions per launch
  const float r2 = v * v + x * x;
  const float ir = fast_rsqrt(r2);          // 1 / r
  const float r = r2 * ir;
  const float p = v * ir;
  const float rmx = v * v * fast_rcp(r + x);
  const float expo = rmx - v * fast_log(1.0f + (v + rmx) * fast_rcp(x));
  const float p2 = p * p;
  float pk = 1.0f, sum = 0.0f;
#pragma unroll
  for (int k = 0; k < VMF_DEBYE_K_F32; ++k) {
    float poly = 0.0f;
#pragma unroll
    for (int j = k; j >= 0; --j) poly = poly * p2 + (float)DEBYE_U[k][j];
    sum += pk * poly;
    pk *= ir;
  }
  return expo - 0.5f * fast_log(6.28318530717958647692f * r) + fast_log(sum);
}

// vMF constants that depend on m only (host side)
struct VmfDim {
  int m;
  double v;         // m/2 - 1
  double lg_v1;     // lgamma(v + 1) = lgamma(m/2)
  double lg_v2;     // lgamma(v + 2) = lgamma(m/2 + 1)
  double half_m_log2pi;
  double log_area;  // log |S^{m-1}| = log 2 + (m/2) log pi - lgamma(m/2)
};
static inline VmfDim make_vmf_dim(int m) {
  VmfDim d;
  d.m = m;
  d.v = 0.5 * m - 1.0;
  d.lg_v1 = lgamma(d.v + 1.0);
  d.lg_v2 = lgamma(d.v + 2.0);
  d.half_m_log2pi = 0.5 * m * log(2.0 * M_PI);
  d.log_area = log(2.0) + 0.5 * m * log(M_PI) - lgamma(0.5 * m);
  return d;
}

// log C_m(k) = v log k - (m/2) log 2pi - (log ive(v,k) + k),  v = m/2 - 1
VMF_HD double vmf_log_norm(const VmfDim& d, double k) {
  return d.v * ::log(k) - d.half_m_log2pi - (log_ive_pos(d.v, d.lg_v1, k) + k);
}
// the same in single precision (|error| ~ 1e-6 (1 + |log C_m|)); used when the output tensor is fp32
VMF_HD float vmf_log_norm_f32v(float v, float lg_v1, float half_m_log2pi, float k) {
  return v * fast_log(k) - half_m_log2pi - (log_ive_pos_f32(v, lg_v1, k) + k);
}
VMF_HD float vmf_log_norm_f32(const VmfDim& d, float k) {
  return vmf_log_norm_f32v((float)d.v, (float)d.lg_v1, (float)d.half_m_log2pi, k);
}
// A_m(k) = I_{m/2}(k) / I_{m/2-1}(k)
VMF_HD double vmf_mean_resultant(const VmfDim& d, double k) {
  return bessel_ratio_eval(d.v + 1.0, d.lg_v2, d.lg_v1, k);
}
// Both through the coefficient tables of the orders m/2 - 1 and m/2 (built once per launch on the host, read through
// scalar loads): what the double-precision kernels evaluate per element
struct VmfDimTab {
  IveTab t0, t1;
};
static inline VmfDimTab make_vmf_dim_tab(const VmfDim& d) {
  VmfDimTab t;
  t.t0 = make_ive_tab(make_bessel_order(d.v, 0));
  t.t1 = make_ive_tab(make_bessel_order(d.v + 1.0, 0));
  return t;
}
VMF_HD double vmf_log_norm_tab(const VmfDim& d, const VmfDimTab& t, double k) {
  return d.v * ::log(k) - d.half_m_log2pi - (log_ive_pos_tab(t.t0, d.v, d.lg_v1, k) + k);
}
VMF_HD double vmf_mean_resultant_tab(const VmfDim& d, const VmfDimTab& t, double k) {
  return bessel_ratio_tab(t.t1, t.t0, d.v + 1.0, d.lg_v2, d.lg_v1, k);
}
// log C_m(k) and A_m(k) together: the order m/2 - 1 is evaluated once
VMF_HD void vmf_log_norm_and_resultant_tab(const VmfDim& d, const VmfDimTab& t, double k, double* logc, double* A) {
  if (!(k > 0.0 && k <= VMF_IVE_TAB_XMAX)) {
    *logc = vmf_log_norm(d, k);
    *A = vmf_mean_resultant(d, k);
    return;
  }
  double E0, P0, E1, P1;
  ive_pos_tab_parts(t.t0, d.v, d.lg_v1, k, &E0, &P0);
  ive_pos_tab_parts(t.t1, d.v + 1.0, d.lg_v2, k, &E1, &P1);
  *logc = d.v * ::log(k) - d.half_m_log2pi - (E0 + ::log(P0) + k);
  *A = ::exp(E1 - E0) * P1 * rcp_nr(P0);
}
