// consumers_math.h -- per-row math of the steps on either side of the sampler: the torus embedding of two
// sphere draws and the supervised latent (geodesic) losses.  __host__ __device__ so the host check library
// can run the same functions on the CPU.
#pragma once
#include "common.h"

template <typename T> struct MT;  // trigonometry the sampler itself never needs
template <> struct MT<float> {
  // on the device: the hardware sine / cosine (arguments in revolutions; the angles here are differences of values in
  // [0, 2 pi], so no range reduction is needed and the error stays near 1e-6 absolute) instead of the library routines
  // with their large-argument paths
#if defined(__HIP_DEVICE_COMPILE__)
  static VMF_HD float sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189533577f); }
  static VMF_HD float cos(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189533577f); }
#else
  static VMF_HD float sin(float x) { return ::sinf(x); }
  static VMF_HD float cos(float x) { return ::cosf(x); }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  // atan2 on the device: atan(min / max) of the absolute values by a degree-8 polynomial in t^2 on [0, 1] (1e-7 absolute,
  // the hardware reciprocal for the quotient), then the octant / quadrant / sign folds -- about 25 instructions against
  // the library routine's 70.  Zero, huge, infinite and NaN operands take the library routine.
  static VMF_HD float atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    if (!(mx > 1e-30f && mx < 1e30f)) return ::atan2f(y, x);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float q = 2.456636634e-03f;
    q = fmaf(q, s, -1.440098602e-02f);
    q = fmaf(q, s, 3.978057951e-02f);
    q = fmaf(q, s, -7.234798372e-02f);
    q = fmaf(q, s, 1.049891561e-01f);
    q = fmaf(q, s, -1.416122019e-01f);
    q = fmaf(q, s, 1.998590529e-01f);
    q = fmaf(q, s, -3.333259821e-01f);
    q = fmaf(q, s, 9.999998808e-01f);
    float r = q * t;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return copysignf(r, y);
  }
  // acos x = atan2(sqrt((1 - x)(1 + x)), x): both factors are exact where they are small
  static VMF_HD float acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return ::acosf(x);
    return atan2(__builtin_amdgcn_sqrtf((1.0f - x) * (1.0f + x)), x);
  }
#else
  static VMF_HD float acos(float x) { return ::acosf(x); }
  static VMF_HD float atan2(float y, float x) { return ::atan2f(y, x); }
#endif
  static VMF_HD float fmod(float a, float b) { return ::fmodf(a, b); }
  // quotients and roots through the 1-ulp hardware reciprocal / square root (the IEEE sequences cost ~10 instructions
  // each, and a row's gradient has eight quotients)
  static VMF_HD float div(float a, float b) { return a * fast_rcp(b); }
  static VMF_HD float sqrt(float x) { return fast_sqrt(x); }
};
template <> struct MT<double> {
  static VMF_HD double sin(double x) { return ::sin(x); }
  static VMF_HD double cos(double x) { return ::cos(x); }
  static VMF_HD double acos(double x) { return ::acos(x); }
  static VMF_HD double atan2(double y, double x) { return ::atan2(y, x); }
  static VMF_HD double fmod(double a, double b) { return ::fmod(a, b); }
  static VMF_HD double div(double a, double b) { return a / b; }
  static VMF_HD double sqrt(double x) { return ::sqrt(x); }
};

// (phi + 2 pi) % (2 pi) as torch evaluates it: the Python scalar 2*pi is rounded to the tensor's dtype, and
// torch's % is the floored modulo (result takes the divisor's sign); dataloader/utils.py:117,142,168-169
template <typename T>
VMF_HD T wrap_2pi(T a) {
  const T two_pi = (T)6.283185307179586476925;
  const T s = a + two_pi;
  // every caller passes an atan2 result, so s lies in [pi, 3 pi]: there the remainder is s or s - 2 pi, and that
  // subtraction is exact (Sterbenz) -- the same bits as fmod without its division loop
  if (s >= (T)0 && s < (T)2 * two_pi) return s >= two_pi ? s - two_pi : s;
  T r = MT<T>::fmod(s, two_pi);
  if (r < (T)0) r += two_pi;
  return r;
}

// ---- torus embedding (model/torus_vae.py:60-81) --------------------------------------------------------------
// (cos, sin) of each angle are the first two components of the two sphere draws; y uses (r + r cos theta),
// as the reference does (:78).
template <typename T>
VMF_HD void torus_point(T ct, T st, T cp, T sp, T R, T r, T out[3]) {
  out[0] = (R + r * ct) * cp;
  out[1] = (r + r * ct) * sp;
  out[2] = r * st;
}
// gradients wrt (ct, st) and (cp, sp) given g = d loss / d out
template <typename T>
VMF_HD void torus_point_bwd(T ct, T cp, T sp, T R, T r, const T g[3], T gt[2], T gp[2]) {
  gt[0] = r * (cp * g[0] + sp * g[1]);
  gt[1] = r * g[2];
  gp[0] = (R + r * ct) * g[0];
  gp[1] = (r + r * ct) * g[1];
}

// ---- latent losses (dataloader/utils.py:101-180), one row ----------------------------------------------------
// Each returns the row's contribution l(z, label); the caller averages.  With GRAD the derivative wrt the row of z
// is written to dz[0..zdim) (not yet scaled by the upstream gradient / row count).
enum { VMF_LL_S1 = VMF_LATENT_S1, VMF_LL_S2 = VMF_LATENT_S2, VMF_LL_T2 = VMF_LATENT_T2 };

// Single precision takes the same quantities WITHOUT the inverse trigonometric functions: every angle the reference
// recovers (atan2, acos) enters its loss only through cosines and sines of differences with a label angle, and
//   1 - cos(a - b) = [(cos a - cos b)^2 + (sin a - sin b)^2] / 2,   sin(a - b) = sin a cos b - cos a sin b
// with (cos, sin) of the recovered angle read off the row itself (x / rho, y / rho; z_3 / |z|, sqrt(1 - (z_3/|z|)^2)).
// The half-chord form has no cancellation, so a nearly converged row (difference d) keeps the relative accuracy the
// reference's float evaluation has (absolute error ~1e-7 in each unit vector against a chord of length d) -- and a row
// costs two sine / cosine pairs and two reciprocal roots instead of two polynomial arctangents, an arccosine and five
// sines / cosines.  Rows at a pole of the angle (rho or the tube radius below 1e-15 or beyond 1e15) take the
// reference's sequence.
VMF_HD bool ll_tame(float a) { return a > 1e-30f && a < 1e30f; }
VMF_HD float ll_half_chord2(float c0, float s0, float c1, float s1) {
  const float dc = c0 - c1, ds = s0 - s1;
  return 0.5f * (dc * dc + ds * ds);
}

template <typename T, bool GRAD>
VMF_HD T latent_loss_s1(const T* z, int zdim, T gt_phi, T* dz) {
  const T x = z[0], y = z[1];
  if constexpr (sizeof(T) == 4) {
    const float r2 = x * x + y * y;
    if (ll_tame(r2)) {
      const float ir = MT<float>::div(1.0f, MT<float>::sqrt(r2));
      const float cp = x * ir, sp = y * ir, cg = MT<float>::cos(gt_phi), sg = MT<float>::sin(gt_phi);
      const float om = ll_half_chord2(cp, sp, cg, sg);
      if (GRAD) {
        const float dphi = 2.0f * om * (sp * cg - cp * sg);
        const float q = MT<float>::div(dphi, r2);
        dz[0] = -y * q;
        dz[1] = x * q;
        for (int c = 2; c < zdim; ++c) dz[c] = 0.0f;
      }
      return om * om;
    }
  }
  T phi = wrap_2pi<T>(MT<T>::atan2(y, x));             // :116-117
  T a = phi - gt_phi;
  T d = MT<T>::cos(a);                                 // :121
  T om = (T)1 - d;
  if (GRAD) {
    T dphi = (T)2 * om * MT<T>::sin(a);
    T q = MT<T>::div(dphi, x * x + y * y);
    dz[0] = -y * q;
    dz[1] = x * q;
    for (int c = 2; c < zdim; ++c) dz[c] = (T)0;
  }
  return om * om;                                      // :122
}

template <typename T, bool GRAD>
VMF_HD T latent_loss_s2(const T* z, int zdim, T theta_gt, T phi_gt, T* dz) {
  T ss = (T)0;
  for (int c = 0; c < zdim; ++c) ss += z[c] * z[c];
  const T nrm = MT<T>::sqrt(ss);                       // :134
  const T x = z[0], y = z[1];
  const T raw = MT<T>::div(z[2], nrm);
  const T zc = raw < (T)-1 ? (T)-1 : (raw > (T)1 ? (T)1 : raw);   // :138
  if constexpr (sizeof(T) == 4) {
    const float r2 = x * x + y * y;
    if (ll_tame(r2) && ll_tame(ss)) {
      const float sth = MT<float>::sqrt((1.0f - zc) * (1.0f + zc));  // sin(theta_hat) >= 0: theta_hat lies in [0, pi]
      const float ir = MT<float>::div(1.0f, MT<float>::sqrt(r2));
      const float cph = x * ir, sph = y * ir;
      const float cgt = MT<float>::cos(theta_gt), sgt = MT<float>::sin(theta_gt);
      const float cpg = MT<float>::cos(phi_gt), spg = MT<float>::sin(phi_gt);
      const float term1 = ll_half_chord2(cgt, sgt, zc, sth);   // 1 - cos(theta_gt - theta_hat)   :147
      const float omc = ll_half_chord2(cpg, spg, cph, sph);    // 1 - cos(phi_gt - phi_hat)
      const float term2 = sgt * sth * omc;                     // :148
      const float s = term1 + term2;
      if (GRAD) {
        const float dl = 2.0f * s;
        const float sin_dth = sgt * zc - cgt * sth, sin_dph = spg * cph - cpg * sph;
        const float d_theta_hat = dl * (-sin_dth + sgt * zc * omc);
        const float d_phi_hat = dl * (-(sgt * sth * sin_dph));
        const float d_raw = (raw >= -1.0f && raw <= 1.0f) ? MT<float>::div(-d_theta_hat, sth) : 0.0f;
        const float qphi = MT<float>::div(d_phi_hat, r2);
        const float rn = MT<float>::div(d_raw, nrm);
        const float k3 = MT<float>::div(-rn * z[2], ss);
        for (int c = 0; c < zdim; ++c) dz[c] = k3 * z[c];
        dz[2] += rn;
        dz[0] -= y * qphi;
        dz[1] += x * qphi;
      }
      return s * s;                                            // :149
    }
  }
  const T theta_hat = MT<T>::acos(zc);                 // :140
  const T phi_hat = wrap_2pi<T>(MT<T>::atan2(y, x));   // :141-142
  const T dth = theta_gt - theta_hat, dph = phi_gt - phi_hat;
  const T sgt = MT<T>::sin(theta_gt), sth = MT<T>::sin(theta_hat);
  const T omc = (T)1 - MT<T>::cos(dph);
  const T term1 = (T)1 - MT<T>::cos(dth);              // :147
  const T term2 = sgt * sth * omc;                     // :148
  const T s = term1 + term2;
  if (GRAD) {
    const T dl = (T)2 * s;
    const T d_theta_hat = dl * (-MT<T>::sin(dth) + sgt * MT<T>::cos(theta_hat) * omc);
    const T d_phi_hat = dl * (-(sgt * sth * MT<T>::sin(dph)));
    // acos' = -1/sqrt(1-c^2); clamp passes the gradient only inside [-1, 1]
    const T d_raw = (raw >= (T)-1 && raw <= (T)1) ? MT<T>::div(-d_theta_hat, MT<T>::sqrt((T)1 - zc * zc)) : (T)0;
    const T qphi = MT<T>::div(d_phi_hat, x * x + y * y);
    const T rn = MT<T>::div(d_raw, nrm);              // d raw / |z|
    const T k3 = MT<T>::div(-rn * z[2], ss);          // -d raw z_3 / |z|^3
    for (int c = 0; c < zdim; ++c) dz[c] = k3 * z[c];
    dz[2] += rn;
    dz[0] -= y * qphi;
    dz[1] += x * qphi;
  }
  return s * s;                                        // :149
}

template <typename T, bool GRAD>
VMF_HD T latent_loss_t2(const T* z, int zdim, T l_theta, T l_phi, T R, T r, T* dz) {
  const T x = z[0], y = z[1], zc = z[2];
  const T rho = MT<T>::sqrt(x * x + y * y);            // :162
  const T s = rho - R;
  if constexpr (sizeof(T) == 4) {
    const float r2 = x * x + y * y, h2 = s * s + zc * zc;
    if (ll_tame(r2) && ll_tame(h2)) {
      const float ir = MT<float>::div(1.0f, rho), ih = MT<float>::div(1.0f, MT<float>::sqrt(h2));
      const float cph = x * ir, sph = y * ir, cth = s * ih, sth = zc * ih;
      const float clt = MT<float>::cos(l_theta), slt = MT<float>::sin(l_theta);
      const float clp = MT<float>::cos(l_phi), slp = MT<float>::sin(l_phi);
      const float q = h2 - r * r;                              // :174
      const float val = ll_half_chord2(cth, sth, clt, slt) + ll_half_chord2(cph, sph, clp, slp) + 0.1f * (q * q);
      if (GRAD) {
        const float d_theta = sth * clt - cth * slt, d_phi = sph * clp - cph * slp, d_q = 0.2f * q;
        const float qth = MT<float>::div(d_theta, h2);
        const float d_s = -zc * qth + d_q * 2.0f * s;
        const float d_zc = s * qth + d_q * 2.0f * zc;
        const float qs = d_s * ir, qphi = MT<float>::div(d_phi, r2);
        dz[0] = x * qs - y * qphi;
        dz[1] = y * qs + x * qphi;
        dz[2] = d_zc;
        for (int c = 3; c < zdim; ++c) dz[c] = 0.0f;
      }
      return val;
    }
  }
  const T phi = wrap_2pi<T>(MT<T>::atan2(y, x));       // :164,168
  const T theta = wrap_2pi<T>(MT<T>::atan2(zc, s));    // :165,169
  const T a = theta - l_theta, b = phi - l_phi;
  const T q = s * s + zc * zc - r * r;                 // :174
  // mean(1-cos a) + mean(1-cos b) + 0.1 mean(q^2)  ==  mean of the row value below (:172-177)
  const T val = ((T)1 - MT<T>::cos(a)) + ((T)1 - MT<T>::cos(b)) + (T)0.1 * (q * q);
  if (GRAD) {
    const T d_theta = MT<T>::sin(a), d_phi = MT<T>::sin(b), d_q = (T)0.2 * q;
    const T qth = MT<T>::div(d_theta, s * s + zc * zc);
    const T d_s = -zc * qth + d_q * (T)2 * s;
    const T d_zc = s * qth + d_q * (T)2 * zc;
    const T qs = MT<T>::div(d_s, rho), qphi = MT<T>::div(d_phi, rho * rho);
    dz[0] = x * qs - y * qphi;
    dz[1] = y * qs + x * qphi;
    dz[2] = d_zc;
    for (int c = 3; c < zdim; ++c) dz[c] = (T)0;
  }
  return val;
}
