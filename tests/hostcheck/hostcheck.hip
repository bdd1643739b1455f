// hostcheck.hip -- TEST ONLY.  Runs the library's __host__ __device__ per-element math (bessel_dev.h,
// vmf_math.h, consumers_math.h) on the CPU so that the arithmetic the kernels execute can be compared with the oracle without a
// GPU.  It is not part of libvmf_hip.so and nothing in the package loads it.
#include "../../medicalimageanalysis_amd/csrc/vmf_math.h"
#include "../../medicalimageanalysis_amd/csrc/consumers_math.h"

void vmf_set_error(const char*, ...) {}
int vmf_check_launch(const char*) { return 0; }

extern "C" {
void hc_ive_fwd(double v, const double* z, double* out, int64_t n) {
  BesselOrder o = make_bessel_order(v, !(v == 0.0 || v == 1.0));
  const IveTab tab = make_ive_tab(o);  // what vmf_ive_fwd hands its kernel
  for (int64_t i = 0; i < n; ++i) out[i] = ive_eval_tab(o, tab, ive_kind_of(o), z[i]);
}
void hc_ive_vm1(double v, const double* z, double* out, int64_t n) {
  BesselOrder o = make_bessel_order(v - 1.0, v != 1.0);
  const IveTab tab = make_ive_tab(o);  // what vmf_ive_bwd hands its kernel for the order v - 1
  for (int64_t i = 0; i < n; ++i) out[i] = ive_eval_tab(o, tab, ive_kind_of(o), z[i]);
}
void hc_log_ive(double v, const double* z, double* out, int64_t n) {
  double lg = lgamma(v + 1.0);
  for (int64_t i = 0; i < n; ++i) out[i] = log_ive_pos(v, lg, z[i]);
}
void hc_ratio(double v, const double* z, double* out, int64_t n) {
  double lg1 = lgamma(v + 1.0), lg0 = lgamma(v);
  for (int64_t i = 0; i < n; ++i) out[i] = bessel_ratio_eval(v, lg1, lg0, z[i]);
}
// the table-driven forms the kernels evaluate (tables as the entry points build them)
void hc_log_ive_tab(double v, const double* z, double* out, int64_t n) {
  const double lg = lgamma(v + 1.0);
  const IveTab tab = make_ive_tab(make_bessel_order(v, 0));
  for (int64_t i = 0; i < n; ++i) out[i] = log_ive_pos_tab(tab, v, lg, z[i]);
}
void hc_ratio_tab(double v, const double* z, double* out, int64_t n) {
  const double lg1 = lgamma(v + 1.0), lg0 = lgamma(v);
  const IveTab t1 = make_ive_tab(make_bessel_order(v, 0)), t0 = make_ive_tab(make_bessel_order(v - 1.0, 0));
  for (int64_t i = 0; i < n; ++i) out[i] = bessel_ratio_tab(t1, t0, v, lg1, lg0, z[i]);
}
void hc_ive_half_pair(const double* z, double* i12, double* i32, int64_t n) {
  for (int64_t i = 0; i < n; ++i) ive_half_pair(z[i], &i12[i], &i32[i]);
}
void hc_log_norm_tab(int m, const double* k, double* out, double* a_out, int64_t n) {
  const VmfDim d = make_vmf_dim(m);
  const VmfDimTab t = make_vmf_dim_tab(d);
  for (int64_t i = 0; i < n; ++i) {
    double lc, A;
    vmf_log_norm_and_resultant_tab(d, t, k[i], &lc, &A);
    out[i] = vmf_log_norm_tab(d, t, k[i]);
    a_out[i] = A;
    if (lc != out[i]) out[i] = NAN;  // the two spellings must agree exactly
  }
}
void hc_log_norm(int m, const double* k, double* out, int64_t n) {
  VmfDim d = make_vmf_dim(m);
  for (int64_t i = 0; i < n; ++i) out[i] = vmf_log_norm(d, k[i]);
}
void hc_log_norm_f32(int m, const float* k, float* out, int64_t n) {
  VmfDim d = make_vmf_dim(m);
  for (int64_t i = 0; i < n; ++i) out[i] = vmf_log_norm_f32(d, k[i]);
}
void hc_log_norm_circle_f32(const float* k, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = vmf_log_norm_circle_f32(k[i]);
}
#define HC_LATENT(NAME, T)                                                                                       \
  void NAME(const T* z, const T* labels, T* rows, T* dz, int64_t n, int zdim, int64_t nl, int lc, int variant,   \
            T R, T r) {                                                                                          \
    for (int64_t i = 0; i < n; ++i) {                                                                            \
      const T* lab = labels + (i % nl) * lc;                                                                     \
      T d[8];                                                                                                    \
      T v;                                                                                                       \
      if (variant == VMF_LL_S1) v = latent_loss_s1<T, true>(z + i * zdim, zdim, lab[0], d);                      \
      else if (variant == VMF_LL_S2) v = latent_loss_s2<T, true>(z + i * zdim, zdim, lab[0], lab[1], d);         \
      else v = latent_loss_t2<T, true>(z + i * zdim, zdim, lab[0], lab[1], R, r, d);                             \
      rows[i] = v;                                                                                               \
      for (int c = 0; c < zdim; ++c) dz[i * zdim + c] = d[c];                                                    \
    }                                                                                                            \
  }
HC_LATENT(hc_latent_f64, double)
HC_LATENT(hc_latent_f32, float)
#define HC_TORUS(NAME, T)                                                                                        \
  void NAME(const T* zt, const T* zp, const T* g, T* out, T* gt, T* gp, int64_t n, int zdim, T R, T r) {         \
    for (int64_t i = 0; i < n; ++i) {                                                                            \
      torus_point<T>(zt[i * zdim], zt[i * zdim + 1], zp[i * zdim], zp[i * zdim + 1], R, r, out + 3 * i);         \
      torus_point_bwd<T>(zt[i * zdim], zp[i * zdim], zp[i * zdim + 1], R, r, g + 3 * i, gt + 2 * i, gp + 2 * i); \
    }                                                                                                            \
  }
HC_TORUS(hc_torus_f64, double)
HC_TORUS(hc_torus_f32, float)
void hc_kl_f64(const double* k, double* out, double* dout, int64_t n, int variant) {
  KlBessel kb = make_kl_bessel();
  for (int64_t i = 0; i < n; ++i) out[i] = kl_value<double>(k[i], variant, kb, &dout[i]);
}
// returns 0 when the variant has no fast form
int hc_kl_fast_f32(const float* k, float* out, float* dout, int64_t n, int variant) {
  int ok = 1;
  for (int64_t i = 0; i < n; ++i) ok &= (int)kl_value_fast(k[i], variant, &out[i], &dout[i]);
  return ok;
}
// the value-only form of the fp32 forward kernels (for the S^1 variants: polynomial fits of the KL itself)
void hc_kl_value_f32(const float* k, float* out, int64_t n, int variant) {
  KlBessel kb = make_kl_bessel();
  for (int64_t i = 0; i < n; ++i) out[i] = kl_eval_value<float, false>(k[i], variant, kb);
}

void hc_kl_f32(const float* k, float* out, float* dout, int64_t n, int variant) {
  KlBessel kb = make_kl_bessel();
  for (int64_t i = 0; i < n; ++i) out[i] = kl_value<float>(k[i], variant, kb, &dout[i]);
}
// elementwise-layout rsample forward/backward rows
void hc_v3_fwd_f64(const double* loc, const double* scale, const double* u, const double* eps, double* z, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    double l[3], t[3], v[3], nn, p, s, L, x2;
    bool inside;
    v3_normalise_loc<double>(loc + 3 * i, l, &nn);
    double w = v3_w<double>(u[i], scale[i], &L, &x2);
    v3_tangent<double>(eps + 3 * i, l, t, v, &p, &s);
    double f = v3_factor<double>(w, &inside);
    for (int c = 0; c < 3; ++c) z[3 * i + c] = w * l[c] + f * v[c];
  }
}
void hc_v3_bwd_f64(const double* loc, const double* scale, const double* u, const double* eps, const double* g,
                   double* dloc, double* dscale, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    v3_backward_row<double>(loc + 3 * i, scale[i], u[i], eps + 3 * i, g + 3 * i, dloc + 3 * i, &dscale[i]);
}
void hc_wood_f64(const double* kappa, const double* pe, const double* pu, double* w, int32_t* idx, int64_t n, int m,
                 int n_prop) {
  for (int64_t i = 0; i < n; ++i) {
    double a, b, d, ww = 0, omw = 0;
    wood_consts<double>(kappa[i], m, &a, &b, &d);
    int k = 0;
    for (; k < n_prop; ++k) {
      bool acc = wood_try<double>(a, b, d, m, pe[i * n_prop + k], pu[i * n_prop + k], &ww, &omw);
      if (k + 1 >= n_prop) acc = true;
      if (acc) break;
    }
    w[i] = ww;
    idx[i] = k;
  }
}
}
