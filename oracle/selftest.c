/* TEST INFRASTRUCTURE ONLY -- a standalone driver that runs the oracle's entry points on small ragged inputs.  Built
   with -fsanitize=address,undefined by `make -C oracle selftest` and executed by tests/test_oracle_sanitized.py: the
   checker the parity tests rely on must itself be free of out-of-bounds accesses and undefined arithmetic. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int orc_vmf_rsample_philox_f32(const float*, const float*, float*, float*, int32_t*, int64_t, int, int, uint64_t,
                               uint64_t, uint64_t);
int orc_vmf_rsample_philox_f64(const double*, const double*, double*, double*, int32_t*, int64_t, int, int, uint64_t,
                               uint64_t, uint64_t);
int orc_vmf_log_prob_f64(const double*, const double*, const double*, double*, int64_t, int);
int orc_vmf_entropy_kl_f64(const double*, double*, double*, int64_t, int);
int orc_vmf_kl_f64(const double*, double*, double*, double*, int64_t, int);
int orc_vmf_kl_f32(const float*, float*, float*, double*, int64_t, int);
int orc_ive_fwd_f64(double, const double*, double*, int64_t);
int orc_ive_bwd_f64(double, const double*, const double*, double*, int64_t);
int orc_vmf3_rsample_fwd_f64(const double*, const double*, const double*, const double*, double*, double*, int,
                             const int64_t*, const int64_t*, const int64_t*, const int64_t*, const int64_t*);

static double frand(uint64_t* s) {
  *s = *s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)((*s >> 11) + 1) / 9007199254740994.0;
}

int main(void) {
  uint64_t s = 1;
  int fails = 0;
  const int ms[] = {2, 3, 5, 10, 64, 130};
  for (unsigned t = 0; t < sizeof(ms) / sizeof(ms[0]); ++t) {
    const int m = ms[t];
    const int64_t n = 257;
    double* mu = malloc(sizeof(double) * n * m);
    double* z = malloc(sizeof(double) * n * m);
    double *kap = malloc(sizeof(double) * n), *w = malloc(sizeof(double) * n), *lp = malloc(sizeof(double) * n);
    float *muf = malloc(sizeof(float) * n * m), *zf = malloc(sizeof(float) * n * m);
    float *kapf = malloc(sizeof(float) * n), *wf = malloc(sizeof(float) * n);
    int32_t* idx = malloc(sizeof(int32_t) * n);
    for (int64_t i = 0; i < n; ++i) {
      double nn = 0;
      for (int c = 0; c < m; ++c) {
        mu[i * m + c] = frand(&s) - 0.5;
        nn += mu[i * m + c] * mu[i * m + c];
      }
      for (int c = 0; c < m; ++c) {
        mu[i * m + c] /= sqrt(nn);
        muf[i * m + c] = (float)mu[i * m + c];
      }
      kap[i] = exp(frand(&s) * log(1e5));
      kapf[i] = (float)kap[i];
    }
    fails += orc_vmf_rsample_philox_f64(mu, kap, z, w, idx, n, m, 64, 7, 3, 1) != 0;
    fails += orc_vmf_rsample_philox_f32(muf, kapf, zf, wf, idx, n, m, 64, 7, 3, 1) != 0;
    fails += orc_vmf_log_prob_f64(mu, kap, z, lp, n, m) != 0;
    fails += orc_vmf_entropy_kl_f64(kap, w, lp, n, m) != 0;
    for (int64_t i = 0; i < n; ++i) {
      double nn = 0;
      for (int c = 0; c < m; ++c) nn += z[i * m + c] * z[i * m + c];
      if (!(fabs(nn - 1.0) < 1e-9) || !isfinite(lp[i])) ++fails;
    }
    free(mu); free(z); free(kap); free(w); free(lp); free(muf); free(zf); free(kapf); free(wf); free(idx);
  }
  {
    const int64_t n = 101;
    double kd[101], out[101], dout[101], sum;
    float kf[101], outf[101], doutf[101];
    for (int64_t i = 0; i < n; ++i) {
      kd[i] = exp(-7.0 + 14.0 * (double)i / (double)(n - 1));  /* 1e-3 .. 1e3: every branch, overflow included */
      kf[i] = (float)kd[i];
    }
    for (int v = 0; v <= 5; ++v) {
      fails += orc_vmf_kl_f64(kd, out, dout, &sum, n, v) != 0;
      fails += orc_vmf_kl_f32(kf, outf, doutf, &sum, n, v) != 0;
    }
    const double orders[] = {0.0, 1.0, 0.5, 1.5, -0.5, 31.0, 255.0};
    for (unsigned t = 0; t < sizeof(orders) / sizeof(orders[0]); ++t) {
      fails += orc_ive_fwd_f64(orders[t], kd, out, n) != 0;
      if (orders[t] >= 0) fails += orc_ive_bwd_f64(orders[t], kd, out, dout, n) != 0;
    }
  }
  {
    /* the (B, 1) scale broadcast: output (4, 4, 3) */
    const int64_t odims[2] = {4, 4}, w_str[2] = {1, 0}, k_str[2] = {1, 0}, l_str[2] = {0, 1}, e_str[2] = {0, 1};
    double loc[12], scale[4], u[4], eps[12], z[48], w[4];
    for (int i = 0; i < 12; ++i) { loc[i] = frand(&s) - 0.5; eps[i] = frand(&s) - 0.5; }
    for (int i = 0; i < 4; ++i) { scale[i] = 1.0 + 50.0 * frand(&s); u[i] = frand(&s); }
    fails += orc_vmf3_rsample_fwd_f64(loc, scale, u, eps, z, w, 2, odims, w_str, k_str, l_str, e_str) != 0;
  }
  printf("oracle selftest: %d failure(s)\n", fails);
  return fails != 0;
}
