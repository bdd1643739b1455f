// abi_bench -- the headline workload driven through the C ABI of libvmf_hip.so from plain C++ (no Python, no
// PyTorch): hipMalloc'ed buffers, unit mean directions from the library's own sphere sampler, kappa log-uniform on
// [1, 1e4], then `steps` timed launches of vmf_rsample_fwd (+ fused log_prob) bracketed by HIP events.  Prints one
// JSON line.  Cross-checks bench.py's timing and shows what a non-Python binding has to do.
//   usage: abi_bench [rows=16777216] [m=64] [steps=50] [warmup=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../include/vmf_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define VMF_OK_(x)                                                                 \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ != 0) {                                                                \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, vmf_last_error_string());         \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

__global__ void fill_kappa(float* k, long long n, float log_hi) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // a fixed low-discrepancy sequence on [0, 1): the golden-ratio rotation
    double u = fmod(0.5 + 0.6180339887498949 * (double)i, 1.0);
    k[i] = expf((float)u * log_hi);
  }
}

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : (1ll << 24);
  const int m = argc > 2 ? atoi(argv[2]) : 64;
  const int steps = argc > 3 ? atoi(argv[3]) : 50;
  const int warmup = argc > 4 ? atoi(argv[4]) : 20;
  if (n <= 0 || m < 2 || steps < 1 || warmup < 0) {
    fprintf(stderr, "usage: abi_bench [rows] [m] [steps] [warmup]\n");
    return 1;
  }
  float *mu, *kappa, *z, *logp;
  HIP_OK(hipMalloc(&mu, sizeof(float) * n * m));
  HIP_OK(hipMalloc(&z, sizeof(float) * n * m));
  HIP_OK(hipMalloc(&kappa, sizeof(float) * n));
  HIP_OK(hipMalloc(&logp, sizeof(float) * n));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  VMF_OK_(vmf_sphere_uniform_sample(nullptr, mu, n, m, 0, 1234, 0, nullptr, VMF_F32, stream));
  hipLaunchKernelGGL(fill_kappa, dim3(2048), dim3(256), 0, stream, kappa, n, logf(1e4f));
  HIP_OK(hipGetLastError());
  for (int i = 0; i < warmup; ++i)
    VMF_OK_(vmf_rsample_fwd(mu, m, kappa, 1, z, nullptr, nullptr, nullptr, logp, nullptr, n, m, nullptr, nullptr, 0, nullptr, 64,
                            0, 0x5EED, (uint64_t)i, nullptr, 0, VMF_F32, stream));
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<hipEvent_t> ev(2 * steps);
  for (auto& e : ev) HIP_OK(hipEventCreate(&e));
  for (int i = 0; i < steps; ++i) {
    HIP_OK(hipEventRecord(ev[2 * i], stream));
    VMF_OK_(vmf_rsample_fwd(mu, m, kappa, 1, z, nullptr, nullptr, nullptr, logp, nullptr, n, m, nullptr, nullptr, 0, nullptr, 64,
                            0, 0x5EED, (uint64_t)(warmup + i), nullptr, 0, VMF_F32, stream));
    HIP_OK(hipEventRecord(ev[2 * i + 1], stream));
  }
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<float> ms(steps);
  double sum = 0;
  for (int i = 0; i < steps; ++i) {
    HIP_OK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]));
    sum += ms[i];
  }
  std::sort(ms.begin(), ms.end());
  // self-check on a strided sample of rows: unit norm, finite log p
  const long long nchk = std::min<long long>(n, 4096), stride = std::max<long long>(1, n / nchk);
  std::vector<float> row(m);
  int bad = 0;
  for (long long r = 0; r < n && bad == 0; r += stride) {
    float lp;
    HIP_OK(hipMemcpy(row.data(), z + r * m, sizeof(float) * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&lp, logp + r, sizeof(float), hipMemcpyDeviceToHost));
    double nn = 0;
    for (int c = 0; c < m; ++c) nn += (double)row[c] * row[c];
    if (!(std::fabs(nn - 1.0) < 1e-3) || !std::isfinite(lp)) ++bad;
  }
  const double mean_ms = sum / steps, bytes = (double)n * (2.0 * m + 2.0) * 4.0;
  printf("{\"driver\": \"C ABI, no Python\", \"rows\": %lld, \"m\": %d, \"steps\": %d, \"warmup\": %d, "
         "\"kernel_ms_mean\": %.4f, \"kernel_ms_median\": %.4f, \"kernel_ms_min\": %.4f, \"samples_per_s\": %.4g, "
         "\"algorithmic_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"self_check\": \"%s\"}\n",
         n, m, steps, warmup, mean_ms, ms[steps / 2], ms[0], (double)n / (mean_ms * 1e-3),
         bytes / (mean_ms * 1e-3) / 1e9, bytes / (mean_ms * 1e-3) / 1e9 / 8000.0, bad ? "FAILED" : "ok");
  (void)hipFree(mu);
  (void)hipFree(z);
  (void)hipFree(kappa);
  (void)hipFree(logp);
  return bad ? 4 : 0;
}
