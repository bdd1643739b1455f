// vmf3.hip -- VonMisesFisher3D on gfx950: fused normalise + inverse-CDF w + tangent + combine, its backward,
// the m = 3 log_prob / mean, and the library's error plumbing.
//
// One thread owns one OUTPUT row (3 values).  The reference launches ~25 elementwise kernels and
// materialises every intermediate (sampler/von_mises_fisher.py:30, 50-86); here each row is 28 bytes of
// algorithmic traffic in elementwise layout (loc 12 + scale 4 + z 12 in fp32) and everything else stays in
// registers.  Broadcast results (the (B,1)-scale quirk that yields (B,B,3)) re-derive the per-i and per-j
// pieces in every thread: the arithmetic is ~100 flops per 12 written bytes, far below the HBM time.
#include <stdarg.h>
#include <stdio.h>

#include "vmf_math.h"

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void vmf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vmf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vmf_set_error("%s: %s", what, hipGetErrorString(e));
    return VMF_ERR_LAUNCH;
  }
  return VMF_OK;
}

extern "C" int vmf_version(void) { return VMF_HIP_VERSION; }
extern "C" const char* vmf_last_error_string(void) { return g_err; }

// ---- broadcast descriptor ----------------------------------------------------------------------
struct Bcast {
  int ndim;
  int identity;  // every operand is indexed by the output row itself
  int small;     // the result has fewer than 2^31 rows: the index is decomposed with 32-bit divisions
  int64_t odims[8], w_str[8], k_str[8], l_str[8], e_str[8];
};

__device__ __forceinline__ void unravel(const Bcast& bc, int64_t o, int64_t* wi, int64_t* ki, int64_t* li,
                                        int64_t* ei) {
  if (bc.identity) {
    *wi = *ki = *li = *ei = o;
    return;
  }
  int64_t a = 0, b = 0, c = 0, d = 0;
  if (bc.small) {  // a 64-bit division costs several times the rest of the row's arithmetic
    uint32_t o32 = (uint32_t)o;
    for (int dm = bc.ndim - 1; dm >= 0; --dm) {
      const uint32_t dim = (uint32_t)bc.odims[dm];
      const uint32_t q = o32 / dim;
      const int64_t idx = (int64_t)(o32 - q * dim);
      o32 = q;
      a += idx * bc.w_str[dm]; b += idx * bc.k_str[dm]; c += idx * bc.l_str[dm]; d += idx * bc.e_str[dm];
    }
    *wi = a; *ki = b; *li = c; *ei = d;
    return;
  }
  for (int dm = bc.ndim - 1; dm >= 0; --dm) {
    int64_t q = o / bc.odims[dm];
    int64_t idx = o - q * bc.odims[dm];
    o = q;
    a += idx * bc.w_str[dm]; b += idx * bc.k_str[dm]; c += idx * bc.l_str[dm]; d += idx * bc.e_str[dm];
  }
  *wi = a; *ki = b; *li = c; *ei = d;
}

static int make_bcast(Bcast* bc, int64_t* n_out, int ndim, const int64_t* odims, const int64_t* w_str,
                      const int64_t* k_str, const int64_t* l_str, const int64_t* e_str) {
  VMF_REQUIRE(ndim >= 1 && ndim <= 8, VMF_ERR_BAD_ARG, "ndim=%d outside [1,8]", ndim);
  VMF_REQUIRE(odims && w_str && k_str && l_str && e_str, VMF_ERR_BAD_ARG, "null broadcast arrays");
  bc->ndim = ndim;
  int64_t n = 1;
  for (int d = 0; d < ndim; ++d) {
    VMF_REQUIRE(odims[d] >= 0, VMF_ERR_BAD_ARG, "negative dim");
    VMF_REQUIRE(w_str[d] >= 0 && k_str[d] >= 0 && l_str[d] >= 0 && e_str[d] >= 0, VMF_ERR_BAD_ARG,
                "negative stride");
    bc->odims[d] = odims[d]; bc->w_str[d] = w_str[d]; bc->k_str[d] = k_str[d];
    bc->l_str[d] = l_str[d]; bc->e_str[d] = e_str[d];
    n *= odims[d];
  }
  for (int d = ndim; d < 8; ++d) {
    bc->odims[d] = 1; bc->w_str[d] = bc->k_str[d] = bc->l_str[d] = bc->e_str[d] = 0;
  }
  // identity: contiguous strides for all four operands
  int ident = 1;
  int64_t s = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    if (odims[d] != 1 && (w_str[d] != s || k_str[d] != s || l_str[d] != s || e_str[d] != s)) ident = 0;
    s *= odims[d];
  }
  bc->identity = ident;
  bc->small = n < ((int64_t)1 << 31);
  *n_out = n;
  return VMF_OK;
}

// noise for one row: u from Philox(index wi, slot 0), eps row from Philox(index ei, slot 1)
template <typename T>
__device__ __forceinline__ T v3_draw_u(const T* u, int64_t wi, PhiloxKey key) {
  if (u) return u[wi];
  return u01<T>(philox4x32_10((uint64_t)wi, 0u, key).x);
}
template <typename T>
__device__ __forceinline__ void v3_draw_eps(const T* eps, int64_t ei, PhiloxKey key, T e[3]) {
  if (eps) {
    e[0] = eps[3 * ei]; e[1] = eps[3 * ei + 1]; e[2] = eps[3 * ei + 2];
    return;
  }
  T n4[4];
  philox_normal4<T>((uint64_t)ei, 1u, key, n4);
  e[0] = n4[0]; e[1] = n4[1]; e[2] = n4[2];
}

template <typename T>
__global__ __launch_bounds__(256) void vmf3_fwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                       const T* __restrict__ u, const T* __restrict__ eps,
                                                       T* __restrict__ z, T* __restrict__ w_out, Bcast bc,
                                                       int64_t n_out, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
    int64_t wi, ki, li, ei;
    unravel(bc, o, &wi, &ki, &li, &ei);
    T r[3] = {loc[3 * li], loc[3 * li + 1], loc[3 * li + 2]};
    T k = scale[ki];
    T uu = v3_draw_u<T>(u, wi, key);
    T e[3];
    v3_draw_eps<T>(eps, ei, key, e);
    T l[3], t[3], v[3], n, p, s, L, x2;
    bool inside;
    v3_normalise_loc<T>(r, l, &n);
    T w = v3_w<T>(uu, k, &L, &x2);
    v3_tangent<T>(e, l, t, v, &p, &s);
    T f = v3_factor<T>(w, &inside);
    z[3 * o] = w * l[0] + f * v[0];
    z[3 * o + 1] = w * l[1] + f * v[1];
    z[3 * o + 2] = w * l[2] + f * v[2];
    if (w_out) w_out[wi] = w;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vmf3_bwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                       const T* __restrict__ u, const T* __restrict__ eps,
                                                       const T* __restrict__ gz, T* __restrict__ dloc,
                                                       T* __restrict__ dscale, Bcast bc, int64_t n_out,
                                                       int per_row, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
    int64_t wi, ki, li, ei;
    unravel(bc, o, &wi, &ki, &li, &ei);
    T r[3] = {loc[3 * li], loc[3 * li + 1], loc[3 * li + 2]};
    T k = scale[ki];
    T uu = v3_draw_u<T>(u, wi, key);
    T e[3];
    v3_draw_eps<T>(eps, ei, key, e);
    T g[3] = {gz[3 * o], gz[3 * o + 1], gz[3 * o + 2]};
    T dr[3], dk;
    v3_backward_row<T>(r, k, uu, e, g, dr, &dk);
    if (per_row) {  // one gradient row per output row; the caller sums the rows that share an operand
      dloc[3 * o] = dr[0]; dloc[3 * o + 1] = dr[1]; dloc[3 * o + 2] = dr[2];
      dscale[o] = dk;
    } else if (bc.identity) {
      dloc[3 * li] = dr[0]; dloc[3 * li + 1] = dr[1]; dloc[3 * li + 2] = dr[2];
      dscale[ki] = dk;
    } else {
      atomicAdd(&dloc[3 * li], dr[0]);
      atomicAdd(&dloc[3 * li + 1], dr[1]);
      atomicAdd(&dloc[3 * li + 2], dr[2]);
      atomicAdd(&dscale[ki], dk);
    }
  }
}

extern "C" int vmf3_rsample_fwd(const void* loc, const void* scale, const void* u, const void* eps, void* z,
                                void* w_out, int ndim, const int64_t* odims, const int64_t* w_str,
                                const int64_t* k_str, const int64_t* l_str, const int64_t* e_str, int dtype,
                                uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                vmf_stream_t stream) {
  Bcast bc;
  int64_t n_out = 0;
  int rc = make_bcast(&bc, &n_out, ndim, odims, w_str, k_str, l_str, e_str);
  if (rc) return rc;
  if (n_out == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && z, VMF_ERR_BAD_ARG, "vmf3_rsample_fwd: null loc/scale/z");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  int grid = vmf_grid_for(n_out, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_fwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)loc, (const T*)scale, (const T*)u, (const T*)eps, (T*)z,
                                               (T*)w_out, bc, n_out, key));
  return vmf_check_launch("vmf3_rsample_fwd");
}

extern "C" int vmf3_rsample_bwd(const void* loc, const void* scale, const void* u, const void* eps,
                                const void* grad_z, void* dloc, void* dscale, int64_t n_loc, int64_t n_scale,
                                int ndim, const int64_t* odims, const int64_t* w_str, const int64_t* k_str,
                                const int64_t* l_str, const int64_t* e_str, int dtype, uint64_t seed,
                                uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  Bcast bc;
  int64_t n_out = 0;
  int rc = make_bcast(&bc, &n_out, ndim, odims, w_str, k_str, l_str, e_str);
  if (rc) return rc;
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE(n_loc >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "negative sizes");
  if (n_loc == 0 && n_scale == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && grad_z && dloc && dscale, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd: null pointer");
  size_t es = dtype == VMF_F32 ? 4 : 8;
  // identity also requires that the outputs are exactly as large as the result
  if (bc.identity && !(n_loc == n_out && n_scale == n_out)) bc.identity = 0;
  // Outputs as large as the result although operands are shared: per-row mode.  Row o of dloc / dscale receives the
  // contribution of output row o and the caller adds up the rows that share a loc row / scale element in an order
  // of its choosing -- the run-to-run reproducible alternative to the float atomics below.
  const int per_row = !bc.identity && n_loc == n_out && n_scale == n_out;
  if (!bc.identity && !per_row) {
    if (hipMemsetAsync(dloc, 0, es * 3 * (size_t)n_loc, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(dscale, 0, es * (size_t)n_scale, (hipStream_t)stream) != hipSuccess)
      return vmf_check_launch("vmf3_rsample_bwd memset");
  }
  if (n_out == 0) return VMF_OK;
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  int grid = vmf_grid_for(n_out, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_bwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)loc, (const T*)scale, (const T*)u, (const T*)eps,
                                               (const T*)grad_z, (T*)dloc, (T*)dscale, bc, n_out, per_row,
                                               key));
  return vmf_check_launch("vmf3_rsample_bwd");
}

// ---- _sample_w3 alone (sampler/von_mises_fisher.py:50-64): w = 1 + logsumexp([log u, log(1-u) - 2k]) / k ----------
// u has shape sample_shape + scale.shape, so element i pairs with scale[i mod n_scale].
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void vmf3_w_kernel(const T* __restrict__ scale, const T* __restrict__ u,
                                                     const T* __restrict__ gw, T* __restrict__ out, int64_t n,
                                                     int64_t n_scale, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T k = scale[i % n_scale];
    const T uu = v3_draw_u<T>(u, i, key);
    T L, x2;
    const T w = v3_w<T>(uu, k, &L, &x2);
    if (!BWD) {
      out[i] = w;
    } else {  // d w / d k = -L / k^2 - 2 sigma / k, sigma = exp(x2 - L): per-element term, the caller sums over samples
      const T sig = M<T>::exp(x2 - L);
      out[i] = gw[i] * (-L / (k * k) - (T)2 * sig / k);
    }
  }
}

extern "C" int vmf3_sample_w_fwd(const void* scale, const void* u, void* w, int64_t n, int64_t n_scale, int dtype,
                                 uint64_t seed, uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_REQUIRE(n >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: negative size");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(n_scale > 0 && n % n_scale == 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: n %lld does not tile n_scale %lld",
              (long long)n, (long long)n_scale);
  VMF_REQUIRE(scale && w, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vmf3_w_kernel<T, false>), dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)scale, (const T*)u, (const T*)nullptr,
                                               (T*)w, n, n_scale, key));
  return vmf_check_launch("vmf3_sample_w_fwd");
}

extern "C" int vmf3_sample_w_bwd(const void* scale, const void* u, const void* grad_w, void* dscale_rows, int64_t n,
                                 int64_t n_scale, int dtype, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_REQUIRE(n >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: negative size");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(n_scale > 0 && n % n_scale == 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: n does not tile n_scale");
  VMF_REQUIRE(scale && grad_w && dscale_rows, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vmf3_w_kernel<T, true>), dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)scale, (const T*)u, (const T*)grad_w,
                                               (T*)dscale_rows, n, n_scale, key));
  return vmf_check_launch("vmf3_sample_w_bwd");
}

// ---- log_prob (reference sign) and mean ----------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vmf3_log_prob_ref_kernel(const T* __restrict__ loc,
                                                                const T* __restrict__ scale,
                                                                const T* __restrict__ x, T* __restrict__ logp,
                                                                T* __restrict__ lognorm, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T k = scale[i];
    // CNN_experiment/von_mises_fisher.py:103-108
    T ln = M<T>::log(k) - (T)2.5310242469692907 /* log(4 pi) */ - M<T>::log(M<T>::sinh(k) + (T)1e-10);
    if (lognorm) lognorm[i] = ln;
    if (logp) {
      T r[3] = {loc[3 * i], loc[3 * i + 1], loc[3 * i + 2]};
      T l[3], nn;
      v3_normalise_loc<T>(r, l, &nn);
      T dot = l[0] * x[3 * i] + l[1] * x[3 * i + 1] + l[2] * x[3 * i + 2];
      logp[i] = k * dot - ln;  // :92-95 (the reference subtracts)
    }
  }
}

extern "C" int vmf3_log_prob_ref(const void* loc, const void* scale, const void* x, void* logp, void* lognorm,
                                 int64_t n, int dtype, vmf_stream_t stream) {
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(scale && (!logp || (loc && x)), VMF_ERR_BAD_ARG, "vmf3_log_prob_ref: null pointer");
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_log_prob_ref_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)loc, (const T*)scale, (const T*)x,
                                               (T*)logp, (T*)lognorm, n));
  return vmf_check_launch("vmf3_log_prob_ref");
}

// backward of the above through loc, scale and x (what autograd derives from CNN_experiment/von_mises_fisher.py:92-108)
template <typename T>
__global__ __launch_bounds__(256) void vmf3_log_prob_ref_bwd_kernel(const T* __restrict__ loc,
                                                                    const T* __restrict__ scale,
                                                                    const T* __restrict__ x, const T* __restrict__ gl,
                                                                    T* __restrict__ dloc, T* __restrict__ dscale,
                                                                    T* __restrict__ dx, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T k = scale[i], g = gl[i];
    T r[3] = {loc[3 * i], loc[3 * i + 1], loc[3 * i + 2]};
    T xv[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    T l[3], nn;
    v3_normalise_loc<T>(r, l, &nn);
    const T dot = l[0] * xv[0] + l[1] * xv[1] + l[2] * xv[2];
    if (dscale) {
      // d/dk [k dot - (log k - log 4pi - log(sinh k + 1e-10))] = dot - 1/k + cosh k / (sinh k + 1e-10)
      const T sh = M<T>::sinh(k) + (T)1e-10;
      dscale[i] = g * (dot - (T)1 / k + M<T>::cosh(k) / sh);
    }
    if (dx) {
      dx[3 * i] = g * k * l[0]; dx[3 * i + 1] = g * k * l[1]; dx[3 * i + 2] = g * k * l[2];
    }
    if (dloc) {
      T dl[3] = {g * k * xv[0], g * k * xv[1], g * k * xv[2]};
      const T nden = nn + (T)1e-5;
      const T rdl = r[0] * dl[0] + r[1] * dl[1] + r[2] * dl[2];
      for (int c = 0; c < 3; ++c) dloc[3 * i + c] = dl[c] / nden - r[c] * rdl / (nn * nden * nden);
    }
  }
}

extern "C" int vmf3_log_prob_ref_bwd(const void* loc, const void* scale, const void* x, const void* gl, void* dloc,
                                     void* dscale, void* dx, int64_t n, int dtype, vmf_stream_t stream) {
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && x && gl, VMF_ERR_BAD_ARG, "vmf3_log_prob_ref_bwd: null pointer");
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_log_prob_ref_bwd_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)loc, (const T*)scale, (const T*)x,
                                               (const T*)gl, (T*)dloc, (T*)dscale, (T*)dx, n));
  return vmf_check_launch("vmf3_log_prob_ref_bwd");
}

template <typename T>
__global__ __launch_bounds__(256) void vmf3_mean_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                        T* __restrict__ mean, int64_t n, BesselOrder o32,
                                                        BesselOrder o12) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T r[3] = {loc[3 * i], loc[3 * i + 1], loc[3 * i + 2]};
    T l[3], nn;
    v3_normalise_loc<T>(r, l, &nn);
    T a = (T)ive_eval(o32, (double)scale[i]);
    T b = (T)ive_eval(o12, (double)scale[i]);
    T A = a / b;  // sampler/von_mises_fisher.py:42
    mean[3 * i] = l[0] * A; mean[3 * i + 1] = l[1] * A; mean[3 * i + 2] = l[2] * A;
  }
}

extern "C" int vmf3_mean(const void* loc, const void* scale, void* mean, int64_t n, int dtype,
                         vmf_stream_t stream) {
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && mean, VMF_ERR_BAD_ARG, "vmf3_mean: null pointer");
  BesselOrder o32 = make_bessel_order(1.5, 1), o12 = make_bessel_order(0.5, 1);
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_mean_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)loc, (const T*)scale, (T*)mean, n, o32, o12));
  return vmf_check_launch("vmf3_mean");
}
