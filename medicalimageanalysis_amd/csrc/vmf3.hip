// vmf3.hip -- VonMisesFisher3D on gfx950: fused normalise + inverse-CDF w + tangent + combine, its backward,
// the m = 3 log_prob / mean, and the library's error plumbing.
//
// The reference launches ~25 elementwise kernels and materialises every intermediate
// (sampler/von_mises_fisher.py:30, 50-86).  Here the entry points classify the broadcast of a call: the two shapes
// the reference's models produce -- one row per (loc, scale) pair, and the (B, 1)-scale outer product (B, B, 3) --
// have kernels of their own in vmf3_fast.hip; any other broadcast (sample shapes in front of shared parameters ...)
// takes the thread-per-output-row kernels below.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "vmf3_fast.h"
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

// ---- marker ranges ----------------------------------------------------------------------------
namespace {
typedef int (*range_push_fn)(const char*);
typedef int (*range_pop_fn)(void);
range_push_fn g_range_push = nullptr;
range_pop_fn g_range_pop = nullptr;
std::once_flag g_range_once;

void range_init() {
  const char* e = getenv("VMF_ROCTX");
  if (!e || !*e || !strcmp(e, "0")) return;
  // rocprofv3 intercepts the markers of the SDK's library; the older tracer library is the fall-back
  static const char* const names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                                      "libroctx64.so.4"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    range_push_fn a = (range_push_fn)dlsym(h, "roctxRangePushA");
    range_pop_fn b = (range_pop_fn)dlsym(h, "roctxRangePop");
    if (a && b) {
      g_range_push = a;
      g_range_pop = b;
      return;
    }
  }
  fprintf(stderr, "libvmf_hip: VMF_ROCTX is set but no marker library could be loaded (%s)\n", dlerror());
}
}  // namespace

VmfRange::VmfRange(const char* name) {
  std::call_once(g_range_once, range_init);
  on = g_range_push != nullptr;
  if (on) g_range_push(name);
}
VmfRange::~VmfRange() {
  if (on) g_range_pop();
}

// ---- loss scalars across ranks -------------------------------------------------------------------
// The one exchange of the data-parallel path: a handful of doubles (sum KL, sum reconstruction, sum log p, count)
// summed over the ranks' shards, in place, on the caller's stream.  The collective library is the one the process has
// already loaded (the communicator comes from it); it is looked up at run time for the same reason as the markers.
extern "C" int vmf_allreduce_scalars(void* comm, double* buf, int count, vmf_stream_t stream) {
  VMF_RANGE("vmf_allreduce_scalars");
  VMF_REQUIRE(comm && buf && count > 0 && count <= VMF_ALLREDUCE_MAX_SCALARS, VMF_ERR_BAD_ARG,
              "vmf_allreduce_scalars: communicator and buffer must be non-null, 1 <= count <= %d (got %d)",
              VMF_ALLREDUCE_MAX_SCALARS, count);
  typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  static allreduce_fn fn = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    fn = (allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!fn) {
      void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h) fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
    }
  });
  VMF_REQUIRE(fn, VMF_ERR_UNSUPPORTED, "vmf_allreduce_scalars: ncclAllReduce not found (is librccl loaded?)");
  const int kFloat64 = 8, kSum = 0;  // ncclFloat64, ncclSum
  const int rc = fn(buf, buf, (size_t)count, kFloat64, kSum, comm, (hipStream_t)stream);
  VMF_REQUIRE(rc == 0, VMF_ERR_LAUNCH, "vmf_allreduce_scalars: ncclAllReduce returned %d", rc);
  return VMF_OK;
}

// ---- one ticket of the device-resident noise counter -----------------------------------------------------
__global__ void philox_ticket_kernel(int64_t* __restrict__ counter, int64_t* __restrict__ snapshot) {
  if (threadIdx.x == 0) {
    const int64_t v = *counter;
    *snapshot = v;
    *counter = (int64_t)((uint64_t)v + 1u);
  }
}
extern "C" int vmf_philox_ticket(int64_t* counter, int64_t* snapshot, vmf_stream_t stream) {
  VMF_RANGE("vmf_philox_ticket");
  VMF_REQUIRE(counter && snapshot && counter != snapshot, VMF_ERR_BAD_ARG,
              "vmf_philox_ticket: counter and snapshot must be two device addresses");
  hipLaunchKernelGGL(philox_ticket_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, snapshot);
  return vmf_check_launch("vmf_philox_ticket");
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

// What shape a call has once size-1 dims are dropped and dims that walk all four operands contiguously are merged:
// ROWS (every operand indexed by the output row), OUTER (u / scale indexed by the first dim, loc / eps by the second:
// the (B, 1)-scale broadcast) or anything else (sample shapes in front of shared parameters ...).
enum { V3_KIND_GENERAL = 0, V3_KIND_ROWS = 1, V3_KIND_OUTER = 2 };
static int bcast_kind(const Bcast& bc, int64_t* n_i, int64_t* n_j) {
  int64_t d[8], w[8], k[8], l[8], e[8];
  int nd = 0;
  for (int x = 0; x < bc.ndim; ++x) {
    if (bc.odims[x] == 1) continue;
    d[nd] = bc.odims[x]; w[nd] = bc.w_str[x]; k[nd] = bc.k_str[x]; l[nd] = bc.l_str[x]; e[nd] = bc.e_str[x];
    ++nd;
  }
  for (int x = nd - 2; x >= 0; --x) {  // merge dim x+1 into dim x where all four operands allow it
    const int64_t m = d[x + 1];
    if (w[x] == w[x + 1] * m && k[x] == k[x + 1] * m && l[x] == l[x + 1] * m && e[x] == e[x + 1] * m) {
      d[x] *= m; w[x] = w[x + 1]; k[x] = k[x + 1]; l[x] = l[x + 1]; e[x] = e[x + 1];
      for (int y = x + 1; y < nd - 1; ++y) { d[y] = d[y + 1]; w[y] = w[y + 1]; k[y] = k[y + 1]; l[y] = l[y + 1]; e[y] = e[y + 1]; }
      --nd;
    }
  }
  if (nd == 0) { *n_i = 1; *n_j = 1; return V3_KIND_ROWS; }
  if (nd == 1 && w[0] == 1 && k[0] == 1 && l[0] == 1 && e[0] == 1) { *n_i = d[0]; *n_j = d[0]; return V3_KIND_ROWS; }
  if (nd == 2 && w[0] == 1 && k[0] == 1 && l[0] == 0 && e[0] == 0 && w[1] == 0 && k[1] == 0 && l[1] == 1 && e[1] == 1) {
    *n_i = d[0]; *n_j = d[1];
    return V3_KIND_OUTER;
  }
  *n_i = *n_j = 0;
  return V3_KIND_GENERAL;
}

extern "C" int vmf3_broadcast_kind(int ndim, const int64_t* odims, const int64_t* w_str, const int64_t* k_str,
                                   const int64_t* l_str, const int64_t* e_str, int64_t* n_i, int64_t* n_j) {
  Bcast bc;
  int64_t n_out = 0, a = 0, b = 0;
  int rc = make_bcast(&bc, &n_out, ndim, odims, w_str, k_str, l_str, e_str);
  if (rc) return rc;
  const int kind = n_out == 0 ? V3_KIND_GENERAL : bcast_kind(bc, &a, &b);
  if (n_i) *n_i = a;
  if (n_j) *n_j = b;
  return kind;
}

// The general kernel: one thread per output row, operands found through the broadcast descriptor.
template <typename T, typename MP>
__global__ __launch_bounds__(256) void vmf3_fwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                       const T* __restrict__ u, const T* __restrict__ eps,
                                                       T* __restrict__ z, T* __restrict__ w_out,
                                                       T* __restrict__ kappa_out, Bcast bc, int64_t n_out,
                                                       int softplus, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
    int64_t wi, ki, li, ei;
    unravel(bc, o, &wi, &ki, &li, &ei);
    T r[3] = {loc[3 * li], loc[3 * li + 1], loc[3 * li + 2]};
    T sig;
    const T k = softplus ? v3_softplus_kappa<T, MP>(scale[ki], &sig) : scale[ki];
    T uu = v3_draw_u<T>(u, wi, key);
    T e[3];
    v3_draw_eps<T>(eps, ei, key, e);
    T l[3], t[3], v[3], n, p, s, L, x2;
    bool inside;
    v3_normalise_loc<T, MP>(r, l, &n);
    T w = v3_w<T, MP>(uu, k, &L, &x2);
    v3_tangent<T, MP>(e, l, t, v, &p, &s);
    T f = v3_factor<T, MP>(w, &inside);
    z[3 * o] = w * l[0] + f * v[0];
    z[3 * o + 1] = w * l[1] + f * v[1];
    z[3 * o + 2] = w * l[2] + f * v[2];
    if (w_out) w_out[wi] = w;
    if (kappa_out) kappa_out[ki] = k;  // rows that share a scale element write the same value
  }
}

template <typename T, typename MP>
__global__ __launch_bounds__(256) void vmf3_bwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                       const T* __restrict__ u, const T* __restrict__ eps,
                                                       const T* __restrict__ gz, T* __restrict__ dloc,
                                                       T* __restrict__ dscale, Bcast bc, int64_t n_out,
                                                       int per_row, int softplus, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
    int64_t wi, ki, li, ei;
    unravel(bc, o, &wi, &ki, &li, &ei);
    T r[3] = {loc[3 * li], loc[3 * li + 1], loc[3 * li + 2]};
    T sig = (T)1;
    const T k = softplus ? v3_softplus_kappa<T, MP>(scale[ki], &sig) : scale[ki];
    T uu = v3_draw_u<T>(u, wi, key);
    T e[3];
    v3_draw_eps<T>(eps, ei, key, e);
    T g[3] = {gz[3 * o], gz[3 * o + 1], gz[3 * o + 2]};
    T dr[3], dk;
    v3_backward_row<T, MP>(r, k, uu, e, g, dr, &dk);
    dk *= sig;
    if (per_row) {  // one gradient row per output row; the caller sums the rows that share an operand
      dloc[3 * o] = dr[0]; dloc[3 * o + 1] = dr[1]; dloc[3 * o + 2] = dr[2];
      dscale[o] = dk;
    } else {
      atomicAdd(&dloc[3 * li], dr[0]);
      atomicAdd(&dloc[3 * li + 1], dr[1]);
      atomicAdd(&dloc[3 * li + 2], dr[2]);
      atomicAdd(&dscale[ki], dk);
    }
  }
}

static V3Call make_call(const void* loc, const void* scale, const void* u, const void* eps, int flags, int dtype,
                        uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  V3Call c = {};
  c.loc = loc; c.scale = scale; c.u = u; c.eps = eps;
  c.softplus = (flags & VMF_FLAG_SOFTPLUS_KAPPA) ? 1 : 0;
  c.dtype = dtype;
  c.ka = make_philox_arg(seed, offset, offset_dev, row0);
  c.stream = (hipStream_t)stream;
  return c;
}

extern "C" int vmf3_rsample_fwd(const void* loc, const void* scale, const void* u, const void* eps, void* z,
                                void* w_out, void* kappa_out, int ndim, const int64_t* odims, const int64_t* w_str,
                                const int64_t* k_str, const int64_t* l_str, const int64_t* e_str, int flags, int dtype,
                                uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_RANGE("vmf3_rsample_fwd");
  Bcast bc;
  int64_t n_out = 0;
  int rc = make_bcast(&bc, &n_out, ndim, odims, w_str, k_str, l_str, e_str);
  if (rc) return rc;
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d is not VMF_F32/VMF_F64", dtype);
  VMF_REQUIRE((flags & ~VMF_FLAG_SOFTPLUS_KAPPA) == 0, VMF_ERR_BAD_ARG, "vmf3_rsample_fwd: unknown flags %d", flags);
  if (n_out == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && z, VMF_ERR_BAD_ARG, "vmf3_rsample_fwd: null loc/scale/z");
  V3Call c = make_call(loc, scale, u, eps, flags, dtype, row0, seed, offset, offset_dev, stream);
  c.z = z; c.w_out = w_out; c.kappa_out = kappa_out;
  int64_t n_i = 0, n_j = 0;
  const int kind = bcast_kind(bc, &n_i, &n_j);
  if (kind == V3_KIND_ROWS) {
    rc = vmf3_fwd_rows(c, n_out);
  } else if (kind == V3_KIND_OUTER) {
    rc = vmf3_fwd_outer(c, n_i, n_j);
  } else {
    int grid = vmf_grid_for(n_out, 256);
    if (dtype == VMF_F32)
      hipLaunchKernelGGL((vmf3_fwd_kernel<float, MFast>), dim3(grid), dim3(256), 0, c.stream, (const float*)loc,
                         (const float*)scale, (const float*)u, (const float*)eps, (float*)z, (float*)w_out,
                         (float*)kappa_out, bc, n_out, c.softplus, c.ka);
    else
      hipLaunchKernelGGL((vmf3_fwd_kernel<double, M<double>>), dim3(grid), dim3(256), 0, c.stream, (const double*)loc,
                         (const double*)scale, (const double*)u, (const double*)eps, (double*)z, (double*)w_out,
                         (double*)kappa_out, bc, n_out, c.softplus, c.ka);
  }
  if (rc) return rc;
  return vmf_check_launch("vmf3_rsample_fwd");
}

extern "C" int vmf3_rsample_bwd(const void* loc, const void* scale, const void* u, const void* eps,
                                const void* grad_z, const void* grad_kappa, void* dloc, void* dscale, int64_t n_loc,
                                int64_t n_scale, int ndim, const int64_t* odims, const int64_t* w_str,
                                const int64_t* k_str, const int64_t* l_str, const int64_t* e_str, int flags, int dtype,
                                uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_RANGE("vmf3_rsample_bwd");
  Bcast bc;
  int64_t n_out = 0;
  int rc = make_bcast(&bc, &n_out, ndim, odims, w_str, k_str, l_str, e_str);
  if (rc) return rc;
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE((flags & ~VMF_FLAG_SOFTPLUS_KAPPA) == 0, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd: unknown flags %d", flags);
  VMF_REQUIRE(n_loc >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "negative sizes");
  if (n_loc == 0 && n_scale == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && grad_z && dloc && dscale, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd: null pointer");
  V3Call c = make_call(loc, scale, u, eps, flags, dtype, row0, seed, offset, offset_dev, stream);
  c.gz = grad_z; c.gkappa = grad_kappa; c.dloc = dloc; c.dscale = dscale;
  int64_t n_i = 0, n_j = 0;
  const int kind = n_out ? bcast_kind(bc, &n_i, &n_j) : V3_KIND_GENERAL;
  if (kind == V3_KIND_ROWS && n_loc == n_out && n_scale == n_out) {
    rc = vmf3_bwd_rows(c, n_out);
    if (rc) return rc;
    return vmf_check_launch("vmf3_rsample_bwd");
  }
  VMF_REQUIRE(!grad_kappa, VMF_ERR_UNSUPPORTED,
              "vmf3_rsample_bwd: grad_kappa is taken by the row-per-row shape (and by vmf3_rsample_bwd_outer) only");
  size_t es = dtype == VMF_F32 ? 4 : 8;
  // Outputs as large as the result although operands are shared: per-row mode.  Row o of dloc / dscale receives the
  // contribution of output row o and the caller adds up the rows that share a loc row / scale element in an order
  // of its choosing -- the run-to-run reproducible alternative to the float atomics below.
  const int per_row = n_loc == n_out && n_scale == n_out;
  if (!per_row) {
    if (hipMemsetAsync(dloc, 0, es * 3 * (size_t)n_loc, c.stream) != hipSuccess ||
        hipMemsetAsync(dscale, 0, es * (size_t)n_scale, c.stream) != hipSuccess)
      return vmf_check_launch("vmf3_rsample_bwd memset");
  }
  if (n_out == 0) return VMF_OK;
  int grid = vmf_grid_for(n_out, 256);
  if (dtype == VMF_F32)
    hipLaunchKernelGGL((vmf3_bwd_kernel<float, MFast>), dim3(grid), dim3(256), 0, c.stream, (const float*)loc,
                       (const float*)scale, (const float*)u, (const float*)eps, (const float*)grad_z, (float*)dloc,
                       (float*)dscale, bc, n_out, per_row, c.softplus, c.ka);
  else
    hipLaunchKernelGGL((vmf3_bwd_kernel<double, M<double>>), dim3(grid), dim3(256), 0, c.stream, (const double*)loc,
                       (const double*)scale, (const double*)u, (const double*)eps, (const double*)grad_z,
                       (double*)dloc, (double*)dscale, bc, n_out, per_row, c.softplus, c.ka);
  return vmf_check_launch("vmf3_rsample_bwd");
}

extern "C" int vmf3_rsample_bwd_outer_workspace(int64_t n_i, int64_t n_j, int dtype, int64_t* bytes_out) {
  VMF_REQUIRE(n_i >= 0 && n_j >= 0 && bytes_out, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd_outer_workspace: bad argument");
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  *bytes_out = vmf3_bwd_outer_workspace_bytes(n_i, n_j, dtype);
  return VMF_OK;
}

extern "C" int vmf3_rsample_bwd_outer(const void* loc, const void* scale, const void* u, const void* eps,
                                      const void* grad_z, const void* grad_kappa, void* dloc, void* dscale,
                                      int64_t n_i, int64_t n_j, void* workspace, int64_t workspace_bytes, int flags,
                                      int dtype, uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                      vmf_stream_t stream) {
  VMF_RANGE("vmf3_rsample_bwd_outer");
  VMF_REQUIRE(n_i >= 0 && n_j >= 0, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd_outer: negative size");
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE((flags & ~VMF_FLAG_SOFTPLUS_KAPPA) == 0, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd_outer: unknown flags %d", flags);
  if (n_i == 0 && n_j == 0) return VMF_OK;
  VMF_REQUIRE(n_i > 0 && n_j > 0, VMF_ERR_UNSUPPORTED, "vmf3_rsample_bwd_outer: an empty dimension next to a non-empty "
              "one (zero-fill the gradients instead)");
  VMF_REQUIRE(loc && scale && grad_z && dloc && dscale, VMF_ERR_BAD_ARG, "vmf3_rsample_bwd_outer: null pointer");
  V3Call c = make_call(loc, scale, u, eps, flags, dtype, row0, seed, offset, offset_dev, stream);
  c.gz = grad_z; c.gkappa = grad_kappa; c.dloc = dloc; c.dscale = dscale;
  int rc = vmf3_bwd_outer(c, n_i, n_j, workspace, workspace_bytes);
  if (rc) return rc;
  return vmf_check_launch("vmf3_rsample_bwd_outer");
}

// ---- _sample_w3 alone (sampler/von_mises_fisher.py:50-64): w = 1 + logsumexp([log u, log(1-u) - 2k]) / k ----------
// u has shape sample_shape + scale.shape, so element i pairs with scale[i mod n_scale].
template <typename T, typename MP, bool BWD>
__global__ __launch_bounds__(256) void vmf3_w_kernel(const T* __restrict__ scale, const T* __restrict__ u,
                                                     const T* __restrict__ gw, T* __restrict__ out, int64_t n,
                                                     int64_t n_scale, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T k = scale[i % n_scale];
    const T uu = v3_draw_u<T>(u, i, key);
    T L, x2;
    const T w = v3_w<T, MP>(uu, k, &L, &x2);
    if (!BWD) {
      out[i] = w;
    } else {  // d w / d k = -L / k^2 - 2 sigma / k, sigma = exp(x2 - L): per-element term, the caller sums over samples
      const T sig = MP::exp(x2 - L);
      out[i] = gw[i] * (-MP::div(L, k * k) - (T)2 * MP::div(sig, k));
    }
  }
}

extern "C" int vmf3_sample_w_fwd(const void* scale, const void* u, void* w, int64_t n, int64_t n_scale, int dtype,
                                 uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_RANGE("vmf3_sample_w_fwd");
  VMF_REQUIRE(n >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: negative size");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(n_scale > 0 && n % n_scale == 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: n %lld does not tile n_scale %lld",
              (long long)n, (long long)n_scale);
  VMF_REQUIRE(scale && w, VMF_ERR_BAD_ARG, "vmf3_sample_w_fwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev, row0);
  int grid = vmf_grid_for(n, 256);
  // fp32 runs the arithmetic of the fp32 rsample kernels (hardware transcendentals), so both give the same w
  if (dtype == VMF_F32)
    hipLaunchKernelGGL((vmf3_w_kernel<float, MFast, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scale, (const float*)u, (const float*)nullptr, (float*)w, n, n_scale, key);
  else if (dtype == VMF_F64)
    hipLaunchKernelGGL((vmf3_w_kernel<double, M<double>, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const double*)scale, (const double*)u, (const double*)nullptr, (double*)w, n, n_scale, key);
  else
    VMF_REQUIRE(false, VMF_ERR_DTYPE, "dtype %d is not VMF_F32/VMF_F64", dtype);
  return vmf_check_launch("vmf3_sample_w_fwd");
}

extern "C" int vmf3_sample_w_bwd(const void* scale, const void* u, const void* grad_w, void* dscale_rows, int64_t n,
                                 int64_t n_scale, int dtype, uint64_t row0, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_dev, vmf_stream_t stream) {
  VMF_RANGE("vmf3_sample_w_bwd");
  VMF_REQUIRE(n >= 0 && n_scale >= 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: negative size");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(n_scale > 0 && n % n_scale == 0, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: n does not tile n_scale");
  VMF_REQUIRE(scale && grad_w && dscale_rows, VMF_ERR_BAD_ARG, "vmf3_sample_w_bwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev, row0);
  int grid = vmf_grid_for(n, 256);
  if (dtype == VMF_F32)
    hipLaunchKernelGGL((vmf3_w_kernel<float, MFast, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scale, (const float*)u, (const float*)grad_w, (float*)dscale_rows, n, n_scale, key);
  else if (dtype == VMF_F64)
    hipLaunchKernelGGL((vmf3_w_kernel<double, M<double>, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const double*)scale, (const double*)u, (const double*)grad_w, (double*)dscale_rows, n, n_scale,
                       key);
  else
    VMF_REQUIRE(false, VMF_ERR_DTYPE, "dtype %d is not VMF_F32/VMF_F64", dtype);
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
    const T k = scale[i];
    T ln;
    if (logp) {
      const T r[3] = {loc[3 * i], loc[3 * i + 1], loc[3 * i + 2]};
      const T xv[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
      logp[i] = v3_log_prob_ref_row<T>(r, k, xv, &ln);
    } else {
      v3_log_norm_ref<T>(k, &ln, (T*)nullptr);  // CNN_experiment/von_mises_fisher.py:103-108
    }
    if (lognorm) lognorm[i] = ln;
  }
}

extern "C" int vmf3_log_prob_ref(const void* loc, const void* scale, const void* x, void* logp, void* lognorm,
                                 int64_t n, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf3_log_prob_ref");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(scale && (!logp || (loc && x)), VMF_ERR_BAD_ARG, "vmf3_log_prob_ref: null pointer");
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  // whole tiles first, the remaining rows (and unaligned calls) one thread per row
  const int64_t done = vmf3_log_prob_ref_tiles(logp ? loc : nullptr, scale, logp ? x : nullptr, logp, lognorm, n, dtype,
                                               (hipStream_t)stream);
  if (done < n) {
    const int64_t es = dtype == VMF_F32 ? 4 : 8, rest = n - done;
    const auto at = [&](const void* q, int64_t per_row) {
      return q ? (const void*)((const char*)q + done * per_row * es) : nullptr;
    };
    int grid = vmf_grid_for(rest, 256);
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_log_prob_ref_kernel<T>, dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)at(loc, 3), (const T*)at(scale, 1),
                                                 (const T*)at(x, 3), (T*)at(logp, 1), (T*)at(lognorm, 1), rest));
  }
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
    T dl[3], dk, dxr[3];
    v3_log_prob_ref_bwd_row<T>(r, k, xv, g, dl, &dk, dxr);
    if (dscale) dscale[i] = dk;
    if (dx) {
      dx[3 * i] = dxr[0]; dx[3 * i + 1] = dxr[1]; dx[3 * i + 2] = dxr[2];
    }
    if (dloc) {
      dloc[3 * i] = dl[0]; dloc[3 * i + 1] = dl[1]; dloc[3 * i + 2] = dl[2];
    }
  }
}

extern "C" int vmf3_log_prob_ref_bwd(const void* loc, const void* scale, const void* x, const void* gl, void* dloc,
                                     void* dscale, void* dx, int64_t n, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf3_log_prob_ref_bwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && x && gl, VMF_ERR_BAD_ARG, "vmf3_log_prob_ref_bwd: null pointer");
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  const int64_t done = vmf3_log_prob_ref_bwd_tiles(loc, scale, x, gl, dloc, dscale, dx, n, dtype, (hipStream_t)stream);
  if (done < n) {
    const int64_t es = dtype == VMF_F32 ? 4 : 8, rest = n - done;
    const auto at = [&](const void* q, int64_t per_row) {
      return q ? (const void*)((const char*)q + done * per_row * es) : nullptr;
    };
    int grid = vmf_grid_for(rest, 256);
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_log_prob_ref_bwd_kernel<T>, dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)at(loc, 3), (const T*)at(scale, 1),
                                                 (const T*)at(x, 3), (const T*)at(gl, 1), (T*)at(dloc, 3),
                                                 (T*)at(dscale, 1), (T*)at(dx, 3), rest));
  }
  return vmf_check_launch("vmf3_log_prob_ref_bwd");
}

template <typename T>
__global__ __launch_bounds__(256) void vmf3_mean_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                        T* __restrict__ mean, int64_t n, BesselOrder o32,
                                                        BesselOrder o12) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T r[3] = {loc[3 * i], loc[3 * i + 1], loc[3 * i + 2]};
    T o[3];
    v3_mean_row<T>(r, scale[i], o32, o12, o);  // sampler/von_mises_fisher.py:42
    mean[3 * i] = o[0]; mean[3 * i + 1] = o[1]; mean[3 * i + 2] = o[2];
  }
}

extern "C" int vmf3_mean(const void* loc, const void* scale, void* mean, int64_t n, int dtype,
                         vmf_stream_t stream) {
  VMF_RANGE("vmf3_mean");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(loc && scale && mean, VMF_ERR_BAD_ARG, "vmf3_mean: null pointer");
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  BesselOrder o32 = make_bessel_order(1.5, 1), o12 = make_bessel_order(0.5, 1);
  // whole tiles of 256 rows when the pointers allow 16-byte accesses, the thread-per-row kernel for the rest
  const int64_t done = vmf3_mean_tiles(loc, scale, mean, n, dtype, (hipStream_t)stream);
  if (done < n) {
    const int64_t es = dtype == VMF_F32 ? 4 : 8, rest = n - done;
    int grid = vmf_grid_for(rest, 256);
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf3_mean_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                 (const T*)((const char*)loc + 3 * done * es),
                                                 (const T*)((const char*)scale + done * es),
                                                 (T*)((char*)mean + 3 * done * es), rest, o32, o12));
  }
  return vmf_check_launch("vmf3_mean");
}
