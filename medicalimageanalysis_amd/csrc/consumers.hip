// consumers.hip -- the steps on either side of the sampler in the reference's train step, each as one pass:
//   * the ELBO (reconstruction MSE + KL mean) as one fused reduction        model/spherical_vae.py:119-138
//   * the torus embedding of two sphere draws                               model/torus_vae.py:60-81
//   * the supervised latent (geodesic) losses on S^1, S^2, T^2               dataloader/utils.py:101-180
// All are HBM streams: x_hat (and z) are read once, sums are carried in double, the finish is one small block.
#include "consumers_math.h"
#include "vmf_math.h"

typedef float v4f __attribute__((ext_vector_type(4)));
// 128-bit accesses of the fp32 fast path; streamed operands bypass the caches' retention (nontemporal)
__device__ __forceinline__ void load4(const float* p, float* dst) {
  v4f v = *reinterpret_cast<const v4f*>(p);
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}
__device__ __forceinline__ void load4_stream(const float* p, float* dst) {
  v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}
__device__ __forceinline__ void store4_stream(float* p, const float* src) {
  v4f v = {src[0], src[1], src[2], src[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}
// never instantiated for double (the vector path is fp32 only); present so both branches type-check
__device__ __forceinline__ void load4(const double*, double*) {}
__device__ __forceinline__ void load4_stream(const double*, double*) {}
__device__ __forceinline__ void store4_stream(double*, const double*) {}

// ---- ELBO -------------------------------------------------------------------------------------------------
// loss = a_rec * sum_e (x[e mod n_x] - x_hat[e])^2 + a_kl * sum_i KL(kappa_i)
// x is broadcast over the leading dimensions of x_hat: with kappa of shape (B, 1) the reference's sample is
// (B, B, 3), the decoder output (B, B, D), and `x - x_mu` repeats x (B, D) along the first axis.
template <typename T, int VEC, bool BESSEL>
__global__ __launch_bounds__(256) void elbo_fwd_kernel(const T* __restrict__ x, const T* __restrict__ xhat,
                                                       const T* __restrict__ kappa, double* __restrict__ partial,
                                                       int64_t n_x, int64_t n_hat, int64_t n_kappa, int variant,
                                                       KlBessel kb) {
  __shared__ double red[4];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double acc_rec = 0, acc_kl = 0;
  {
    const int64_t step = nthreads * VEC;
    const int64_t step_mod = step % n_x;
    int64_t e = tid * VEC;
    int64_t xi = e % n_x;
    for (; e < n_hat; e += step) {
      T a[VEC], b[VEC];
      if (VEC == 4) {
        load4(x + xi, a);
        load4_stream(xhat + e, b);
      } else {
        a[0] = x[xi];
        b[0] = xhat[e];
      }
      T s = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        T d = a[j] - b[j];
        s += d * d;
      }
      acc_rec += (double)s;
      xi += step_mod;
      if (xi >= n_x) xi -= n_x;
    }
  }
  for (int64_t i = tid; i < n_kappa; i += nthreads) {
    acc_kl += (double)kl_eval_value<T, BESSEL>(kappa[i], variant, kb);
  }
  double t0 = block_sum_double(acc_rec, red);
  __syncthreads();
  double t1 = block_sum_double(acc_kl, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = t0;
    partial[gridDim.x + blockIdx.x] = t1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void elbo_final_kernel(const double* __restrict__ partial, int nblk, double a_rec,
                                                         double a_kl, T* __restrict__ loss,
                                                         double* __restrict__ parts) {
  __shared__ double red[4];
  double r = 0, k = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    r += partial[i];
    k += partial[nblk + i];
  }
  double t0 = block_sum_double(r, red);
  __syncthreads();
  double t1 = block_sum_double(k, red);
  if (threadIdx.x == 0) {
    // the two terms are rounded to T before they are combined, as gamma * recon + beta * KL on T tensors is
    *loss = (T)(a_rec * t0) + (T)(a_kl * t1);
    if (parts) {
      parts[0] = a_rec * t0;
      parts[1] = a_kl * t1;
    }
  }
}

template <typename T, int VEC, bool BESSEL>
__global__ __launch_bounds__(256) void elbo_bwd_kernel(const T* __restrict__ x, const T* __restrict__ xhat,
                                                       const T* __restrict__ kappa, const T* __restrict__ gout,
                                                       T a_rec, T a_kl, T* __restrict__ d_xhat,
                                                       T* __restrict__ d_kappa, int64_t n_x, int64_t n_hat,
                                                       int64_t n_kappa, int variant, KlBessel kb) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const T go = gout[0];
  if (d_xhat) {
    const T c = (T)2 * a_rec * go;
    const int64_t step = nthreads * VEC;
    const int64_t step_mod = step % n_x;
    int64_t e = tid * VEC;
    int64_t xi = e % n_x;
    for (; e < n_hat; e += step) {
      T a[VEC], b[VEC], o[VEC];
      if (VEC == 4) {
        load4(x + xi, a);
        load4_stream(xhat + e, b);
      } else {
        a[0] = x[xi];
        b[0] = xhat[e];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = c * (b[j] - a[j]);
      if (VEC == 4) store4_stream(d_xhat + e, o);
      else d_xhat[e] = o[0];
      xi += step_mod;
      if (xi >= n_x) xi -= n_x;
    }
  }
  if (d_kappa) {
    const T c = a_kl * go;
    for (int64_t i = tid; i < n_kappa; i += nthreads) {
      T g;
      kl_eval<T, BESSEL>(kappa[i], variant, kb, &g);
      d_kappa[i] = c * g;
    }
  }
}

// d x_hat for large tensors: the grid covers the tensor once -- a wave owns EB_TILE contiguous elements of x_hat /
// d x_hat, every load issued before the stores -- and tiles are dealt so that an XCD works on one contiguous eighth of
// the tensor.  A 1:1 read/write stream moves faster in this shape than as a grid-stride walk (plain copies of 4 GiB on
// this card: 4.6-5.0 TB/s grid-stride, 6.2-6.4 as one contiguous tile per wave).  (1024, 1024, 784), ms per call:
// grid-stride 1.300; 16 KiB per wave in four rounds 1.186, 4 KiB 1.130, 2 KiB 1.074 (6.1 TB/s), 1 KiB 1.074.
// x is indexed modulo n_x (it stays in L2).
#ifndef VMF_EB_MIN_ELEMS
#define VMF_EB_MIN_ELEMS (1 << 25)
#endif
#ifndef VMF_EB_U
#define VMF_EB_U 2
#endif
#ifndef VMF_EB_ROUNDS
#define VMF_EB_ROUNDS 1
#endif
constexpr int EB_U = VMF_EB_U, EB_ROUNDS = VMF_EB_ROUNDS, EB_TILE = 64 * 4 * EB_U * EB_ROUNDS;
template <bool BESSEL>
__global__ __launch_bounds__(256) void elbo_bwd_tile_kernel(const float* __restrict__ x, const float* __restrict__ xhat,
                                                            const float* __restrict__ kappa,
                                                            const float* __restrict__ gout, float a_rec, float a_kl,
                                                            float* __restrict__ d_xhat, float* __restrict__ d_kappa,
                                                            int64_t n_x, int64_t n_hat, int64_t n_kappa, int variant,
                                                            KlBessel kb) {
  const float go = gout[0];
  const float c = 2.0f * a_rec * go;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)xcd_block(blockIdx.x, gridDim.x) * 4 + wave;
  const int64_t e0 = tile * EB_TILE;
  if (e0 < n_hat) {
    const int64_t x0 = e0 % n_x;       // wave-uniform
    const bool wide = n_x >= EB_TILE;  // then one conditional subtraction wraps any offset inside the tile
#pragma unroll
    for (int r = 0; r < EB_ROUNDS; ++r) {
      float a[EB_U][4], b[EB_U][4];
#pragma unroll
      for (int u = 0; u < EB_U; ++u) {
        const int off = ((r * EB_U + u) * 64 + lane) * 4;
        if (e0 + off < n_hat) {
          int64_t xi = x0 + off;
          if (wide) xi = xi >= n_x ? xi - n_x : xi;
          else xi %= n_x;
          load4_stream(xhat + e0 + off, b[u]);
          load4(x + xi, a[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < EB_U; ++u) {
        const int off = ((r * EB_U + u) * 64 + lane) * 4;
        if (e0 + off < n_hat) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = c * (b[u][j] - a[u][j]);
          store4_stream(d_xhat + e0 + off, o);
        }
      }
    }
  }
  if (d_kappa) {
    const float ck = a_kl * go;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kappa; i += nthreads) {
      float g;
      kl_eval<float, BESSEL>(kappa[i], variant, kb, &g);
      d_kappa[i] = ck * g;
    }
  }
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int vmf_elbo_fwd(const void* x, const void* x_hat, const void* kappa, void* loss_out, double* parts_out,
                            double* workspace, double a_rec, double a_kl, int64_t n_x, int64_t n_hat,
                            int64_t n_kappa, int variant, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_elbo_fwd");
  VMF_REQUIRE(n_x > 0 && n_hat >= 0 && n_kappa >= 0, VMF_ERR_BAD_ARG, "vmf_elbo_fwd: bad sizes");
  VMF_REQUIRE(n_hat % n_x == 0, VMF_ERR_BAD_ARG, "vmf_elbo_fwd: n_hat %lld is not a multiple of n_x %lld",
              (long long)n_hat, (long long)n_x);
  VMF_REQUIRE(variant >= 0 && variant <= VMF_KL_STABLE_M2, VMF_ERR_BAD_ARG, "unknown KL variant %d", variant);
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE(x && loss_out && workspace && (n_hat == 0 || x_hat) && (n_kappa == 0 || kappa), VMF_ERR_BAD_ARG,
              "vmf_elbo_fwd: null pointer");
  KlBessel kb = make_kl_bessel();
  const int es = dtype == VMF_F32 ? 4 : 8;
  const bool vec = (n_x % 4 == 0) && aligned16(x) && aligned16(x_hat) && es == 4;
  int64_t work = vec ? n_hat / 4 : n_hat;
  if (work < n_kappa) work = n_kappa;
  const int grid = vmf_grid_for(work, 256, VMF_ELBO_WORKSPACE_DOUBLES / 2);
  if (vec) {
    // fp32: the instantiation without the double-precision Bessel code (a third of the registers)
    hipLaunchKernelGGL((elbo_fwd_kernel<float, 4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (const float*)x_hat, (const float*)kappa, workspace, n_x, n_hat, n_kappa,
                       variant, kb);
  } else {
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((elbo_fwd_kernel<T, 1, true>), dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)x, (const T*)x_hat,
                                                 (const T*)kappa, workspace, n_x, n_hat, n_kappa, variant, kb));
  }
  int rc = vmf_check_launch("vmf_elbo_fwd");
  if (rc) return rc;
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(elbo_final_kernel<T>, dim3(1), dim3(256), 0, (hipStream_t)stream,
                                               workspace, grid, a_rec, a_kl, (T*)loss_out, parts_out));
  return vmf_check_launch("vmf_elbo_fwd final");
}

extern "C" int vmf_elbo_bwd(const void* x, const void* x_hat, const void* kappa, const void* grad_loss,
                            void* d_x_hat, void* d_kappa, double a_rec, double a_kl, int64_t n_x, int64_t n_hat,
                            int64_t n_kappa, int variant, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_elbo_bwd");
  VMF_REQUIRE(n_x > 0 && n_hat >= 0 && n_kappa >= 0, VMF_ERR_BAD_ARG, "vmf_elbo_bwd: bad sizes");
  VMF_REQUIRE(n_hat % n_x == 0, VMF_ERR_BAD_ARG, "vmf_elbo_bwd: n_hat is not a multiple of n_x");
  VMF_REQUIRE(variant >= 0 && variant <= VMF_KL_STABLE_M2, VMF_ERR_BAD_ARG, "unknown KL variant %d", variant);
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE(grad_loss && (!d_x_hat || (x && x_hat)) && (!d_kappa || kappa), VMF_ERR_BAD_ARG,
              "vmf_elbo_bwd: null pointer");
  if ((!d_x_hat || n_hat == 0) && (!d_kappa || n_kappa == 0)) return VMF_OK;
  KlBessel kb = make_kl_bessel();
  const int es = dtype == VMF_F32 ? 4 : 8;
  const bool vec = (n_x % 4 == 0) && aligned16(x) && aligned16(x_hat) && aligned16(d_x_hat) && es == 4;
  int64_t work = d_x_hat ? (vec ? n_hat / 4 : n_hat) : 0;
  if (d_kappa && work < n_kappa) work = n_kappa;
  const int grid = vmf_grid_for(work, 256);
  // enough tiles to fill the card several times over: one contiguous tile per wave
  const int64_t tiles = (n_hat + EB_TILE - 1) / EB_TILE;
  if (vec && d_x_hat && n_hat >= VMF_EB_MIN_ELEMS && (tiles + 3) / 4 < 0x7fffffff) {
    hipLaunchKernelGGL((elbo_bwd_tile_kernel<false>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)x, (const float*)x_hat, (const float*)kappa,
                       (const float*)grad_loss, (float)a_rec, (float)a_kl, (float*)d_x_hat, (float*)d_kappa, n_x, n_hat,
                       n_kappa, variant, kb);
  } else if (vec) {
    hipLaunchKernelGGL((elbo_bwd_kernel<float, 4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (const float*)x_hat, (const float*)kappa, (const float*)grad_loss, (float)a_rec,
                       (float)a_kl, (float*)d_x_hat, (float*)d_kappa, n_x, n_hat, n_kappa, variant, kb);
  } else {
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((elbo_bwd_kernel<T, 1, true>), dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)x, (const T*)x_hat,
                                                 (const T*)kappa, (const T*)grad_loss, (T)a_rec, (T)a_kl,
                                                 (T*)d_x_hat, (T*)d_kappa, n_x, n_hat, n_kappa, variant, kb));
  }
  return vmf_check_launch("vmf_elbo_bwd");
}

// ---- torus embedding --------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void build_torus_kernel(const T* __restrict__ zt, const T* __restrict__ zp,
                                                          const T* __restrict__ gout, T* __restrict__ out,
                                                          T* __restrict__ gt, T* __restrict__ gp, int64_t n,
                                                          int zdim, T R, T r) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T ct = zt[i * zdim], st = zt[i * zdim + 1], cp = zp[i * zdim], sp = zp[i * zdim + 1];
    if (!BWD) {
      T o[3];
      torus_point<T>(ct, st, cp, sp, R, r, o);
      out[i * 3] = o[0];
      out[i * 3 + 1] = o[1];
      out[i * 3 + 2] = o[2];
    } else {
      T g[3] = {gout[i * 3], gout[i * 3 + 1], gout[i * 3 + 2]}, a[2], b[2];
      torus_point_bwd<T>(ct, cp, sp, R, r, g, a, b);
      gt[i * zdim] = a[0];
      gt[i * zdim + 1] = a[1];
      gp[i * zdim] = b[0];
      gp[i * zdim + 1] = b[1];
      for (int c = 2; c < zdim; ++c) {
        gt[i * zdim + c] = (T)0;
        gp[i * zdim + c] = (T)0;
      }
    }
  }
}

// fp32 rows of three components, every array 16-byte aligned: a wave takes one tile of 256 rows.  Each (256, 3) array
// moves between HBM and the wave as three fully coalesced 1 KiB accesses (all inputs sent for before any is used) and
// is re-dealt through a wave-private LDS slab so that lane l owns rows 4 l .. 4 l + 3; the results leave the same way.
// The thread-per-row kernel above reads and writes 4-byte elements 12 bytes apart from lane to lane -- seven such
// accesses per row, and the address unit's time for them is that kernel's time.  Tiles are dealt so that an XCD works
// on one contiguous eighth of the rows.  Same per-row functions, same bits.
constexpr int BT_TILE = 256;
__device__ __forceinline__ void bt_deal_in(u32x4* slab, const u32x4 (&u)[3], int lane, float (&a)[12]) {
  u32x4 v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) slab[lane + 64 * k] = u[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = slab[3 * lane + k];
  wave_lds_sync();
  __builtin_memcpy(a, v, 48);
}
__device__ __forceinline__ void bt_deal_out(u32x4* slab, const float (&a)[12], int lane, float* dst) {
  u32x4 v[3];
  __builtin_memcpy(v, a, 48);
#pragma unroll
  for (int k = 0; k < 3; ++k) slab[3 * lane + k] = v[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = slab[lane + 64 * k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f32x4 f;
    __builtin_memcpy(&f, &v[k], 16);
    store_nt(f, dst + 4 * (lane + 64 * k));
  }
}
template <bool BWD>
__global__ __launch_bounds__(256) void build_torus_tile_kernel(const float* __restrict__ zt, const float* __restrict__ zp,
                                                               const float* __restrict__ gout, float* __restrict__ out,
                                                               float* __restrict__ gt, float* __restrict__ gp,
                                                               int64_t ntiles, float R, float r) {
  __shared__ u32x4 s_slab[4][3 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4* const slab = s_slab[wave];
  const int64_t t = (int64_t)xcd_block(blockIdx.x, gridDim.x) * 4 + wave;
  if (t >= ntiles) return;  // (no workgroup barrier in this kernel)
  const int64_t e0 = 3 * BT_TILE * t;
  u32x4 ut[3], up[3], ug[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) ut[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(zt + e0) + lane + 64 * k);
#pragma unroll
  for (int k = 0; k < 3; ++k) up[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(zp + e0) + lane + 64 * k);
  if (BWD) {
#pragma unroll
    for (int k = 0; k < 3; ++k) ug[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gout + e0) + lane + 64 * k);
  }
  float a[12], b[12], g[12], o[12], o2[12];
  bt_deal_in(slab, ut, lane, a);
  bt_deal_in(slab, up, lane, b);
  if (BWD) bt_deal_in(slab, ug, lane, g);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float ct = a[3 * j], st = a[3 * j + 1], cp = b[3 * j], sp = b[3 * j + 1];
    if (!BWD) {
      torus_point<float>(ct, st, cp, sp, R, r, o + 3 * j);
    } else {
      float da[2], db[2];
      torus_point_bwd<float>(ct, cp, sp, R, r, g + 3 * j, da, db);
      o[3 * j] = da[0]; o[3 * j + 1] = da[1]; o[3 * j + 2] = 0.0f;
      o2[3 * j] = db[0]; o2[3 * j + 1] = db[1]; o2[3 * j + 2] = 0.0f;
    }
  }
  if (!BWD) {
    bt_deal_out(slab, o, lane, out + e0);
  } else {
    bt_deal_out(slab, o, lane, gt + e0);
    bt_deal_out(slab, o2, lane, gp + e0);
  }
}
static inline bool bt_tile_ok(int dtype, int zdim, int64_t n, const void* a, const void* b, const void* c, const void* d,
                              const void* e) {
  const auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  return dtype == VMF_F32 && zdim == 3 && n >= BT_TILE && al(a) && al(b) && al(c) && al(d) && al(e) &&
         (n / BT_TILE + 3) / 4 < 0x7fffffff;
}

extern "C" int vmf_build_torus_fwd(const void* z_theta, const void* z_phi, void* out, int64_t n, int zdim, double R,
                                   double r, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_build_torus_fwd");
  VMF_REQUIRE(n >= 0 && zdim >= 2, VMF_ERR_BAD_ARG, "vmf_build_torus_fwd: n %lld, zdim %d", (long long)n, zdim);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z_theta && z_phi && out, VMF_ERR_BAD_ARG, "vmf_build_torus_fwd: null pointer");
  int64_t done = 0;  // rows the tile kernel takes; the thread-per-row kernel takes the rest
  if (bt_tile_ok(dtype, zdim, n, z_theta, z_phi, out, nullptr, nullptr)) {
    const int64_t ntiles = n / BT_TILE;
    hipLaunchKernelGGL((build_torus_tile_kernel<false>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)z_theta, (const float*)z_phi, (const float*)nullptr,
                       (float*)out, (float*)nullptr, (float*)nullptr, ntiles, (float)R, (float)r);
    done = ntiles * BT_TILE;
    if (done == n) return vmf_check_launch("vmf_build_torus_fwd");
  }
  const int es = dtype == VMF_F32 ? 4 : 8;
  const char* zt_ = (const char*)z_theta + done * zdim * es;
  const char* zp_ = (const char*)z_phi + done * zdim * es;
  char* out_ = (char*)out + done * 3 * es;
  int grid = vmf_grid_for(n - done, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((build_torus_kernel<T, false>), dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)zt_, (const T*)zp_,
                                               (const T*)nullptr, (T*)out_, (T*)nullptr, (T*)nullptr, n - done, zdim,
                                               (T)R, (T)r));
  return vmf_check_launch("vmf_build_torus_fwd");
}

extern "C" int vmf_build_torus_bwd(const void* z_theta, const void* z_phi, const void* grad_out, void* g_theta,
                                   void* g_phi, int64_t n, int zdim, double R, double r, int dtype,
                                   vmf_stream_t stream) {
  VMF_RANGE("vmf_build_torus_bwd");
  VMF_REQUIRE(n >= 0 && zdim >= 2, VMF_ERR_BAD_ARG, "vmf_build_torus_bwd: n %lld, zdim %d", (long long)n, zdim);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z_theta && z_phi && grad_out && g_theta && g_phi, VMF_ERR_BAD_ARG, "vmf_build_torus_bwd: null pointer");
  int64_t done = 0;
  if (bt_tile_ok(dtype, zdim, n, z_theta, z_phi, grad_out, g_theta, g_phi)) {
    const int64_t ntiles = n / BT_TILE;
    hipLaunchKernelGGL((build_torus_tile_kernel<true>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)z_theta, (const float*)z_phi, (const float*)grad_out,
                       (float*)nullptr, (float*)g_theta, (float*)g_phi, ntiles, (float)R, (float)r);
    done = ntiles * BT_TILE;
    if (done == n) return vmf_check_launch("vmf_build_torus_bwd");
  }
  const int es = dtype == VMF_F32 ? 4 : 8;
  const char* zt_ = (const char*)z_theta + done * zdim * es;
  const char* zp_ = (const char*)z_phi + done * zdim * es;
  const char* go_ = (const char*)grad_out + done * 3 * es;
  char* gt_ = (char*)g_theta + done * zdim * es;
  char* gp_ = (char*)g_phi + done * zdim * es;
  int grid = vmf_grid_for(n - done, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((build_torus_kernel<T, true>), dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)zt_, (const T*)zp_,
                                               (const T*)go_, (T*)nullptr, (T*)gt_, (T*)gp_, n - done, zdim,
                                               (T)R, (T)r));
  return vmf_check_launch("vmf_build_torus_bwd");
}

// ---- latent losses ------------------------------------------------------------------------------------------
// Row i of z pairs with label row (i mod n_labels): z may carry leading broadcast dimensions ((B, B, 3) against
// labels (B, L)), exactly as `latent_phi - gt_phi` broadcasts in dataloader/utils.py:121.
#define VMF_LL_MAX_ZDIM 8

template <typename T, bool GRAD>
__device__ __forceinline__ T latent_loss_row(const T* zr, int zdim, const T* lab, int variant, T R, T r, T* dz) {
  switch (variant) {
    case VMF_LL_S1: return latent_loss_s1<T, GRAD>(zr, zdim, lab[0], dz);
    case VMF_LL_S2: return latent_loss_s2<T, GRAD>(zr, zdim, lab[0], lab[1], dz);
    default: return latent_loss_t2<T, GRAD>(zr, zdim, lab[0], lab[1], R, r, dz);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void latent_loss_fwd_kernel(const T* __restrict__ z, const T* __restrict__ labels,
                                                              T* __restrict__ per_row, double* __restrict__ partial,
                                                              int64_t n, int zdim, int64_t n_labels, int lcols,
                                                              int variant, T R, T r, T* __restrict__ direct_mean,
                                                              double alpha) {
  __shared__ double red[4];
  double acc = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the label row of z row i is i mod n_labels: one 64-bit division per thread, then a conditional subtraction per step
  const int64_t lstep = stride % n_labels;
  int64_t li = i0 % n_labels;
  for (int64_t i = i0; i < n; i += stride) {
    T zr[VMF_LL_MAX_ZDIM];
    for (int c = 0; c < zdim; ++c) zr[c] = z[i * zdim + c];
    const T* lab = labels + li * lcols;
    T v = latent_loss_row<T, false>(zr, zdim, lab, variant, R, r, nullptr);
    if (per_row) per_row[i] = v;
    acc += (double)v;
    li += lstep;
    if (li >= n_labels) li -= n_labels;
  }
  double t = block_sum_double(acc, red);
  // a one-block launch (batch-sized inputs) is its own finish: same bits as the finish kernel over a single partial
  if (threadIdx.x == 0) {
    if (direct_mean) *direct_mean = (T)(alpha * t);
    else partial[blockIdx.x] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void latent_loss_final_kernel(const double* __restrict__ partial, int nblk,
                                                                double alpha, T* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[i];
  double t = block_sum_double(acc, red);
  if (threadIdx.x == 0) *out = (T)(alpha * t);
}

template <typename T>
__global__ __launch_bounds__(256) void latent_loss_bwd_kernel(const T* __restrict__ z, const T* __restrict__ labels,
                                                              const T* __restrict__ gout, T gscale,
                                                              T* __restrict__ dz_out, int64_t n, int zdim,
                                                              int64_t n_labels, int lcols, int variant, T R, T r) {
  const T g = gout[0] * gscale;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t lstep = stride % n_labels;
  int64_t li = i0 % n_labels;
  for (int64_t i = i0; i < n; i += stride) {
    T zr[VMF_LL_MAX_ZDIM], dz[VMF_LL_MAX_ZDIM];
    for (int c = 0; c < zdim; ++c) zr[c] = z[i * zdim + c];
    const T* lab = labels + li * lcols;
    latent_loss_row<T, true>(zr, zdim, lab, variant, R, r, dz);
    for (int c = 0; c < zdim; ++c) dz_out[i * zdim + c] = g * dz[c];
    li += lstep;
    if (li >= n_labels) li -= n_labels;
  }
}

// fp32 rows of exactly three components, z (and the outputs) 16-byte aligned: a wave takes tiles of 256 rows.  The
// tile's 3 KiB move between HBM and the wave as three fully coalesced 1 KiB accesses and are re-dealt through a
// wave-private LDS slab so that lane l owns rows 4 l .. 4 l + 3 (48 contiguous bytes); the gradient leaves the same way.
// 512-thread blocks: with the grid capped by the partial-sum workspace (1024 blocks) eight waves per SIMD hide the load
// latency.  The thread-per-row kernels below it in the dispatch (any dtype / width / alignment) wait out one load round
// trip per row at four waves per SIMD: 2^24 rows S1 / S2 / T2 forward 0.084 / 0.137 / 0.102 ms there.  Same per-row
// functions, same bits per row.
constexpr int LL_BLOCK = 512;
constexpr int LL_TILE = 256;
// four consecutive floats with the alignment of one (a lane's label rows start wherever row mod n_labels puts them)
struct __attribute__((packed, aligned(4))) LlQuad {
  float v[4];
};
template <bool BWD>
__global__ __launch_bounds__(LL_BLOCK) void latent_loss_tile_kernel(const float* __restrict__ z,
                                                                    const float* __restrict__ labels,
                                                                    float* __restrict__ per_row,
                                                                    double* __restrict__ partial,
                                                                    const float* __restrict__ gout, float gscale,
                                                                    float* __restrict__ dz_out, int64_t n,
                                                                    int64_t n_labels, int lcols, int variant, float R,
                                                                    float r, float* __restrict__ direct_mean,
                                                                    double alpha) {
  __shared__ double red[LL_BLOCK / 64];
  __shared__ u32x4 s_slab[LL_BLOCK / 64][3 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4* const slab = s_slab[wave];
  double acc = 0;
  const float g = BWD ? gout[0] * gscale : 0.0f;
  const int64_t ntiles = n / LL_TILE;
  const int64_t wstride = (int64_t)gridDim.x * (LL_BLOCK / 64);
  const int64_t w0 = (int64_t)blockIdx.x * (LL_BLOCK / 64) + wave;
  // label row of z row i is i mod n_labels: one division per lane, then conditional subtractions
  const int64_t lstep = (wstride * LL_TILE) % n_labels;
  int64_t li = (w0 * LL_TILE + 4 * lane) % n_labels;
  // the NEXT tile's three accesses are sent for as soon as the current tile sits in the slab: a wave walks eight or
  // more tiles, and without this every one of them waited out a full HBM round trip before its arithmetic started
  u32x4 nx[3];
  if (w0 < ntiles) {
    const u32x4* src = reinterpret_cast<const u32x4*>(z + 3 * LL_TILE * w0);
#pragma unroll
    for (int k = 0; k < 3; ++k) nx[k] = __builtin_nontemporal_load(src + lane + 64 * k);
  }
  for (int64_t t = w0; t < ntiles; t += wstride) {
    u32x4 u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) slab[lane + 64 * k] = nx[k];
    wave_lds_sync();
    if (t + wstride < ntiles) {
      const u32x4* src = reinterpret_cast<const u32x4*>(z + 3 * LL_TILE * (t + wstride));
#pragma unroll
      for (int k = 0; k < 3; ++k) nx[k] = __builtin_nontemporal_load(src + lane + 64 * k);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = slab[3 * lane + k];
    wave_lds_sync();
    float a[12], d[12], v[4];
    __builtin_memcpy(a, u, 48);
    // The lane's four label rows are consecutive (row -> row mod n_labels) unless the table wraps inside them: one or
    // two 128-bit accesses fetch them.  (Row by row they were four or eight 4-byte gathers per lane and tile, 32 bytes
    // apart from lane to lane -- the address unit's time for those was the kernel's time: S1, one column, ran at 43 us
    // per 2^24 rows, S2 / T2, two columns, at 70.)
    float lab[4][2];
    if (li + 4 <= n_labels && lcols == 2) {
      const LlQuad q0 = *reinterpret_cast<const LlQuad*>(labels + 2 * li);
      const LlQuad q1 = *reinterpret_cast<const LlQuad*>(labels + 2 * li + 4);
      lab[0][0] = q0.v[0]; lab[0][1] = q0.v[1]; lab[1][0] = q0.v[2]; lab[1][1] = q0.v[3];
      lab[2][0] = q1.v[0]; lab[2][1] = q1.v[1]; lab[3][0] = q1.v[2]; lab[3][1] = q1.v[3];
    } else if (li + 4 <= n_labels && lcols == 1) {
      const LlQuad q0 = *reinterpret_cast<const LlQuad*>(labels + li);
#pragma unroll
      for (int j = 0; j < 4; ++j) { lab[j][0] = q0.v[j]; lab[j][1] = 0.0f; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t l = li + j;
        while (l >= n_labels) l -= n_labels;
        lab[j][0] = labels[l * lcols];
        lab[j][1] = lcols > 1 ? labels[l * lcols + 1] : 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = latent_loss_row<float, BWD>(a + 3 * j, 3, lab[j], variant, R, r, BWD ? d + 3 * j : nullptr);
    if (BWD) {
#pragma unroll
      for (int c = 0; c < 12; ++c) d[c] *= g;
      __builtin_memcpy(u, d, 48);
#pragma unroll
      for (int k = 0; k < 3; ++k) slab[3 * lane + k] = u[k];
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 3; ++k) u[k] = slab[lane + 64 * k];
      wave_lds_sync();
      float* dst = dz_out + 3 * LL_TILE * t;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        f32x4 f;
        __builtin_memcpy(&f, &u[k], 16);
        store_nt(f, dst + 4 * (lane + 64 * k));
      }
    } else {
      if (per_row) store4_stream(per_row + LL_TILE * t + 4 * lane, v);
      acc += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    }
    li += lstep;
    if (li >= n_labels) li -= n_labels;
  }
  // the last n mod 256 rows, one per thread
  for (int64_t i = LL_TILE * ntiles + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float zr[3] = {z[3 * i], z[3 * i + 1], z[3 * i + 2]};
    float d[3];
    const float v = latent_loss_row<float, BWD>(zr, 3, labels + (i % n_labels) * lcols, variant, R, r, BWD ? d : nullptr);
    if (BWD) {
      for (int c = 0; c < 3; ++c) dz_out[3 * i + c] = g * d[c];
    } else {
      if (per_row) per_row[i] = v;
      acc += (double)v;
    }
  }
  if (!BWD) {
    double t = block_sum_double(acc, red);
    if (threadIdx.x == 0) {
      if (direct_mean) *direct_mean = (float)(alpha * t);
      else partial[blockIdx.x] = t;
    }
  }
}

// blocks of the tile kernel for n rows: one wave per tile up to the cap, at least one block for the remainder rows
static inline int ll_tile_grid(int64_t n, int max_blocks) {
  return vmf_grid_for((n + LL_TILE - 1) / LL_TILE, LL_BLOCK / 64, max_blocks);
}

static inline bool ll_tile_ok(int dtype, int zdim, const void* z, const void* a, const void* b) {
  return dtype == VMF_F32 && zdim == 3 && aligned16(z) && aligned16(a) && aligned16(b);
}

static int latent_loss_check(const char* who, int64_t n, int zdim, int64_t n_labels, int lcols, int variant) {
  VMF_REQUIRE(variant >= VMF_LL_S1 && variant <= VMF_LL_T2, VMF_ERR_BAD_ARG, "%s: unknown variant %d", who, variant);
  VMF_REQUIRE(n >= 0 && n_labels > 0, VMF_ERR_BAD_ARG, "%s: bad row counts", who);
  VMF_REQUIRE(n % n_labels == 0, VMF_ERR_BAD_ARG, "%s: %lld rows of z do not tile %lld label rows", who,
              (long long)n, (long long)n_labels);
  const int need_z = variant == VMF_LL_S1 ? 2 : 3, need_l = variant == VMF_LL_S1 ? 1 : 2;
  VMF_REQUIRE(zdim >= need_z && zdim <= VMF_LL_MAX_ZDIM, VMF_ERR_BAD_ARG, "%s: zdim %d outside [%d, %d]", who, zdim,
              need_z, VMF_LL_MAX_ZDIM);
  VMF_REQUIRE(lcols >= need_l, VMF_ERR_BAD_ARG, "%s: labels need %d columns, got %d", who, need_l, lcols);
  return VMF_OK;
}

extern "C" int vmf_latent_loss_fwd(const void* z, const void* labels, void* per_row_out, void* mean_out,
                                   double* workspace, int64_t n, int zdim, int64_t n_labels, int label_cols,
                                   int variant, double R, double r, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_latent_loss_fwd");
  int rc = latent_loss_check("vmf_latent_loss_fwd", n, zdim, n_labels, label_cols, variant);
  if (rc) return rc;
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE(mean_out && workspace && labels && (n == 0 || z), VMF_ERR_BAD_ARG, "vmf_latent_loss_fwd: null pointer");
  int grid = vmf_grid_for(n, 256, VMF_KL_WORKSPACE_DOUBLES);
  if (n > 0 && ll_tile_ok(dtype, zdim, z, per_row_out, nullptr)) {
    grid = ll_tile_grid(n, VMF_KL_WORKSPACE_DOUBLES);
    const bool direct = grid == 1;
    hipLaunchKernelGGL(latent_loss_tile_kernel<false>, dim3(grid), dim3(LL_BLOCK), 0, (hipStream_t)stream,
                       (const float*)z, (const float*)labels, (float*)per_row_out, workspace, (const float*)nullptr,
                       0.0f, (float*)nullptr, n, n_labels, label_cols, variant, (float)R, (float)r,
                       direct ? (float*)mean_out : (float*)nullptr, 1.0 / (double)n);
    rc = vmf_check_launch("vmf_latent_loss_fwd");
    if (rc || direct) return rc;
  } else if (n > 0) {
    const bool direct = grid == 1;
    VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(latent_loss_fwd_kernel<T>, dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)z, (const T*)labels,
                                                 (T*)per_row_out, workspace, n, zdim, n_labels, label_cols, variant,
                                                 (T)R, (T)r, direct ? (T*)mean_out : (T*)nullptr, 1.0 / (double)n));
    rc = vmf_check_launch("vmf_latent_loss_fwd");
    if (rc || direct) return rc;
  }
  // mean over zero rows is NaN, as torch's .mean() of an empty tensor is
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(latent_loss_final_kernel<T>, dim3(1), dim3(256), 0,
                                               (hipStream_t)stream, workspace, n > 0 ? grid : 0,
                                               n > 0 ? 1.0 / (double)n : (double)NAN, (T*)mean_out));
  return vmf_check_launch("vmf_latent_loss_fwd final");
}

extern "C" int vmf_latent_loss_bwd(const void* z, const void* labels, const void* grad_mean, void* dz, int64_t n,
                                   int zdim, int64_t n_labels, int label_cols, int variant, double R, double r,
                                   int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_latent_loss_bwd");
  int rc = latent_loss_check("vmf_latent_loss_bwd", n, zdim, n_labels, label_cols, variant);
  if (rc) return rc;
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z && labels && grad_mean && dz, VMF_ERR_BAD_ARG, "vmf_latent_loss_bwd: null pointer");
  if (ll_tile_ok(dtype, zdim, z, dz, nullptr)) {
    hipLaunchKernelGGL(latent_loss_tile_kernel<true>, dim3(ll_tile_grid(n, 256 * 8)), dim3(LL_BLOCK), 0,
                       (hipStream_t)stream, (const float*)z, (const float*)labels, (float*)nullptr, (double*)nullptr,
                       (const float*)grad_mean, (float)(1.0 / (double)n), (float*)dz, n, n_labels, label_cols, variant,
                       (float)R, (float)r, (float*)nullptr, 0.0);
    return vmf_check_launch("vmf_latent_loss_bwd");
  }
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(latent_loss_bwd_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)z, (const T*)labels,
                                               (const T*)grad_mean, (T)(1.0 / (double)n), (T*)dz, n, zdim, n_labels,
                                               label_cols, variant, (T)R, (T)r));
  return vmf_check_launch("vmf_latent_loss_bwd");
}
