// elementwise.hip -- ive / log_ive / Bessel ratio, the KL variants with their fused sum, the uniform prior
// on the sphere and the S^1 projected-normal sampler.  All are streams over kappa / rows: read once, write
// once, grid capped at 2048 workgroups with a grid-stride loop.
#include <atomic>
#include <mutex>

#include "vmf_math.h"
#include "kl_m2_table.h"
#include "kl_m3_table.h"

// ---- ive ---------------------------------------------------------------------------------------------
// The reference moves z to the host, calls scipy (AMOS / cephes, double internally for either dtype) and
// copies the result back (sampler/ive.py:18-28).  Here each lane evaluates in double and rounds once to T.
// The order's coefficient tables (IveTab) travel as kernel arguments: the kernels read them through scalar loads.
// KIND / KIND1: the evaluator of the order (and of the order minus one), see ive_kind_of.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void ive_fwd_kernel(const T* __restrict__ z, T* __restrict__ out, int64_t n,
                                                      BesselOrder o, IveTab tab) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (T)ive_eval_kind<KIND>(o, tab, (double)z[i]);
}

template <typename T, int KIND, int KIND1>
__global__ __launch_bounds__(256) void ive_bwd_kernel(const T* __restrict__ z, const T* __restrict__ gout,
                                                      T* __restrict__ gz, int64_t n, BesselOrder o,
                                                      BesselOrder om1, IveTab tab, IveTab tabm1, T v) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T zi = z[i];
    T a = (T)ive_eval_kind<KIND>(o, tab, (double)zi), b = (T)ive_eval_kind<KIND1>(om1, tabm1, (double)zi);
    T zs = M<T>::abs(zi) < (T)1e-8 ? (T)1e-8 : zi;                       // sampler/ive.py:55-56
    T sg = (T)((zi > (T)0) - (zi < (T)0));                               // torch.sign
    gz[i] = gout[i] * (b - (v / zs + sg) * a);                           // :60-62
  }
}

template <typename T>
__global__ __launch_bounds__(256) void log_ive_kernel(const T* __restrict__ z, T* __restrict__ out, int64_t n,
                                                      double v, double lg_v1, IveTab tab) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double x = (double)z[i];
    out[i] = (T)(x > 0 ? log_ive_pos_tab(tab, v, lg_v1, x) : (x == 0 ? (v == 0 ? 0.0 : -INFINITY) : NAN));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bessel_ratio_kernel(const T* __restrict__ z, T* __restrict__ out,
                                                           int64_t n, double v, double lg_v1, double lg_v,
                                                           IveTab tab1, IveTab tab0) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double x = (double)z[i];
    out[i] = (T)(x >= 0 ? bessel_ratio_tab(tab1, tab0, v, lg_v1, lg_v, x) : NAN);
  }
}

extern "C" int vmf_ive_fwd(double v, const void* z, void* out, int64_t n, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_ive_fwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z && out, VMF_ERR_BAD_ARG, "vmf_ive_fwd: null pointer");
  // forward call site: i0e / i1e for v == 0 / 1, scipy.special.ive otherwise (sampler/ive.py:21-26)
  BesselOrder o = make_bessel_order(v, !(v == 0.0 || v == 1.0));
  const IveTab tab = make_ive_tab(o);
  int grid = vmf_grid_for(n, 256);
#define VMF_IVE_FWD(K)                                                                                          \
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ive_fwd_kernel<T, K>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                               (const T*)z, (T*)out, n, o, tab))
  switch (ive_kind_of(o)) {
    case IVE_KIND_CHEB: VMF_IVE_FWD(IVE_KIND_CHEB); break;
    case IVE_KIND_HALF: VMF_IVE_FWD(IVE_KIND_HALF); break;
    case IVE_KIND_GENERIC: VMF_IVE_FWD(IVE_KIND_GENERIC); break;
    default: VMF_IVE_FWD(IVE_KIND_SLOW); break;
  }
#undef VMF_IVE_FWD
  return vmf_check_launch("vmf_ive_fwd");
}

extern "C" int vmf_ive_bwd(double v, const void* z, const void* gout, void* gz, int64_t n, int dtype,
                           vmf_stream_t stream) {
  VMF_RANGE("vmf_ive_bwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z && gout && gz, VMF_ERR_BAD_ARG, "vmf_ive_bwd: null pointer");
  // backward call sites (sampler/ive.py:40-48): v == 0 -> (i0e, ive(-1)); v == 1 -> (i1e, i0e); else ive, ive
  BesselOrder o = make_bessel_order(v, !(v == 0.0 || v == 1.0));
  BesselOrder om1 = make_bessel_order(v - 1.0, v != 1.0);
  const IveTab tab = make_ive_tab(o), tabm1 = make_ive_tab(om1);
  int grid = vmf_grid_for(n, 256);
#define VMF_IVE_BWD(K, K1)                                                                                     \
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ive_bwd_kernel<T, K, K1>), dim3(grid), dim3(256), 0,            \
                                               (hipStream_t)stream, (const T*)z, (const T*)gout, (T*)gz, n, o, om1, \
                                               tab, tabm1, (T)v))
  // the pairs (order, order - 1) that occur: integers (2 -> generic + Chebyshev, beyond: generic twice; 0 and 1:
  // Chebyshev twice), the half-integers with closed forms, other positive orders above one; anything else (an order
  // minus one in (-1, 0)) takes the general evaluator for both
  const int k = ive_kind_of(o), k1 = ive_kind_of(om1);
  if (k == IVE_KIND_CHEB && k1 == IVE_KIND_CHEB) VMF_IVE_BWD(IVE_KIND_CHEB, IVE_KIND_CHEB);
  else if (k == IVE_KIND_HALF && k1 == IVE_KIND_HALF) VMF_IVE_BWD(IVE_KIND_HALF, IVE_KIND_HALF);
  else if (k == IVE_KIND_GENERIC && k1 == IVE_KIND_CHEB) VMF_IVE_BWD(IVE_KIND_GENERIC, IVE_KIND_CHEB);
  else if (k == IVE_KIND_GENERIC && k1 == IVE_KIND_GENERIC) VMF_IVE_BWD(IVE_KIND_GENERIC, IVE_KIND_GENERIC);
  else if (k == IVE_KIND_GENERIC && k1 == IVE_KIND_HALF) VMF_IVE_BWD(IVE_KIND_GENERIC, IVE_KIND_HALF);
  else VMF_IVE_BWD(IVE_KIND_SLOW, IVE_KIND_SLOW);
#undef VMF_IVE_BWD
  return vmf_check_launch("vmf_ive_bwd");
}

extern "C" int vmf_log_ive_fwd(double v, const void* z, void* out, int64_t n, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_log_ive_fwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  VMF_REQUIRE(v >= 0, VMF_ERR_UNSUPPORTED, "vmf_log_ive_fwd: order %g < 0", v);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z && out, VMF_ERR_BAD_ARG, "vmf_log_ive_fwd: null pointer");
  int grid = vmf_grid_for(n, 256);
  double lg = lgamma(v + 1.0);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(log_ive_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)z, (T*)out, n, v, lg, make_ive_tab(make_bessel_order(v, 0))));
  return vmf_check_launch("vmf_log_ive_fwd");
}

extern "C" int vmf_bessel_ratio_fwd(double v, const void* z, void* out, int64_t n, int dtype,
                                    vmf_stream_t stream) {
  VMF_RANGE("vmf_bessel_ratio_fwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  VMF_REQUIRE(v >= 1.0, VMF_ERR_UNSUPPORTED, "vmf_bessel_ratio_fwd: order %g < 1", v);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(z && out, VMF_ERR_BAD_ARG, "vmf_bessel_ratio_fwd: null pointer");
  int grid = vmf_grid_for(n, 256);
  double lg1 = lgamma(v + 1.0), lg0 = lgamma(v);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bessel_ratio_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)z, (T*)out, n, v, lg1, lg0,
                                               make_ive_tab(make_bessel_order(v, 0)),
                                               make_ive_tab(make_bessel_order(v - 1.0, 0))));
  return vmf_check_launch("vmf_bessel_ratio_fwd");
}

// ---- KL ------------------------------------------------------------------------------------------------
// Replaces the boolean-mask gather/scatter and the two `.any()` host syncs of
// model/spherical_vae.py:153-182 with one pass; the `.mean()` of :134 is the fused two-stage sum.
// VEC = 4: 128-bit loads / stores of four consecutive elements per lane (fp32, 16-byte aligned); the n % 4 tail is
// taken by the first lanes with scalar accesses.
// Each lane keeps a few 128-bit loads in flight (the element loop is a dependent chain otherwise: one 1 KiB request per
// wave at a time leaves HBM idle).  The <= 1024 block partials of the workspace cap the grid at four blocks per CU, so
// the block size sets the occupancy: 512 threads = 8 waves per SIMD (every instantiation stays under 64 VGPRs), which
// hides the quarter-rate transcendentals of the m = 3 forms far better than 256 threads x 4 loads did (2^27 elements:
// torus 0.169 -> 0.118 ms, circular 0.167 -> 0.138, stable m=3 0.154 -> 0.126, spherical 0.188 -> 0.170; the
// instruction-bound m = 2 fits 0.178 -> 0.165 / 0.170 -> 0.160 with three loads in flight, the others are best with two).
#ifndef VMF_KL_BLOCK
#define VMF_KL_BLOCK 512
#endif
constexpr int KL_BLOCK = VMF_KL_BLOCK;
#ifdef VMF_KL_UNROLL
constexpr int kl_unroll(int) { return VMF_KL_UNROLL; }
#else
constexpr int kl_unroll(int variant) { return (variant == VMF_KL_S1_I0I1 || variant == VMF_KL_STABLE_M2) ? 3 : 2; }
#endif
// VARIANT is a template argument: with the variant a run-time switch, every one of the 16 inlined per-element evaluations
// of a trip carried all six forms (25 000 instructions in one kernel), and the kernels ran out of the instruction cache
// rather than out of HBM.
// every element enters the sum as a double (the fused sum equals the double-precision sum of the per-element output);
// adding four neighbours in single precision first measured the same time
template <typename T>
__device__ __forceinline__ double sum4_to_double(const T (&v)[4]) {
  return ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
}

// Four consecutive elements.  The m = 3 variants in single precision have one straight-line form on the range an
// encoder produces (kappa = softplus + 1 >= 1, below the point where the variant's own overflow behaviour starts): one
// range check per four elements instead of the per-element region branches (spherical 27 -> 19 instructions per element);
// a quad with any element outside the range -- or NaN -- takes the per-element form.  Both call the same functions
// (the two inlined copies may differ in the last bit: the compiler contracts them separately).
// `tab`: the block's LDS copy of the m = 2 table (fp32 instantiations of the two S^1 variants), null otherwise.
template <typename T, bool BESSEL, int VARIANT>
__device__ __forceinline__ void kl_eval_value4(const T (&k)[4], T (&o)[4], const KlBessel& kb, const f32x4* tab) {
#pragma unroll
  for (int x = 0; x < 4; ++x) o[x] = kl_eval_value<T, BESSEL>(k[x], VARIANT, kb);
}

// ---- the two S^1 variants in single precision: a piecewise cubic read from LDS -----------------------------------------
// KL(vM(kappa) || U(S^1)) = kappa I1/I0 - log I0 through kl_m2_table.h: the float's exponent and top mantissa bits pick
// the piece, the remaining mantissa bits are the position inside it; one 128-bit LDS read and three fused multiply-adds
// per element, no transcendental, the same instructions for every concentration (the two-polynomial form of
// kl_m2_value_f32 costs a wave of a mixed batch both chains, 12 + 17 terms, a logarithm and a reciprocal: 40 vector
// instructions per element against 16).  VMF_KL_S1_I0I1 keeps the reference's overflow behaviour: NaN from 88.75 on,
// the value-and-derivative form, which builds e^kappa and decides the overflow itself, in the window (88.70, 88.75).
constexpr bool kl_uses_m2_table(int variant, int elem_size) {
  return elem_size == 4 && (variant == VMF_KL_S1_I0I1 || variant == VMF_KL_STABLE_M2);
}
// the m = 3 variants read their table in the 128-bit instantiation only (its four-element fast path)
constexpr bool kl_uses_m3_table(int variant, int elem_size, int vec) { return elem_size == 4 && vec == 4 && !kl_uses_m2_table(variant, 4); }
constexpr int kl_table_size(int variant, int elem_size, int vec) {
  return kl_uses_m2_table(variant, elem_size) ? KLM2_T_N : (kl_uses_m3_table(variant, elem_size, vec) ? KLM3_T_N : 1);
}
template <int VARIANT>
__device__ __forceinline__ void kl_stage_table(f32x4* tab) {
  const float(*src)[4] = (VARIANT == VMF_KL_S1_I0I1 || VARIANT == VMF_KL_STABLE_M2) ? KLM2_T
                         : (VARIANT == VMF_KL_SPHERICAL_PIECEWISE ? KLM3_T_EPS1E8
                            : (VARIANT == VMF_KL_CIRCULAR_EPS ? KLM3_T_EPS1E6 : KLM3_T_EPS0));
  const int n = (VARIANT == VMF_KL_S1_I0I1 || VARIANT == VMF_KL_STABLE_M2) ? KLM2_T_N : KLM3_T_N;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tab[i] = *reinterpret_cast<const f32x4*>(src[i]);
  __syncthreads();
}
// false: the element is outside the table (or in the S^1 form's overflow window): the caller takes kl_eval_value
template <int VARIANT>
__device__ __forceinline__ bool kl_m2_table_value(const f32x4* tab, float kappa, float* out) {
  const float a = ::fabsf(kappa);
  const uint32_t bits = __float_as_uint(a);
  const uint32_t idx = (bits >> (23 - KLM2_T_SUB_BITS)) - ((127u + KLM2_T_E0) << KLM2_T_SUB_BITS);
  bool ok = idx < (uint32_t)KLM2_T_N;  // also false for NaN / inf (their exponent is past the table)
  if (VARIANT == VMF_KL_S1_I0I1) ok = ok && (a <= 88.70f || a >= 88.75f);
  const f32x4 c = tab[ok ? idx : 0u];
  const uint32_t tb = ((bits & ((1u << (23 - KLM2_T_SUB_BITS)) - 1u)) << KLM2_T_SUB_BITS) | 0x3F800000u;
  const float t = __uint_as_float(tb) - 1.0f;
  const float v = ::fmaf(::fmaf(::fmaf(c.w, t, c.z), t, c.y), t, c.x);
  *out = (VARIANT == VMF_KL_S1_I0I1 && a >= 88.75f) ? NAN : v;  // e^a = inf: inf / inf in the reference
  return ok;
}
// Elements the tables do not cover (concentrations below 1, the windows around the reference's overflow points, NaN,
// values past the tables' end).  ONE copy of this code per kernel, not one per unrolled position: it keeps the
// streaming loop small, and in the S^1 form's overflow window the value is a difference of two numbers near 88 (a few
// 1e-6 of rounding) -- copies that the compiler contracts differently would make an element's value depend on where in
// a block's trip it sits.
template <int VARIANT>
__device__ __attribute__((noinline)) float kl_uncovered_f32(float k) {
  const KlBessel none = {};  // the single-precision forms never touch the double-precision Bessel orders
  return kl_eval_value<float, false>(k, VARIANT, none);
}
template <bool BESSEL, int VARIANT>
__device__ __forceinline__ float kl_m2_value1(const f32x4* tab, float k, const KlBessel& kb) {
  float v;
  if (kl_m2_table_value<VARIANT>(tab, k, &v)) return v;
  if (!BESSEL) return kl_uncovered_f32<VARIANT>(k);
  return kl_eval_value<float, BESSEL>(k, VARIANT, kb);
}
#define VMF_KL_QUAD_M2(BESSEL, V)                                                                              \
  template <>                                                                                                  \
  __device__ __forceinline__ void kl_eval_value4<float, BESSEL, V>(const float (&k)[4], float (&o)[4],         \
                                                                   const KlBessel& kb, const f32x4* tab) {     \
    bool ok = true;                                                                                            \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) ok = kl_m2_table_value<V>(tab, k[x], &o[x]) && ok;            \
    if (ok) return;                                                                                            \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) o[x] = kl_m2_value1<BESSEL, V>(tab, k[x], kb);                \
  }
VMF_KL_QUAD_M2(false, VMF_KL_S1_I0I1)
VMF_KL_QUAD_M2(false, VMF_KL_STABLE_M2)
#undef VMF_KL_QUAD_M2
// one table element: c0 + t (c1 + t (c2 + t c3)) of the piece that k's exponent and top mantissa bits select; k must
// lie inside the table (1 <= k < 2^28: the callers' range checks see to it)
__device__ __forceinline__ float kl_m3_table_lookup(const f32x4* tab, float k) {
  const uint32_t bits = __float_as_uint(k);
  const uint32_t idx = (bits >> (23 - KLM3_T_SUB_BITS)) - ((127u + KLM3_T_E0) << KLM3_T_SUB_BITS);
  const f32x4 c = tab[idx < (uint32_t)KLM3_T_N ? idx : 0u];
  const uint32_t tb = ((bits & ((1u << (23 - KLM3_T_SUB_BITS)) - 1u)) << KLM3_T_SUB_BITS) | 0x3F800000u;
  const float t = __uint_as_float(tb) - 1.0f;
  return ::fmaf(::fmaf(::fmaf(c.w, t, c.z), t, c.y), t, c.x);
}
template <int VARIANT>
__device__ __forceinline__ bool kl_quad_m3(const float (&k)[4], float (&o)[4], const f32x4* tab) {
  // (the table ends at 2^28: beyond, the overflow-free form's log k + ln 2 - 1 is left to the per-element path)
  constexpr float hi_limit = VARIANT == VMF_KL_SPHERICAL_PIECEWISE ? 1000.0f
                             : (VARIANT == VMF_KL_STABLE_M3 ? 2.6e8f : KL_M3_UNIFIED_MAX);
  const float lo = fminf(fminf(k[0], k[1]), fminf(k[2], k[3])), hi = fmaxf(fmaxf(k[0], k[1]), fmaxf(k[2], k[3]));
  // fminf / fmaxf drop a NaN operand: the sum test catches it
  const bool in_range = lo >= 1.0f && hi <= hi_limit && (k[0] + k[1] + k[2] + k[3]) == (k[0] + k[1] + k[2] + k[3]);
  constexpr bool OVERFLOWS = VARIANT == VMF_KL_TORUS_RAW || VARIANT == VMF_KL_CIRCULAR_EPS;
  // (the vote keeps a wave on ONE of the two straight-line paths: lanes split between them would run both in turn)
  if (OVERFLOWS ? !__all(in_range) : !in_range) {
    // torus / circular forms: concentrations past sinh's overflow point are NaN in the reference; a quad that holds
    // some (a batch with kappa up to 1e3 has them in every wave) still takes the straight-line form -- with the
    // per-element form instead, such a batch ran at 0.26 ms per 2^27 against 0.12
    if (!OVERFLOWS) return false;
    bool ok = true;
#pragma unroll
    for (int x = 0; x < 4; ++x) ok = ok && k[x] >= 1.0f && (k[x] <= KL_M3_UNIFIED_MAX || k[x] >= KL_M3_NAN_MIN);
    if (!ok) return false;  // (a NaN fails k >= 1)
#pragma unroll
    for (int x = 0; x < 4; ++x) o[x] = k[x] >= KL_M3_NAN_MIN ? NAN : kl_m3_table_lookup(tab, k[x]);
    return true;
  }
#pragma unroll
  for (int x = 0; x < 4; ++x) o[x] = kl_m3_table_lookup(tab, k[x]);
  return true;
}
#define VMF_KL_QUAD_M3(BESSEL, V)                                                                              \
  template <>                                                                                                  \
  __device__ __forceinline__ void kl_eval_value4<float, BESSEL, V>(const float (&k)[4], float (&o)[4],         \
                                                                   const KlBessel& kb, const f32x4* tab) {     \
    if (kl_quad_m3<V>(k, o, tab)) return;                                                                           \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) o[x] = kl_uncovered_f32<V>(k[x]);                            \
  }
VMF_KL_QUAD_M3(false, VMF_KL_SPHERICAL_PIECEWISE)
VMF_KL_QUAD_M3(false, VMF_KL_TORUS_RAW)
VMF_KL_QUAD_M3(false, VMF_KL_CIRCULAR_EPS)
VMF_KL_QUAD_M3(false, VMF_KL_STABLE_M3)
#undef VMF_KL_QUAD_M3

template <typename T, int VEC, bool BESSEL, int VARIANT>
__global__ __launch_bounds__(KL_BLOCK) void kl_fwd_kernel(const T* __restrict__ kappa, T* __restrict__ out,
                                                          double* __restrict__ partial, int64_t n, KlBessel kb,
                                                          double* __restrict__ direct_sum, double alpha) {
  constexpr int variant = VARIANT;
  constexpr int KL_UNROLL = kl_unroll(VARIANT);
  __shared__ double red[KL_BLOCK / 64];
  constexpr bool TABLE = kl_uses_m2_table(VARIANT, sizeof(T));              // every fp32 element goes through it
  constexpr bool TABLE3 = kl_uses_m3_table(VARIANT, sizeof(T), VEC);        // the four-element fast path does
  __shared__ f32x4 s_tab[kl_table_size(VARIANT, sizeof(T), VEC)];
  if (TABLE || TABLE3) kl_stage_table<VARIANT>(s_tab);
  const f32x4* const tab = (TABLE || TABLE3) ? s_tab : nullptr;
  double acc = 0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = VEC == 4 ? n / 4 : 0;
  // a block takes KL_UNROLL consecutive 4 KiB pieces per trip, so that the loads a wave has in flight are neighbours
  // (loads one grid stride -- a power of two -- apart land on the same HBM channels: 8 such streams ran at half the rate)
  const int64_t chunk = (int64_t)blockDim.x * KL_UNROLL;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < nvec; base += (int64_t)gridDim.x * chunk) {
    Vec4<T> k4[KL_UNROLL];
#pragma unroll
    for (int u = 0; u < KL_UNROLL; ++u) {
      const int64_t i = base + u * blockDim.x + threadIdx.x;
      if (i < nvec) {
        if constexpr (sizeof(T) == 4) {  // read once: past the caches
          const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kappa + 4 * i));
          k4[u].v[0] = v.x; k4[u].v[1] = v.y; k4[u].v[2] = v.z; k4[u].v[3] = v.w;
        } else {
          k4[u] = *reinterpret_cast<const Vec4<T>*>(kappa + 4 * i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KL_UNROLL; ++u) {
      const int64_t i = base + u * blockDim.x + threadIdx.x;
      if (i < nvec) {
        Vec4<T> o4;
        kl_eval_value4<T, BESSEL, VARIANT>(k4[u].v, o4.v, kb, tab);
        if (out) *reinterpret_cast<Vec4<T>*>(out + 4 * i) = o4;
        acc += sum4_to_double(o4.v);
      }
    }
  }
  for (int64_t i = 4 * nvec + tid; i < n; i += stride) {
    T kl;
    if constexpr (TABLE) kl = kl_m2_value1<BESSEL, VARIANT>(tab, kappa[i], kb);
    else kl = kl_eval_value<T, BESSEL>(kappa[i], variant, kb);
    if (out) out[i] = kl;
    acc += (double)kl;
  }
  if (partial) {
    double t = block_sum_double(acc, red);
    // a one-block launch (n up to a few thousand: the batch sizes of the models) is its own finish
    if (threadIdx.x == 0) {
      if (direct_sum) *direct_sum = alpha * t;
      else partial[blockIdx.x] = t;
    }
  }
}

__global__ __launch_bounds__(256) void kl_final_kernel(const double* __restrict__ partial, int nblk, double alpha,
                                                       double* __restrict__ sum_out) {
  __shared__ double red[4];
  double acc = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[i];
  double t = block_sum_double(acc, red);
  if (threadIdx.x == 0) *sum_out = alpha * t;
}

template <typename T, int VEC, bool BESSEL, int VARIANT>
__global__ __launch_bounds__(256) void kl_bwd_kernel(const T* __restrict__ kappa, const T* __restrict__ gout,
                                                     int64_t gstride, T gscale, T* __restrict__ dkappa, int64_t n,
                                                     KlBessel kb) {
  constexpr int variant = VARIANT;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = VEC == 4 ? n / 4 : 0;
  for (int64_t i = tid; i < nvec; i += stride) {
    const Vec4<T> k4 = *reinterpret_cast<const Vec4<T>*>(kappa + 4 * i);
    Vec4<T> go;
    if (gstride) go = *reinterpret_cast<const Vec4<T>*>(gout + 4 * i);
    else go.v[0] = go.v[1] = go.v[2] = go.v[3] = gout[0];
    Vec4<T> d4;
    T g;
    kl_eval<T, BESSEL>(k4.v[0], variant, kb, &g); d4.v[0] = go.v[0] * gscale * g;
    kl_eval<T, BESSEL>(k4.v[1], variant, kb, &g); d4.v[1] = go.v[1] * gscale * g;
    kl_eval<T, BESSEL>(k4.v[2], variant, kb, &g); d4.v[2] = go.v[2] * gscale * g;
    kl_eval<T, BESSEL>(k4.v[3], variant, kb, &g); d4.v[3] = go.v[3] * gscale * g;
    *reinterpret_cast<Vec4<T>*>(dkappa + 4 * i) = d4;
  }
  for (int64_t i = 4 * nvec + tid; i < n; i += stride) {
    T g;
    kl_eval<T, BESSEL>(kappa[i], variant, kb, &g);
    dkappa[i] = gout[i * gstride] * gscale * g;
  }
}

static inline bool ew_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// shared by vmf_kl_fwd (caller's workspace) and vmf_kl_fwd_static (the library's own scratch array)
static int kl_fwd_launch(const char* who, const void* kappa, void* out, double* sum_out, double* workspace,
                         double alpha, int64_t n, int variant, int dtype, hipStream_t stream) {
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "%s: n < 0", who);
  VMF_REQUIRE(variant >= 0 && variant <= VMF_KL_STABLE_M2, VMF_ERR_BAD_ARG, "unknown KL variant %d", variant);
  VMF_REQUIRE(!sum_out || workspace, VMF_ERR_BAD_ARG, "%s: sum_out needs a workspace", who);
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d", dtype);
  VMF_REQUIRE(n == 0 || kappa, VMF_ERR_BAD_ARG, "%s: null kappa", who);
  KlBessel kb = make_kl_bessel();
  int grid = vmf_grid_for(dtype == VMF_F32 ? (n + 3) / 4 : n, KL_BLOCK, VMF_KL_WORKSPACE_DOUBLES);
  // what one block covers in a single trip of its loop stays one block (batch-sized inputs: no second launch, below)
  if (dtype == VMF_F32 && sum_out && (n + 3) / 4 <= (int64_t)KL_BLOCK * kl_unroll(variant)) grid = 1;
  // one block: the kernel writes alpha * sum itself (same bits as the finish over a single partial) -- one launch
  double* const direct = (sum_out && n > 0 && grid == 1) ? sum_out : nullptr;
  if (n > 0) {
#define VMF_KL_FWD(V)                                                                                                  \
  case V:                                                                                                              \
    if (dtype == VMF_F32 && ew_aligned16(kappa) && ew_aligned16(out))                                                  \
      hipLaunchKernelGGL((kl_fwd_kernel<float, 4, false, V>), dim3(grid), dim3(KL_BLOCK), 0, stream,                   \
                         (const float*)kappa, (float*)out, sum_out ? workspace : nullptr, n, kb, direct, alpha);       \
    else if (dtype == VMF_F32)                                                                                         \
      hipLaunchKernelGGL((kl_fwd_kernel<float, 1, true, V>), dim3(grid), dim3(KL_BLOCK), 0, stream,                    \
                         (const float*)kappa, (float*)out, sum_out ? workspace : nullptr, n, kb, direct, alpha);       \
    else                                                                                                               \
      hipLaunchKernelGGL((kl_fwd_kernel<double, 1, true, V>), dim3(grid), dim3(KL_BLOCK), 0, stream,                   \
                         (const double*)kappa, (double*)out, sum_out ? workspace : nullptr, n, kb, direct, alpha);     \
    break
    // fp32 with 128-bit accesses: every variant has a single-precision form, so that instantiation leaves the
    // double-precision Bessel code out (46 instead of 149 VGPRs)
    switch (variant) {
      VMF_KL_FWD(0); VMF_KL_FWD(1); VMF_KL_FWD(2); VMF_KL_FWD(3); VMF_KL_FWD(4); VMF_KL_FWD(5);
      default: break;
    }
#undef VMF_KL_FWD
    int rc = vmf_check_launch(who);
    if (rc) return rc;
  }
  if (sum_out && !direct) {
    // every block of the first launch has written its partial; the finish reads exactly those `grid` words, so the
    // workspace needs no clearing whatever an earlier call left in it
    hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(256), 0, stream, workspace, n > 0 ? grid : 0, alpha, sum_out);
    return vmf_check_launch(who);
  }
  return VMF_OK;
}

extern "C" int vmf_kl_fwd(const void* kappa, void* out, double* sum_out, double* workspace, double alpha,
                          int64_t n, int variant, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_kl_fwd");
  return kl_fwd_launch("vmf_kl_fwd", kappa, out, sum_out, workspace, alpha, n, variant, dtype, (hipStream_t)stream);
}

// ---- vmf_kl_fwd without a workspace argument: the block partials live in an array of the code object ----------------
// For callers that cannot allocate device memory (a stream under HIP graph capture, a serving loop, a harness that
// forbids hipMalloc).  Each GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; the array's name is only a host-side handle, so the device address is looked up once per device
// (hipGetSymbolAddress, with that device current) and that pointer is what the kernels receive.
static __device__ __align__(256) double vmf_kl_static_scratch[VMF_KL_WORKSPACE_DOUBLES];

constexpr int VMF_MAX_DEVICES = 64;
static std::atomic<double*> g_kl_scratch_ptr[VMF_MAX_DEVICES];
static std::mutex g_kl_scratch_lock;

// device address of the scratch array on the current device (resolved on the first call there)
static int kl_static_scratch(double** ptr_out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  VMF_REQUIRE(e == hipSuccess, VMF_ERR_LAUNCH, "vmf_kl_fwd_static: hipGetDevice: %s", hipGetErrorString(e));
  VMF_REQUIRE(dev >= 0 && dev < VMF_MAX_DEVICES, VMF_ERR_UNSUPPORTED, "vmf_kl_fwd_static: device %d >= %d", dev,
              VMF_MAX_DEVICES);
  double* p = g_kl_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> hold(g_kl_scratch_lock);
    p = g_kl_scratch_ptr[dev].load(std::memory_order_relaxed);
    if (!p) {
      e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(vmf_kl_static_scratch));
      VMF_REQUIRE(e == hipSuccess && p, VMF_ERR_LAUNCH, "vmf_kl_fwd_static: hipGetSymbolAddress on device %d: %s", dev,
                  hipGetErrorString(e));
      g_kl_scratch_ptr[dev].store(p, std::memory_order_release);
    }
  }
  *ptr_out = p;
  return VMF_OK;
}

extern "C" int vmf_kl_fwd_static(const void* kappa, void* out, double* sum_out, double alpha, int64_t n, int variant,
                                 int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_kl_fwd_static");
  VMF_REQUIRE(n <= VMF_KL_STATIC_MAX_N, VMF_ERR_UNSUPPORTED,
              "vmf_kl_fwd_static: n=%lld exceeds VMF_KL_STATIC_MAX_N=%lld (use vmf_kl_fwd with a caller workspace)",
              (long long)n, (long long)VMF_KL_STATIC_MAX_N);
  double* scratch = nullptr;
  if (sum_out) {
    // the array is looked up for the CURRENT device: a stream that belongs to another GPU would launch there with this
    // GPU's address
    if (stream) {
      hipDevice_t sdev = -1;
      int cur = -1;
      if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && hipGetDevice(&cur) == hipSuccess)
        VMF_REQUIRE((int)sdev == cur, VMF_ERR_BAD_ARG,
                    "vmf_kl_fwd_static: the stream belongs to device %d, the current device is %d", (int)sdev, cur);
      else
        (void)hipGetLastError();  // a runtime without the query: nothing to compare
    }
    int rc = kl_static_scratch(&scratch);
    if (rc) return rc;
  }
  return kl_fwd_launch("vmf_kl_fwd_static", kappa, out, sum_out, scratch, alpha, n, variant, dtype,
                       (hipStream_t)stream);
}

extern "C" int vmf_kl_bwd(const void* kappa, const void* gout, int64_t gout_stride, double gscale, void* dkappa,
                          int64_t n, int variant, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_kl_bwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  VMF_REQUIRE(variant >= 0 && variant <= VMF_KL_STABLE_M2, VMF_ERR_BAD_ARG, "unknown KL variant %d", variant);
  VMF_REQUIRE(gout_stride == 0 || gout_stride == 1, VMF_ERR_BAD_ARG, "gout_stride must be 0 or 1");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(kappa && gout && dkappa, VMF_ERR_BAD_ARG, "vmf_kl_bwd: null pointer");
  KlBessel kb = make_kl_bessel();
  int grid = vmf_grid_for(n, 256);
  const bool vec = dtype == VMF_F32 && ew_aligned16(kappa) && ew_aligned16(dkappa) && (gout_stride == 0 || ew_aligned16(gout));
  VMF_REQUIRE(dtype == VMF_F32 || dtype == VMF_F64, VMF_ERR_DTYPE, "dtype %d is not VMF_F32/VMF_F64", dtype);
#define VMF_KL_BWD(V)                                                                                                   \
  case V:                                                                                                               \
    if (vec)                                                                                                            \
      hipLaunchKernelGGL((kl_bwd_kernel<float, 4, false, V>), dim3(vmf_grid_for((n + 3) / 4, 256)), dim3(256), 0,       \
                         (hipStream_t)stream, (const float*)kappa, (const float*)gout, gout_stride, (float)gscale,      \
                         (float*)dkappa, n, kb);                                                                        \
    else if (dtype == VMF_F32)                                                                                          \
      hipLaunchKernelGGL((kl_bwd_kernel<float, 1, true, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream,             \
                         (const float*)kappa, (const float*)gout, gout_stride, (float)gscale, (float*)dkappa, n, kb);   \
    else                                                                                                                \
      hipLaunchKernelGGL((kl_bwd_kernel<double, 1, true, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream,            \
                         (const double*)kappa, (const double*)gout, gout_stride, gscale, (double*)dkappa, n, kb);       \
    break
  switch (variant) {
    VMF_KL_BWD(0); VMF_KL_BWD(1); VMF_KL_BWD(2); VMF_KL_BWD(3); VMF_KL_BWD(4); VMF_KL_BWD(5);
    default: break;
  }
#undef VMF_KL_BWD
  return vmf_check_launch("vmf_kl_bwd");
}

// ---- uniform prior on the sphere / S^1 projected normal ---------------------------------------------------
// One thread per row; rows here are short (dim 2, 3, 10 in the reference's models).  Counter-based noise is
// generated twice (norm pass, write pass) instead of being kept in an indexed register array.
template <typename T>
__device__ __forceinline__ T row_normal(const T* g, int64_t i, int dim, int c, uint64_t row, uint32_t slot0,
                                        PhiloxKey key, T cache[4], int* cached_chunk) {
  if (g) return g[i * dim + c];
  int ch = c >> 2;
  if (ch != *cached_chunk) {
    philox_normal4<T>(row, slot0 + (uint32_t)ch, key, cache);
    *cached_chunk = ch;
  }
  return cache[c & 3];
}

template <typename T>
__global__ __launch_bounds__(256) void sphere_uniform_kernel(const T* __restrict__ gauss, T* __restrict__ out,
                                                             int64_t n, int dim, uint64_t row0, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T cache[4];
    int cc = -1;
    T ss = 0;
    for (int c = 0; c < dim; ++c) {
      T g = row_normal<T>(gauss, i, dim, c, row0 + (uint64_t)i, 0x20000u, key, cache, &cc);
      ss += g * g;
    }
    T nn = M<T>::sqrt(ss);  // sampler/spherical_uniform.py:33, no epsilon
    cc = -1;
    for (int c = 0; c < dim; ++c)
      out[i * dim + c] = row_normal<T>(gauss, i, dim, c, row0 + (uint64_t)i, 0x20000u, key, cache, &cc) / nn;
  }
}

// Rows of up to 16 floats, fp32, in-kernel noise: ONE lane owns a row.  Its normals are drawn once and stay in
// registers (the kernel above walks a row twice, drawing its noise both times, and stores 4-byte elements a row apart
// from lane to lane: 1.0-1.8 TB/s); the wave's tile is written into a wave-private LDS slab and leaves as a flat array
// of 16-byte units, fully coalesced whatever the row width.  Same counters, same normals; the norm through the hardware
// reciprocal square root.
template <int D>
__global__ __launch_bounds__(256) void sphere_uniform_rows_kernel(float* __restrict__ out, int64_t n, uint64_t row0,
                                                                  PhiloxArg ka, int aligned) {
  constexpr int TILE = D <= 4 ? 256 : (D <= 8 ? 128 : 64);  // rows per wave: the slab stays at 2-4 KiB
  constexpr int UNITS = TILE * D / 4, UPL = (UNITS + 63) / 64, CHUNKS = (D + 3) / 4;
  __shared__ __align__(16) float s_rows[4][TILE * D];
  const PhiloxKey key = philox_resolve(ka);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const srow = s_rows[wave];
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * TILE;
  if (tile0 >= n) return;  // (no workgroup barrier in this kernel)
  const int64_t left = n - tile0;
  const int ntile = left < TILE ? (int)left : TILE;
#pragma unroll
  for (int sub = 0; sub < TILE / 64; ++sub) {
    const int r = sub * 64 + lane;
    if (r < ntile) {
      float g[4 * CHUNKS];
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c)
        philox_normal4<float>(row0 + (uint64_t)(tile0 + r), 0x20000u + (uint32_t)c, key, g + 4 * c);
      float ss = 0.0f;
#pragma unroll
      for (int e = 0; e < D; ++e) ss += g[e] * g[e];
      const float inv = F<float>::rsqrt(ss);  // sampler/spherical_uniform.py:33, no epsilon
#pragma unroll
      for (int e = 0; e < D; ++e) srow[r * D + e] = g[e] * inv;
    }
  }
  wave_lds_sync();
  float* const dst = out + tile0 * D;
  if (aligned && ntile == TILE) {
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const int q = lane + 64 * s;
      if (UNITS % 64 == 0 || q < UNITS) store_nt(reinterpret_cast<const f32x4*>(srow)[q], dst + 4 * q);
    }
  } else {
    for (int e = lane; e < ntile * D; e += 64) dst[e] = srow[e];
  }
}
// Wider rows (17 .. 1024 floats), fp32, in-kernel noise: a row is split over LPR lanes (the smallest power of two that
// holds its four-element chunks, at most 64) x up to four passes; a lane draws its chunks once, the squared norm is
// summed inside the lane group with xor shuffles, and every chunk leaves as one 16-byte store when rows are whole
// chunks and aligned (guarded 4-byte stores otherwise).  A wave takes SU_STEPS consecutive steps of 64 / LPR rows.
constexpr int SU_STEPS = 8, SU_MAX_PASSES = 4;
// four consecutive floats with the alignment of one: a whole chunk of a row that is not 16-byte aligned still moves as
// one access
struct __attribute__((packed, aligned(4))) SuQuad {
  float v[4];
};
__global__ __launch_bounds__(256) void sphere_uniform_wide_kernel(float* __restrict__ out, int64_t n, int dim, int lpr,
                                                                  uint64_t row0, PhiloxArg ka, int vec) {
  const PhiloxKey key = philox_resolve(ka);
  const int lane = threadIdx.x & 63;
  const int rpi = 64 / lpr, sub = lane & (lpr - 1), grp = lane / lpr;
  const int chunks = (dim + 3) / 4;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_steps = (n + rpi - 1) / rpi;
  const int64_t step_end = (wave_id + 1) * SU_STEPS < n_steps ? (wave_id + 1) * SU_STEPS : n_steps;
  for (int64_t step = wave_id * SU_STEPS; step < step_end; ++step) {  // wave-uniform
    const int64_t r = step * rpi + grp;
    const bool valid = r < n;
    const int64_t row = valid ? r : 0;  // lanes past the end draw row 0 again (the shuffles need the whole wave) and store nothing
    float g[SU_MAX_PASSES][4];
    float ss = 0.0f;
#pragma unroll
    for (int p = 0; p < SU_MAX_PASSES; ++p) {
      const int c = p * lpr + sub;
      if (c < chunks) {
        philox_normal4<float>(row0 + (uint64_t)row, 0x20000u + (uint32_t)c, key, g[p]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (4 * c + i >= dim) g[p][i] = 0.0f;
          ss += g[p][i] * g[p][i];
        }
      }
    }
    for (int o = 1; o < lpr; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = F<float>::rsqrt(ss);  // sampler/spherical_uniform.py:33, no epsilon
    if (valid) {
      float* const orow = out + row * dim;
#pragma unroll
      for (int p = 0; p < SU_MAX_PASSES; ++p) {
        const int c = p * lpr + sub;
        if (c < chunks) {
          if (vec) {
            store_nt(f32x4{g[p][0] * inv, g[p][1] * inv, g[p][2] * inv, g[p][3] * inv}, orow + 4 * c);
          } else if (4 * c + 4 <= dim) {
            SuQuad q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q.v[i] = g[p][i] * inv;
            *reinterpret_cast<SuQuad*>(orow + 4 * c) = q;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (4 * c + i < dim) orow[4 * c + i] = g[p][i] * inv;
          }
        }
      }
    }
  }
}
static bool launch_sphere_uniform_wide(float* out, int64_t n, int dim, uint64_t row0, const PhiloxArg& key,
                                       hipStream_t s) {
  const int chunks = (dim + 3) / 4;
  if (dim <= 16 || chunks > 64 * SU_MAX_PASSES) return false;
  int lpr = 1;
  while (lpr < chunks && lpr < 64) lpr <<= 1;
  const int64_t steps = (n + 64 / lpr - 1) / (64 / lpr);
  const int64_t blocks = ((steps + SU_STEPS - 1) / SU_STEPS + 3) / 4;
  if (blocks >= 0x7fffffff) return false;
  const int vec = (dim % 4 == 0 && ew_aligned16(out)) ? 1 : 0;
  hipLaunchKernelGGL(sphere_uniform_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, dim, lpr, row0, key,
                     vec);
  return true;
}
static bool launch_sphere_uniform_rows(float* out, int64_t n, int dim, uint64_t row0, const PhiloxArg& key,
                                       hipStream_t s) {
  const int tile = dim <= 4 ? 256 : (dim <= 8 ? 128 : 64);
  const int64_t blocks = ((n + tile - 1) / tile + 3) / 4;
  if (blocks >= 0x7fffffff) return false;
  const int aligned = ew_aligned16(out) ? 1 : 0;
  const dim3 grid((unsigned)blocks), block(256);
  switch (dim) {
#define VMF_SU(D) case D: hipLaunchKernelGGL((sphere_uniform_rows_kernel<D>), grid, block, 0, s, out, n, row0, key, aligned); return true
    VMF_SU(2); VMF_SU(3); VMF_SU(4); VMF_SU(5); VMF_SU(6); VMF_SU(7); VMF_SU(8); VMF_SU(9); VMF_SU(10); VMF_SU(11);
    VMF_SU(12); VMF_SU(13); VMF_SU(14); VMF_SU(15); VMF_SU(16);
#undef VMF_SU
    default: return false;
  }
}

extern "C" int vmf_sphere_uniform_sample(const void* gauss, void* out, int64_t n, int dim, uint64_t row0,
                                         uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int dtype,
                                         vmf_stream_t stream) {
  VMF_RANGE("vmf_sphere_uniform_sample");
  VMF_REQUIRE(n >= 0 && dim >= 1, VMF_ERR_BAD_ARG, "bad n/dim");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(out, VMF_ERR_BAD_ARG, "vmf_sphere_uniform_sample: null out");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  if (dtype == VMF_F32 && !gauss &&
      (launch_sphere_uniform_rows((float*)out, n, dim, row0, key, (hipStream_t)stream) ||
       launch_sphere_uniform_wide((float*)out, n, dim, row0, key, (hipStream_t)stream)))
    return vmf_check_launch("vmf_sphere_uniform_sample");
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sphere_uniform_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)gauss, (T*)out, n, dim, row0, key));
  return vmf_check_launch("vmf_sphere_uniform_sample");
}

// Counter noise of the S^1 sampler at dim = 2 (the circle): a row needs two normals = one Box-Muller pair = two words, so
// ONE Philox call serves the rows 2p and 2p+1: Philox(index p = row >> 1, slot 0x30000), words (x, y) for the even row,
// (z, w) for the odd one.  (Wider rows keep one call per four elements: slot 0x30000 + chunk, index = row.)
template <typename T>
__device__ __forceinline__ void s1_circle_normals(uint64_t row, PhiloxKey key, T* n0, T* n1) {
  const U4 r = philox4x32_10(row >> 1, 0x30000u, key);
  const bool odd = (row & 1u) != 0u;
  box_muller<T>(u01<T>(odd ? r.z : r.x), u01<T>(odd ? r.w : r.y), n0, n1);
}

// regular_cnn/s1_vae.py:48-55: z = F.normalize(mean*kappa + noise)  (eps 1e-12)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void s1_projnormal_kernel(const T* __restrict__ mean, const T* __restrict__ kappa,
                                                            const T* __restrict__ noise, const T* __restrict__ gz,
                                                            T* __restrict__ z, T* __restrict__ dmean,
                                                            T* __restrict__ dkappa, int64_t n, int dim,
                                                            uint64_t row0, PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T cache[4];
    int cc = -1;
    if (dim == 2 && !noise) {  // the circle's noise layout: see s1_circle_normals
      s1_circle_normals<T>(row0 + (uint64_t)i, key, &cache[0], &cache[1]);
      cc = 0;
    }
    T k = kappa[i], ss = 0, yg = 0;
    for (int c = 0; c < dim; ++c) {
      T y = mean[i * dim + c] * k + row_normal<T>(noise, i, dim, c, row0 + (uint64_t)i, 0x30000u, key, cache, &cc);
      ss += y * y;
      if (BWD) yg += y * gz[i * dim + c];
    }
    T nn = M<T>::sqrt(ss);
    bool clamped = nn < (T)1e-12;
    if (clamped) nn = (T)1e-12;
    if (!(dim == 2 && !noise)) cc = -1;  // (the circle's pair of normals is still in the cache)
    T dk = 0;
    for (int c = 0; c < dim; ++c) {
      T y = mean[i * dim + c] * k + row_normal<T>(noise, i, dim, c, row0 + (uint64_t)i, 0x30000u, key, cache, &cc);
      if (!BWD) {
        z[i * dim + c] = y / nn;
      } else {
        T dy = clamped ? gz[i * dim + c] / nn : gz[i * dim + c] / nn - y * yg / (nn * nn * nn);
        dmean[i * dim + c] = k * dy;
        dk += mean[i * dim + c] * dy;
      }
    }
    if (BWD) dkappa[i] = dk;
  }
}

// dim == 2 (the circle, the only shape the reference's S^1 VAE uses): a thread owns the GLOBAL pair of rows 2p, 2p+1 --
// one 128-bit access per operand for both rows (aligned to the row, 8 bytes in fp32, which is all a global access needs
// here) and ONE Philox call for both rows' Box-Muller pairs.  A call whose
// first or last row has no partner in the call (odd row offset, odd count) handles that row alone with 64-bit accesses.
// Same arithmetic as the general kernel above.
template <typename T, bool BWD>
__device__ __forceinline__ void s1_circle_row(T m0, T m1, T k, T n0, T n1, T g0, T g1, T* o0, T* o1, T* dk) {
  const T y0 = m0 * k + n0, y1 = m1 * k + n1;
  T ss = 0;
  ss += y0 * y0;
  ss += y1 * y1;
  // 1 / max(|y|, 1e-12): the 1-ulp hardware reciprocal square root for fp32, the library for fp64
  const T nn = F<T>::sqrt(ss);
  const bool clamped = nn < (T)1e-12;
  const T rn = clamped ? (T)1e12 : F<T>::rsqrt(ss);
  if (!BWD) {
    *o0 = y0 * rn; *o1 = y1 * rn;
  } else {
    T yg = 0;
    yg += y0 * g0;
    yg += y1 * g1;
    const T t = clamped ? (T)0 : yg * (rn * rn * rn);
    const T dy0 = g0 * rn - y0 * t;
    const T dy1 = g1 * rn - y1 * t;
    *o0 = k * dy0; *o1 = k * dy1;
    T d = 0;
    d += m0 * dy0;
    d += m1 * dy1;
    *dk = d;
  }
}
// KP: kappa / d kappa of a pair are one aligned 64-bit unit each (even row offset, arrays aligned to two elements)
template <typename T, bool BWD, bool KP>
__global__ __launch_bounds__(256) void s1_projnormal2_kernel(const T* __restrict__ mean, const T* __restrict__ kappa,
                                                             const T* __restrict__ noise, const T* __restrict__ gz,
                                                             T* __restrict__ z, T* __restrict__ dmean,
                                                             T* __restrict__ dkappa, int64_t n, uint64_t row0,
                                                             PhiloxArg ka) {
  const PhiloxKey key = philox_resolve(ka);
  const uint64_t p0 = row0 >> 1;
  const int64_t npairs = (int64_t)(((row0 + (uint64_t)n + 1u) >> 1) - p0);  // global pairs the call touches
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  T* const o = BWD ? dmean : z;
  // forward (12 B read, 8 B written per row -- a near 1:1 stream): blocks dealt so that an XCD works on one
  // contiguous eighth of the rows, 0.439 -> 0.421 ms per 2^27 rows; the read-heavy backward measures 2 % slower that way
  const int64_t b0 = BWD ? (int64_t)blockIdx.x : (int64_t)xcd_block(blockIdx.x, gridDim.x);
  for (int64_t t = b0 * blockDim.x + threadIdx.x; t < npairs; t += stride) {
    const uint64_t P = p0 + (uint64_t)t;
    const int64_t ia = (int64_t)(2 * P - row0), ib = ia + 1;  // local indices of the pair's rows (ia may be -1)
    const bool va = ia >= 0, vb = ib < n;
    T nz[4] = {0, 0, 0, 0};
    if (!noise) {
      const U4 r = philox4x32_10(P, 0x30000u, key);
      box_muller<T>(u01<T>(r.x), u01<T>(r.y), &nz[0], &nz[1]);
      box_muller<T>(u01<T>(r.z), u01<T>(r.w), &nz[2], &nz[3]);
    }
    if (va && vb) {
      // every array is touched once: non-temporal accesses
      const RowPair<T> mu = load_rows2_nt<T>(mean, ia);
      T ka_, kb_;
      if (KP) {
        const Pair<T> k2 = load_pair_nt<T>(kappa, ia >> 1);
        ka_ = k2.a; kb_ = k2.b;
      } else {
        ka_ = __builtin_nontemporal_load(kappa + ia); kb_ = __builtin_nontemporal_load(kappa + ib);
      }
      if (noise) {
        const RowPair<T> e = load_rows2_nt<T>(noise, ia);
        nz[0] = e.v[0]; nz[1] = e.v[1]; nz[2] = e.v[2]; nz[3] = e.v[3];
      }
      RowPair<T> g = {};
      if (BWD) g = load_rows2_nt<T>(gz, ia);
      RowPair<T> r;
      T dka = 0, dkb = 0;
      s1_circle_row<T, BWD>(mu.v[0], mu.v[1], ka_, nz[0], nz[1], g.v[0], g.v[1], &r.v[0], &r.v[1], &dka);
      s1_circle_row<T, BWD>(mu.v[2], mu.v[3], kb_, nz[2], nz[3], g.v[2], g.v[3], &r.v[2], &r.v[3], &dkb);
      store_rows2_nt<T>(o, ia, r);
      if (BWD) {
        if (KP) {
          store_pair_nt<T>(dkappa, ia >> 1, dka, dkb);
        } else {
          __builtin_nontemporal_store(dka, dkappa + ia);
          __builtin_nontemporal_store(dkb, dkappa + ib);
        }
      }
    } else {
      const int64_t i = va ? ia : ib;  // the call's first row (odd global index) or its last one (even)
      const int h = va ? 0 : 2;
      const Pair<T> mu = load_pair_nt<T>(mean, i);
      const T k = kappa[i];
      if (noise) {
        const Pair<T> e = load_pair_nt<T>(noise, i);
        nz[h] = e.a; nz[h + 1] = e.b;
      }
      Pair<T> g = {};
      if (BWD) g = load_pair_nt<T>(gz, i);
      T r0, r1, dk = 0;
      s1_circle_row<T, BWD>(mu.a, mu.b, k, nz[h], nz[h + 1], g.a, g.b, &r0, &r1, &dk);
      store_pair_nt<T>(o, i, r0, r1);
      if (BWD) dkappa[i] = dk;
    }
  }
}
template <typename T>
static bool pair_aligned(const void* p) { return ((uintptr_t)p & (2 * sizeof(T) - 1)) == 0; }

extern "C" int vmf_s1_projnormal_fwd(const void* mean, const void* kappa, const void* noise, void* z, int64_t n,
                                     int dim, uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                     int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_s1_projnormal_fwd");
  VMF_REQUIRE(n >= 0 && dim >= 1, VMF_ERR_BAD_ARG, "bad n/dim");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mean && kappa && z, VMF_ERR_BAD_ARG, "vmf_s1_projnormal_fwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  // the circle kernel takes one row per thread up to 2^30 rows (backward: 0.87 -> 0.80 ms for 2^27 rows against a
  // 2048-workgroup grid-stride walk; forward unchanged)
  int grid = vmf_grid_for(dim == 2 ? n / 2 + 1 : n, 256, dim == 2 ? (1 << 22) : 2048);
  VMF_DISPATCH_DTYPE(dtype, {
    if (dim == 2 && pair_aligned<T>(mean) && pair_aligned<T>(z) && pair_aligned<T>(noise) && (row0 & 1u) == 0 &&
        pair_aligned<T>(kappa))
      hipLaunchKernelGGL((s1_projnormal2_kernel<T, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)nullptr, (T*)z, (T*)nullptr,
                         (T*)nullptr, n, row0, key);
    else if (dim == 2 && pair_aligned<T>(mean) && pair_aligned<T>(z) && pair_aligned<T>(noise))
      hipLaunchKernelGGL((s1_projnormal2_kernel<T, false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)nullptr, (T*)z, (T*)nullptr,
                         (T*)nullptr, n, row0, key);
    else
      hipLaunchKernelGGL((s1_projnormal_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)nullptr, (T*)z, (T*)nullptr,
                         (T*)nullptr, n, dim, row0, key);
  });
  return vmf_check_launch("vmf_s1_projnormal_fwd");
}

extern "C" int vmf_s1_projnormal_bwd(const void* mean, const void* kappa, const void* noise, const void* gz,
                                     void* dmean, void* dkappa, int64_t n, int dim, uint64_t row0, uint64_t seed,
                                     uint64_t offset, const uint64_t* offset_dev, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_s1_projnormal_bwd");
  VMF_REQUIRE(n >= 0 && dim >= 1, VMF_ERR_BAD_ARG, "bad n/dim");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mean && kappa && gz && dmean && dkappa, VMF_ERR_BAD_ARG, "vmf_s1_projnormal_bwd: null pointer");
  PhiloxArg key = make_philox_arg(seed, offset, offset_dev);
  // the circle kernel takes one row per thread up to 2^30 rows (backward: 0.87 -> 0.80 ms for 2^27 rows against a
  // 2048-workgroup grid-stride walk; forward unchanged)
  int grid = vmf_grid_for(dim == 2 ? n / 2 + 1 : n, 256, dim == 2 ? (1 << 22) : 2048);
  VMF_DISPATCH_DTYPE(dtype, {
    if (dim == 2 && pair_aligned<T>(mean) && pair_aligned<T>(gz) && pair_aligned<T>(dmean) && pair_aligned<T>(noise) &&
        (row0 & 1u) == 0 && pair_aligned<T>(kappa) && pair_aligned<T>(dkappa))
      hipLaunchKernelGGL((s1_projnormal2_kernel<T, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)gz, (T*)nullptr, (T*)dmean,
                         (T*)dkappa, n, row0, key);
    else if (dim == 2 && pair_aligned<T>(mean) && pair_aligned<T>(gz) && pair_aligned<T>(dmean) && pair_aligned<T>(noise))
      hipLaunchKernelGGL((s1_projnormal2_kernel<T, true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)gz, (T*)nullptr, (T*)dmean,
                         (T*)dkappa, n, row0, key);
    else
      hipLaunchKernelGGL((s1_projnormal_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const T*)mean, (const T*)kappa, (const T*)noise, (const T*)gz, (T*)nullptr, (T*)dmean,
                         (T*)dkappa, n, dim, row0, key);
  });
  return vmf_check_launch("vmf_s1_projnormal_bwd");
}
