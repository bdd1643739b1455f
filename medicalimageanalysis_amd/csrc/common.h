// common.h -- shared device helpers for libvmf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vmf_hip.h"

// Per-element math is __host__ __device__ so that a host-only check library (tests/) can run the very
// same functions on the CPU; no product entry point has a CPU path.
#define VMF_HD __host__ __device__ __forceinline__

// ---- error plumbing -------------------------------------------------------------------------
void vmf_set_error(const char* fmt, ...);
int vmf_check_launch(const char* what);

#define VMF_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      vmf_set_error(__VA_ARGS__);      \
      return (code);                   \
    }                                  \
  } while (0)

// Marker range around an entry point, visible in `rocprofv3 --marker-trace`.  Off unless the environment has
// VMF_ROCTX=1; the marker library is looked up at run time, so the shared library links the HIP runtime only.
struct VmfRange {
  explicit VmfRange(const char* name);
  ~VmfRange();
  bool on;
};
#define VMF_RANGE(name) VmfRange vmf_range_guard_(name)

// Launches `expr` once with T = float or double.
#define VMF_DISPATCH_DTYPE(dtype, ...)                    \
  do {                                                    \
    if ((dtype) == VMF_F32) {                             \
      using T = float;                                    \
      __VA_ARGS__;                                        \
    } else if ((dtype) == VMF_F64) {                      \
      using T = double;                                   \
      __VA_ARGS__;                                        \
    } else {                                              \
      vmf_set_error("dtype %d is not VMF_F32/VMF_F64", (int)(dtype)); \
      return VMF_ERR_DTYPE;                               \
    }                                                     \
  } while (0)

// Memory-bound elementwise launches: cap the grid and grid-stride the rest.
static inline int vmf_grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Workgroup i runs on XCD i % 8 (round-robin dispatch), each XCD with its own L2 and translation cache.  This remap
// gives every XCD one contiguous eighth of the forward kernel's tile blocks instead of every eighth block, so that an
// XCD streams through one region of mu and z.  Measured on 2^24 x 64 fp32 (A/B in one process, both orders twice):
// 1.681-1.685 ms plain, 1.635-1.649 ms with eighths; sixteenths the same, 32 / 64 regions 1.654 / 1.662, short runs
// per XCD worse than plain (runs of 16 blocks: 1.794, 128: 1.688, 1024: 1.670).  m = 256 / 512: -4 % / -2.5 %; rows of
// <= 16 elements, the backward and the log_prob kernels measure the same either way and keep the plain order.
// The thread-per-row S^1 kernel is 3-7 % slower with it (0.64 -> 0.67 ms forward for 2^27 rows) and keeps the plain order.
// VMF_XCD_RUN: -1 plain order, 0 grid / 2^VMF_XCD_SHIFT blocks per run, > 0 that many blocks per run.
#ifndef VMF_XCD_SHIFT
#define VMF_XCD_SHIFT 3
#endif
#ifndef VMF_XCD_RUN
#define VMF_XCD_RUN 0
#endif
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t nb) {
#if VMF_XCD_RUN < 0
  return b;
#else
  const uint32_t run = VMF_XCD_RUN > 0 ? (uint32_t)VMF_XCD_RUN : nb >> VMF_XCD_SHIFT;
  const uint32_t span = run << 3;
  const uint32_t base = span ? (b / span) * span : 0u;
  if (span == 0u || base + span > nb) return b;  // the ragged end of the grid keeps the plain order
  const uint32_t r = b - base;
  return base + (r & 7u) * run + (r >> 3);
#endif
}

// ---- Philox4x32-10 --------------------------------------------------------------------------
struct PhiloxKey {
  uint32_t k0, k1, c3;  // key words and the 4th counter word (offset_lo)
  uint64_t i0;          // m = 3 entry points: global index of the call's first u element / eps row (added by the kernels)
};
// What a launch carries: seed, host offset and an optional DEVICE-resident 64-bit offset that is added to the host
// one when the kernel starts.  With the counter in device memory a captured HIP graph draws fresh noise on every
// replay (the host arguments of a captured launch are frozen).
struct PhiloxArg {
  uint32_t seed_lo, seed_hi, off_lo, off_hi;
  const uint64_t* dev_offset;
  uint64_t row0;
};
static inline PhiloxArg make_philox_arg(uint64_t seed, uint64_t offset, const uint64_t* dev_offset, uint64_t row0 = 0) {
  PhiloxArg a;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.off_lo = (uint32_t)offset; a.off_hi = (uint32_t)(offset >> 32);
  a.dev_offset = dev_offset;
  a.row0 = row0;
  return a;
}
VMF_HD PhiloxKey philox_key_from(uint64_t seed, uint64_t offset) {
  PhiloxKey k;
  k.k0 = (uint32_t)seed;
  k.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
  k.c3 = (uint32_t)offset;
  k.i0 = 0;
  return k;
}
// uniform across the grid: one scalar load
__device__ __forceinline__ PhiloxKey philox_resolve(const PhiloxArg& a) {
  uint64_t off = ((uint64_t)a.off_hi << 32) | a.off_lo;
  if (a.dev_offset) off += *a.dev_offset;
  PhiloxKey k = philox_key_from(((uint64_t)a.seed_hi << 32) | a.seed_lo, off);
  k.i0 = a.row0;
  return k;
}

struct U4 {
  uint32_t x, y, z, w;
};

// (VMF_PHILOX_ROUNDS exists for timing ablations only: every test pins the 10-round generator)
#ifndef VMF_PHILOX_ROUNDS
#define VMF_PHILOX_ROUNDS 10
#endif
// a ^ b ^ c in one instruction (gfx950's three-input bit operation, truth table 0x96): a Philox round is two wide
// multiplies and two such folds instead of four two-input ones -- 58 -> 39 instructions per call
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ U4 philox4x32_10(uint64_t index, uint32_t slot, PhiloxKey key) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = slot, c3 = key.c3;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int r = 0; r < VMF_PHILOX_ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// 23-bit uniform strictly inside (0,1): ((x >> 9) + 1/2) * 2^-23, exact in fp32 and fp64.
template <typename T>
__device__ __forceinline__ T u01(uint32_t x) {
  // one shift, one convert, one fma; every step is exact
  return (T)(x >> 9) * (T)(1.0 / 8388608.0) + (T)(1.0 / 16777216.0);
}
// fp32: the same value in two instructions.  v_alignbit puts the 23 bits under the exponent of 1.0, giving
// 1 + (x >> 9) 2^-23 in [1, 2); subtracting 1 - 2^-24 (the float below 1) is exact.
template <>
__device__ __forceinline__ float u01<float>(uint32_t x) {
  return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, x, 9u)) - 0.99999994f;
}

// ---- type-generic math: accurate (ocml) versions --------------------------------------------
template <typename T> struct M;
template <> struct M<float> {
  static VMF_HD float log(float x) { return ::logf(x); }
  static VMF_HD float exp(float x) { return ::expf(x); }
  static VMF_HD float sqrt(float x) { return ::sqrtf(x); }
  static VMF_HD float sinh(float x) { return ::sinhf(x); }
  static VMF_HD float cosh(float x) { return ::coshf(x); }
  static VMF_HD float abs(float x) { return ::fabsf(x); }
  static VMF_HD float log1p(float x) { return ::log1pf(x); }
  static VMF_HD float expm1(float x) { return ::expm1f(x); }
  static VMF_HD float div(float a, float b) { return a / b; }
  static __device__ __forceinline__ void sincospi2(float u, float* s, float* c) { ::sincospif(2.0f * u, s, c); }
};
template <> struct M<double> {
  static VMF_HD double log(double x) { return ::log(x); }
  static VMF_HD double exp(double x) { return ::exp(x); }
  static VMF_HD double sqrt(double x) { return ::sqrt(x); }
  static VMF_HD double sinh(double x) { return ::sinh(x); }
  static VMF_HD double cosh(double x) { return ::cosh(x); }
  static VMF_HD double abs(double x) { return ::fabs(x); }
  static VMF_HD double log1p(double x) { return ::log1p(x); }
  static VMF_HD double expm1(double x) { return ::expm1(x); }
  static VMF_HD double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ void sincospi2(double u, double* s, double* c) { ::sincospi(2.0 * u, s, c); }
};

// ---- single-precision fast math usable from host and device code (the device side maps to one hardware instruction
//      each; the host side, used only by the host-side checks of the same formulas, to libm) ----------------------
VMF_HD float fast_exp2(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_exp2f(x);
#else
  return ::exp2f(x);
#endif
}
VMF_HD float fast_log2(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_logf(x);
#else
  return ::log2f(x);
#endif
}
// e^x: the product x * log2(e) is split so that its rounding error (up to 1e-5 relative at |x| ~ 88) is fed back
VMF_HD float fast_exp(float x) {
  const float L2E = 1.44269504088896341f, L2E_LO = 1.925963033500097e-8f;  // log2(e) = L2E + L2E_LO in fp32 pairs
  const float t = x * L2E;
  // beyond the exponent range the result is 0 or inf whatever r is (and an infinite t would make r NaN)
  const float r = ::fabsf(t) < 256.0f ? ::fmaf(x, L2E, -t) + x * L2E_LO : 0.0f;
  return fast_exp2(t) * (1.0f + 0.69314718056f * r);
}
// a / b through the hardware reciprocal (1 ulp) instead of the IEEE division sequence (~10 instructions)
VMF_HD float fast_rcp(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_rcpf(x);
#else
  // the hardware reciprocal has no denormal results: mimic that so host-side checks see what the device computes
  const float r = 1.0f / x;
  return ::fabsf(r) < 1.17549435e-38f ? ::copysignf(0.0f, r) : r;
#endif
}
VMF_HD float fast_log(float x) { return 0.69314718055994530942f * fast_log2(x); }

VMF_HD float fast_rsqrt(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / ::sqrtf(x);
#endif
}

VMF_HD float fast_sqrt(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_sqrtf(x);
#else
  return ::sqrtf(x);
#endif
}

// Single-precision math policy with the interface of M<float>, built on the 1-ulp hardware instructions: what the fp32
// m = 3 kernels run (the accurate M<float> costs 20-30 instructions per log / exp / division, several times the rest of
// a row's arithmetic).  e^x needs no split product here: its arguments are differences of logarithms, at most a few
// tens in magnitude where the result matters.
struct MFast {
  static VMF_HD float log(float x) { return fast_log(x); }
  static VMF_HD float exp(float x) { return fast_exp2(1.44269504088896340736f * x); }
  static VMF_HD float sqrt(float x) { return fast_sqrt(x); }
  static VMF_HD float abs(float x) { return ::fabsf(x); }
  static VMF_HD float div(float a, float b) { return a * fast_rcp(b); }
};

// ---- hot-path math: hardware transcendentals for float, the accurate library for double --------------------
// v_log_f32 / v_exp_f32 / v_sqrt_f32 / v_rsq_f32 / v_rcp_f32 are 1-ulp single instructions; v_sin_f32 / v_cos_f32
// take their argument in revolutions, which is exactly what Box-Muller's 2 pi u needs (no range reduction).
template <typename T> struct F;
template <> struct F<float> {
  static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
  static __device__ __forceinline__ float log(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ void sincos2pi(float u, float* s, float* c) {
    *s = __builtin_amdgcn_sinf(u);
    *c = __builtin_amdgcn_cosf(u);
  }
  static __device__ __forceinline__ float cos2pi(float u) { return __builtin_amdgcn_cosf(u); }
  static __device__ __forceinline__ float sin2pi(float u) { return __builtin_amdgcn_sinf(u); }
  // (log(1+s) - s) / ln 2 for s > -1, evaluated directly: the rounding of 1+s bounds the absolute error by
  // 2^-24 / ln 2, which the caller multiplies by (m-1) -- negligible against log2 of a uniform.
  static __device__ __forceinline__ float log1p_minus_over_ln2(float s) {
    return __builtin_amdgcn_logf(1.0f + s) - 1.44269504088896340736f * s;
  }
  // 1 - 2^t for t <= 0 (t = y / ln 2): short series where the difference cancels
  static __device__ __forceinline__ float one_minus_exp2(float t) {
    const float y = 0.69314718055994530942f * t;
    const float poly = -y * (1.0f + y * (0.5f + y * (0.166666667f + y * 0.0416666667f)));
    const float direct = 1.0f - __builtin_amdgcn_exp2f(t);
    return y > -0.03125f ? poly : direct;
  }
};
template <> struct F<double> {
  static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
  static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ void sincos2pi(double u, double* s, double* c) { ::sincospi(2.0 * u, s, c); }
  static __device__ __forceinline__ double cos2pi(double u) { return ::cospi(2.0 * u); }
  static __device__ __forceinline__ double sin2pi(double u) { return ::sinpi(2.0 * u); }
  static __device__ __forceinline__ double log1p_minus_over_ln2(double s) {
    double r;
    if (::fabs(s) < 1e-2) {
      double acc = 0.0;  // s^2 (-1/2 + s/3 - s^2/4 ...), 12 terms: < 1e-24 relative at |s| = 1e-2
      for (int k = 13; k >= 2; --k) acc = acc * s + ((k & 1) ? 1.0 : -1.0) / (double)k;
      r = s * s * acc;
    } else {
      r = ::log1p(s) - s;
    }
    return r * 1.44269504088896340736;
  }
  static __device__ __forceinline__ double one_minus_exp2(double t) { return -::expm1(0.69314718055994530942 * t); }
};

// Box-Muller on two uniforms: (sqrt(-2 ln u0) cos 2pi u1, sqrt(-2 ln u0) sin 2pi u1)
template <typename T>
__device__ __forceinline__ void box_muller(T u0, T u1, T* n0, T* n1) {
  T r = F<T>::sqrt((T)-1.38629436111989061883 * F<T>::log2(u0));  // -2 ln u0
  T s, c;
  F<T>::sincos2pi(u1, &s, &c);
  *n0 = r * c;
  *n1 = r * s;
}

// four normals for elements 4*chunk .. 4*chunk+3 of row `index`
template <typename T>
__device__ __forceinline__ void philox_normal4(uint64_t index, uint32_t slot, PhiloxKey key, T out[4]) {
  U4 r = philox4x32_10(index, slot, key);
  box_muller<T>(u01<T>(r.x), u01<T>(r.y), &out[0], &out[1]);
  box_muller<T>(u01<T>(r.z), u01<T>(r.w), &out[2], &out[3]);
}

// ---- cross-lane: DPP all-reduce inside groups of LPR consecutive lanes --------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f(double x) {
  long long b = __double_as_longlong(x);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// Sum over each aligned group of LPR lanes (LPR a power of two, 1..64); every lane gets the total.
template <int LPR, typename T>
__device__ __forceinline__ T group_allreduce_sum(T x) {
  if (LPR >= 2) x += dpp_f<DPP_QUAD_XOR1>(x);
  if (LPR >= 4) x += dpp_f<DPP_QUAD_XOR2>(x);
  if (LPR >= 8) x += dpp_f<DPP_ROW_HALF_MIRROR>(x);
  if (LPR >= 16) x += dpp_f<DPP_ROW_MIRROR>(x);
  if (LPR >= 32) x += __shfl_xor(x, 16, 64);
  if (LPR >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

// Three sums at once.  For float rows of 16 lanes the twelve DPP additions are written out by hand: left to itself the
// compiler packs two of the three chains into v_pk_add_f32, which cannot take a DPP operand, and pays two DPP moves
// per step for it (19 instructions instead of 12).  The three chains are interleaved so that every DPP read of a
// register comes two instructions after the write to it (the wait states that hazard needs); the leading s_nop covers
// the producers of the inputs.
template <int LPR, typename T>
__device__ __forceinline__ void group_allreduce_sum3(T& a, T& b, T& c) {
  a = group_allreduce_sum<LPR, T>(a);
  b = group_allreduce_sum<LPR, T>(b);
  c = group_allreduce_sum<LPR, T>(c);
}
#define VMF_DPP3(ctrl)                                                   \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <>
__device__ __forceinline__ void group_allreduce_sum3<16, float>(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
               VMF_DPP3("quad_perm:[1,0,3,2]") VMF_DPP3("quad_perm:[2,3,0,1]") VMF_DPP3("row_half_mirror")
               VMF_DPP3("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c));
}
// narrower rows: the first stages only
template <>
__device__ __forceinline__ void group_allreduce_sum3<8, float>(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
               VMF_DPP3("quad_perm:[1,0,3,2]") VMF_DPP3("quad_perm:[2,3,0,1]") VMF_DPP3("row_half_mirror")
               : "+v"(a), "+v"(b), "+v"(c));
}
template <>
__device__ __forceinline__ void group_allreduce_sum3<4, float>(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t" VMF_DPP3("quad_perm:[1,0,3,2]") VMF_DPP3("quad_perm:[2,3,0,1]")
               : "+v"(a), "+v"(b), "+v"(c));
}
template <>
__device__ __forceinline__ void group_allreduce_sum3<2, float>(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t" VMF_DPP3("quad_perm:[1,0,3,2]") : "+v"(a), "+v"(b), "+v"(c));
}
#undef VMF_DPP3
// five sums (the backward pass), same construction
template <int LPR, typename T>
__device__ __forceinline__ void group_allreduce_sum5(T& a, T& b, T& c, T& d, T& e) {
  group_allreduce_sum3<LPR, T>(a, b, c);
  d = group_allreduce_sum<LPR, T>(d);
  e = group_allreduce_sum<LPR, T>(e);
}
#define VMF_DPP5(ctrl)                                                   \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <>
__device__ __forceinline__ void group_allreduce_sum5<16, float>(float& a, float& b, float& c, float& d, float& e) {
  asm volatile("s_nop 1\n\t"
               VMF_DPP5("quad_perm:[1,0,3,2]") VMF_DPP5("quad_perm:[2,3,0,1]") VMF_DPP5("row_half_mirror")
               VMF_DPP5("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
#undef VMF_DPP5

// block-wide sum in double; result valid in thread 0.  `red` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum_double(double x, double* red) {
  x = group_allreduce_sum<64, double>(x);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wave] = x;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// one 64-bit (float) / 128-bit (double) access for a row of two values
template <typename T>
struct alignas(2 * sizeof(T)) Pair {
  T a, b;
};

// streaming (non-temporal) access to a Pair: for arrays a kernel reads or writes exactly once
template <typename T>
__device__ __forceinline__ Pair<T> load_pair_nt(const T* base, int64_t i) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  const V2 v = __builtin_nontemporal_load(reinterpret_cast<const V2*>(base) + i);
  Pair<T> r;
  r.a = v.x; r.b = v.y;
  return r;
}
template <typename T>
__device__ __forceinline__ void store_pair_nt(T* base, int64_t i, T a, T b) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  const V2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<V2*>(base) + i);
}

template <typename T>
struct RowPair {  // two rows of two values
  T v[4];
};
// streaming (non-temporal) access to two rows at once, aligned to one row only
template <typename T>
__device__ __forceinline__ RowPair<T> load_rows2_nt(const T* base, int64_t row) {
  typedef T V4 __attribute__((ext_vector_type(4), aligned(2 * sizeof(T))));
  const V4 v = __builtin_nontemporal_load(reinterpret_cast<const V4*>(base + 2 * row));
  RowPair<T> r;
  r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
  return r;
}
template <typename T>
__device__ __forceinline__ void store_rows2_nt(T* base, int64_t row, const RowPair<T>& r) {
  typedef T V4 __attribute__((ext_vector_type(4), aligned(2 * sizeof(T))));
  const V4 v = {r.v[0], r.v[1], r.v[2], r.v[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<V4*>(base + 2 * row));
}

// 16-byte-aligned group of four values (one 128-bit access for float, two for double)
template <typename T>
struct alignas(sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4) Vec4 {
  T v[4];
};

// ---- streaming stores and wave-private LDS hand-offs (shared by the m = 3 and general-m kernels) --------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// cache modifiers of streaming stores (measured on the general-m kernel, 2^24 x 64 rows: plain 1.000, nt 0.989,
// sc0 sc1 nt 0.967 with an nt load)
#ifndef VMF_ST_MOD
#define VMF_ST_MOD "sc0 sc1 nt"
#endif
// Streaming store of one 128-bit piece of z.  Issued from inline asm because the compiler loses the non-temporal
// mark of __builtin_nontemporal_store in the one-pass instantiations (their store came out without `nt`; the
// multi-pass ones kept it).  The s_nop covers the wait states a VALU write to the data registers needs after a store
// wider than 64 bits, which the compiler's hazard pass cannot see inside an asm block.  The memory counter only ever
// reads higher than the compiler assumes because of this store, so its own waits stay sufficient.
__device__ __forceinline__ void store_nt(f32x4 v, float* ptr) {
  asm volatile("global_store_dwordx4 %0, %1, off " VMF_ST_MOD "\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
}

// The w / 1-w^2 slabs are private to a wave: producer and consumer lanes belong to the same wave, whose LDS
// operations execute in program order.  All that is needed between the phases is that the compiler keeps that order
// and that outstanding LDS operations have retired; no workgroup barrier, and no wait on global memory.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

