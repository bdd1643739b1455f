// vmf_general.hip -- von Mises-Fisher on S^{m-1}, general m: rsample (Wood rejection + Householder),
// its backward, log_prob, entropy / KL.  gfx950 only.
//
// Data layout and work mapping of rsample
//   mu (n, m) and z (n, m) are dense row-major; a row is split over LPR = min(64, pow2 >= m/4) consecutive
//   lanes, each lane owning 4 consecutive elements per pass (one 128-bit access in fp32), P passes when
//   m > 256.  A wave therefore moves 64/LPR rows per step with fully coalesced 1 KiB loads and stores, and
//   the three row reductions (|g|^2, |e1-mu|^2, <g,mu>) are DPP butterflies inside the LPR-lane group: no
//   LDS traffic, no atomics.
//   Each wave owns ONE tile of tile_rows(LPR, P) consecutive rows (grid = all tiles; the block index is remapped so that
//   each of the 8 XCDs walks one contiguous eighth of the tiles, see xcd_block in common.h).  Phase 1 (one row per
//   lane) runs the rejection sampler for the tile's rows: lanes whose proposal was accepted retire the row (w goes to the wave's LDS
//   slab) and are refilled with the next unassigned row of the tile through a ballot + prefix popcount, so
//   the loop trip count follows the tile's MEAN number of proposals, not its maximum.  Phase 2 streams the
//   rows: load mu (the next step's rows are prefetched while the current step computes), make m-1 normals from
//   Philox in registers, reduce, reflect, store z (non-temporal).
//   Algorithmic HBM traffic: (2m+1) * sizeof(T) per row (mu in, kappa in, z out); the noise never touches
//   memory.
#include "vmf_math.h"

#include <type_traits>

// build-time tuning knobs (defaults are the measured optimum on MI355X for m = 64 fp32)
#ifndef VMF_PK_SUMS
#define VMF_PK_SUMS 1
#endif
#ifndef VMF_TILE_ROWS
#define VMF_TILE_ROWS 128
#endif
#ifndef VMF_WAVES
#define VMF_WAVES 4
#endif
constexpr int WAVES = VMF_WAVES;
// Waves per workgroup of the kernels that have no workgroup barrier (a one-wave workgroup gives its slot back the moment
// it ends; a four-wave one when its slowest wave does).  The lane-group backward and the log_prob dot kernel take theirs
// from the launch (blockDim).  Same-box A/Bs against four waves everywhere, fp32: backward of 2^24 rows of 8 / 16 / 64
// floats with one wave -2.5 / -2.8 / -3.5 %, 2^22 x 128 -1.1 %, 2^21 x 512 +9 % (stays at four); log_prob backward (dot
// kernel) 2^22 x 64 -4.1 %, 2^19 x 512 -1.7 %, its forward +3 % (stays at four); the lane-per-row log_prob kernels with
// two waves: rows of 16 floats -4 % forward, -12 % backward, rows of 10 the same as with four; the lane-per-row
// sampler backward measures the same with 1, 2 or 4.
#ifdef VMF_BWD_WAVES
static int bwd_waves(int, int) { return VMF_BWD_WAVES; }
#else
static int bwd_waves(int lpr, int kpasses) { return lpr * kpasses <= 32 ? 1 : WAVES; }
#endif
#ifdef VMF_LP_WAVES
static int lp_waves(bool) { return VMF_LP_WAVES; }
#else
static int lp_waves(bool bwd) { return bwd ? 1 : WAVES; }
#endif
#ifndef VMF_LPR_WAVES
#define VMF_LPR_WAVES 2
#endif
constexpr int RLB_WAVES = WAVES, LPR_WAVES = VMF_LPR_WAVES;
// mu, grad_z and x are read once by the backward / log_prob kernels: non-temporal 128-bit loads (2^24 x 64 fp32:
// backward 2.48 -> 2.38 ms, log_prob 1.58 -> 1.37 ms)
#ifndef VMF_BWD_STREAM
#define VMF_BWD_STREAM true
#endif
#ifndef VMF_LP_STREAM
#define VMF_LP_STREAM true
#endif
// Rows per tile.  The rejection phase ends with a tail in which the last few rows of the tile wait for their
// acceptance on a mostly idle wave; its cost is per tile, so narrow rows, whose phase 2 is short, take larger tiles:
// 384 rows for m <= 4 (one lane per row) and 320 for m <= 16; the one-row-per-lane circle instantiation 512 (dense
// aligned fp32 circles take vmf_circle_fwd_kernel instead).  Measured on 2^27 / 2^24 rows against
// 128-row tiles: m = 2 1.66 -> 1.22 ms, m = 3 0.259 -> 0.211, m = 4 0.220 -> 0.179, m = 8 .. 16 -4 .. -10 %.  512-row tiles are
// slower again (the LDS slabs then limit the resident waves), and walking several 128-row tiles per wave was 10-20 %
// slower than one (more waves in flight beat fewer wave launches).
#ifndef VMF_TILE_ROWS_LPR1
#define VMF_TILE_ROWS_LPR1 384
#endif
#ifndef VMF_TILE_ROWS_NARROW
#define VMF_TILE_ROWS_NARROW 320
#endif
#ifndef VMF_TILE_ROWS_CIRCLE
#define VMF_TILE_ROWS_CIRCLE 512
#endif
// Wide rows: the tile shrinks with the row so that a wave's footprint stays near VMF_TILE_ELEMS elements of mu and
// of z (32 KiB each in fp32, what 128 rows of m = 64 are), but not below VMF_TILE_ROWS_MIN rows.  Against 128-row
// tiles for every width (2^22 .. 2^23 rows, fp32): m = 128 1.667 -> 1.564 ms, m = 256 1.723 -> 1.543, m = 512
// 3.365 -> 3.043, m = 1024 3.344 -> 3.009; twice the footprint gives back half of that, half the footprint is
// 2-4 % slower again for m <= 256.
#ifndef VMF_TILE_ELEMS
#define VMF_TILE_ELEMS 8192
#endif
#ifndef VMF_TILE_ROWS_MIN
#define VMF_TILE_ROWS_MIN 8
#endif
constexpr int tile_rows(int lpr, int passes) {
  if (lpr == 1) return VMF_TILE_ROWS_LPR1;
  if (lpr <= 4) return VMF_TILE_ROWS_NARROW;
  const int fit = VMF_TILE_ELEMS / (4 * lpr * passes);
  return fit >= VMF_TILE_ROWS ? VMF_TILE_ROWS : (fit < VMF_TILE_ROWS_MIN ? VMF_TILE_ROWS_MIN : fit);
}
constexpr uint32_t SLOT_GAUSS = 0x10000u;
// Consecutive steps a wave of the backward / log_prob kernels takes (a step = 64 / LPR rows).  log_prob: 16 for every
// width.  Backward, measured per width on 2^21 .. 2^25 rows (ms at 1 / 2 / 4 / 8 / 16 / 32 steps): m = 16 1.26 / 1.16 /
// 1.16 / 1.25 / 1.26 / 1.28; m = 64 2.35 / 2.21 / 2.20 / 2.36 / 2.33 / 2.28; m = 128 - / 2.25 / 2.23 / 2.36 / 2.32 /
// 2.30; m = 256 flat (2.30 .. 2.35); m = 512 - / - / 4.65 / 4.63 / 4.42 / 4.86; m = 1024 - / 5.82 / 4.49 / 4.43 / 4.86 /
// 5.39.  Hence 4, except the two-pass shape (16) and the four- and eight-pass shapes (8).
// log_prob, lane-group kernel: steps (of 64 / LPR rows) per wave.  With non-temporal stores of d mu and d x, measured on
// 2^24 / 2^23 / 2^22 / 2^21 / 2^19 rows of 8 / 16 / 64 / 128 / 512 floats against 16 steps and plain stores:
// forward 2 steps -6 / -5 / -4 / -5 / -6 %; backward 4 steps -12 / -14 %, 1 step -15 / -15 / -16 % (the wide three)
#ifdef VMF_LP_STEPS
constexpr int lp_steps(bool, int) { return VMF_LP_STEPS; }
#else
constexpr int lp_steps(bool bwd, int lpr) { return bwd ? (lpr >= 8 ? 1 : 4) : 2; }
#endif
#ifdef VMF_BWD_STEPS
constexpr int bwd_steps(int, int) { return VMF_BWD_STEPS; }
#else
constexpr int bwd_steps(int lpr, int passes) { return (lpr < 64 || passes == 1) ? 4 : (passes == 2 ? 16 : 8); }
#endif

template <typename T>
struct GenParams {
  const T* mu; int64_t mu_stride;
  const T* kappa; int64_t kappa_stride;
  T* z; T* w_out; T* omw_out; int32_t* accept_idx; T* logp;
  const T* prop_e; const T* prop_u; const T* gauss;
  int64_t n; int m; int max_prop; uint64_t row0;
  int normalize;  // mu rows are raw: use mu / ||mu|| (and chain the gradient through the normalisation)
  int softplus;   // kappa holds raw head outputs: kappa = softplus(raw) + 1, gradient returned for the raw values
  const T* gkappa;  // backward: gradient with respect to the forward's kappa output (nullable)
  PhiloxArg ka; VmfDim dim;
  // backward only
  const T* w_in; const T* omw_in; const T* gz; T* dmu; T* dkappa;
  int dense16;  // one-lane-per-row backward: mu rows dense, mu / grad z / d mu 16-byte aligned (flat 128-bit tile moves)
};

// kappa of a row: the stored value, or softplus(raw) + 1 when the call passes raw head outputs (VMF_FLAG_SOFTPLUS_KAPPA)
template <typename T>
__device__ __forceinline__ T kappa_of(T stored, bool softplus, T* sig) {
  *sig = (T)1;
  if (!softplus) return stored;
  if (sizeof(T) == 4) return (T)v3_softplus_kappa<float, MFast>((float)stored, (float*)sig);
  return (T)v3_softplus_kappa<double, M<double>>((double)stored, (double*)sig);
}

// four consecutive elements with the alignment of one
template <typename T>
struct __attribute__((packed, aligned(sizeof(T)))) ElemAligned4 {
  T v[4];
};

// load the 4 elements [e0, e0+4) of a row (zero beyond m)
template <typename T, bool VEC, bool STREAM = false>
__device__ __forceinline__ void load4(const T* __restrict__ rowp, int e0, int m, T out[4]) {
  if (VEC) {
    if (e0 < m) {
      if (STREAM && sizeof(T) == 4) {  // read once by this kernel: non-temporal 128-bit load
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(rowp + e0));
        out[0] = (T)v.x; out[1] = (T)v.y; out[2] = (T)v.z; out[3] = (T)v.w;
      } else {
        Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(rowp + e0);
        out[0] = v.v[0]; out[1] = v.v[1]; out[2] = v.v[2]; out[3] = v.v[3];
      }
    } else {
      out[0] = out[1] = out[2] = out[3] = (T)0;
    }
  } else if (e0 + 4 <= m) {
    // a whole chunk of a row whose width or base is not a multiple of four elements: still ONE access, aligned to the
    // element only (global accesses need no more than that on this target; per-element accesses made these shapes run
    // at 37 % of the HBM rate, against 56-69 % for widths that are multiples of four)
    const ElemAligned4<T> v = *reinterpret_cast<const ElemAligned4<T>*>(rowp + e0);
    out[0] = v.v[0]; out[1] = v.v[1]; out[2] = v.v[2]; out[3] = v.v[3];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (e0 + i < m) ? rowp[e0 + i] : (T)0;
  }
}
template <typename T, bool VEC, bool STREAM = false>
__device__ __forceinline__ void store4(T* __restrict__ rowp, int e0, int m, const T in[4]) {
  if (VEC) {
    if (e0 < m) {
      if constexpr (STREAM && sizeof(T) == 4) {  // written once, read by a later kernel: non-temporal 128-bit store
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v;
        v.x = in[0]; v.y = in[1]; v.z = in[2]; v.w = in[3];
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(rowp + e0));
        return;
      }
      Vec4<T> v;
      v.v[0] = in[0]; v.v[1] = in[1]; v.v[2] = in[2]; v.v[3] = in[3];
      *reinterpret_cast<Vec4<T>*>(rowp + e0) = v;
    }
  } else if (e0 + 4 <= m) {
    ElemAligned4<T> v;
    v.v[0] = in[0]; v.v[1] = in[1]; v.v[2] = in[2]; v.v[3] = in[3];
    *reinterpret_cast<ElemAligned4<T>*>(rowp + e0) = v;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (e0 + i < m) rowp[e0 + i] = in[i];
  }
}

// normals for elements [e0, e0+4) of global row `grow` (grow_ptr: that row of the pre-drawn normals, or null for the
// counter-based stream); element 0 and elements >= m are zeroed
template <typename T, bool VEC>
__device__ __forceinline__ void noise4(const T* __restrict__ grow_ptr, uint64_t grow, int chunk, int m, PhiloxKey key,
                                       T g[4]) {
  int e0 = 4 * chunk;
  if (grow_ptr) {
    load4<T, VEC>(grow_ptr, e0, m, g);
  } else if (e0 < m) {
    philox_normal4<T>(grow, SLOT_GAUSS + (uint32_t)chunk, key, g);
  } else {
    g[0] = g[1] = g[2] = g[3] = (T)0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (e0 + i >= m || e0 + i == 0) g[i] = (T)0;
}

// Forward kernel.  MODE 0: scalar element access (m % 4 != 0 or unaligned rows), bounds checks, pre-drawn noise
// allowed.  MODE 1: 128-bit access, bounds checks (m < 4 LPR P), pre-drawn noise allowed.  MODE 2: the streaming
// configuration -- m == 4 LPR P exactly, 128-bit access, counter-based noise only, no per-element guards.  MODE 3: the
// circle -- m == 2 exactly, 64-bit access, counter-based noise only (LPR = P = 1).
template <typename T>
struct FwdParams {
  const T* mu; const T* kappa; T* z;
  T* w_out; T* omw_out; int32_t* accept_idx; T* logp; T* kappa_out;
  const T* prop_e; const T* prop_u; const T* gauss;
  int64_t n; uint64_t row0;
  int mu_stride, kappa_stride, m, max_prop;
  int normalize;  // mu rows are raw: the sample is drawn around mu / ||mu||
  int softplus;   // kappa holds raw head outputs: the tile prologue turns them into kappa = softplus(raw) + 1
  int dense16;    // one-lane-per-row kernel: mu rows are dense and mu / z 16-byte aligned (flat 128-bit tile moves)
  PhiloxArg ka;
  float lc_v, lc_lg, lc_c0;  // fp32 fused log_prob: m/2 - 1, lgamma(m/2), (m/2) log 2 pi
};

// Prefetch of the next step's mu rows for the fp32 streaming kernel.  The load is issued from inline asm so that its
// wait can be placed by hand: the memory counter retires operations in order, and a compiler-placed wait at the loop
// head would also wait for the store issued just before it -- the store's whole round trip, every step.  Rules that
// keep this correct: the destination is touched by nothing between prefetch_issue() and prefetch_wait(); the wait
// names the destination as an in/out operand so every use is ordered after it; vmcnt(0) over-waits at worst; and the
// destination must never be spilled between the two (the compiler would reuse the register while the load is in flight
// and the late data would land in whatever lives there then -- an instantiation squeezed under a register cap faulted
// that way): every instantiation that prefetches must have a private segment of zero, which a test reads off the built
// library.
// cache modifiers of the streaming kernel's mu load and z store.  A/B in one process on 2^24 x 64 rows (plain load +
// plain store = 1.000): nt store 0.989, nt load + nt store 0.974, nt load + system-scope nt store 0.967; scope bits
// without nt 0.985.  Every byte is touched once, so nothing is worth keeping in L2.
#ifndef VMF_LD_MOD
#define VMF_LD_MOD "nt"
#endif
#ifndef VMF_FULL_TILE_COPY
#define VMF_FULL_TILE_COPY 1
#endif
__device__ __forceinline__ void prefetch_issue(f32x4& dst, const float* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off " VMF_LD_MOD : "=&v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void prefetch_wait(f32x4& dst) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst) : : "memory");
}

// Second launch bound = waves per SIMD: the one-pass fp32 streaming shapes (m <= 256) fit 8 waves (<= 64 VGPRs)
// without spilling once the compiler is told to; the two-pass shape (m = 512) would spill 7 registers there and
// measures the same at 5, 6 or 8 waves, so it asks for 6; the narrow fp32 shapes (m <= 16) sit two registers above
// the 6-wave step without the hint; the other instantiations keep the default.  (The exact-fit m = 4 shape asks for
// 8 although its LDS slabs allow 6 -- the compiler notes it -- because the tighter register budget measures faster:
// 0.19 vs 0.23 ms for 2^24 rows.)
template <typename T, int LPR, int P, int MODE>
__global__ __launch_bounds__(WAVES * 64, (sizeof(T) == 4 && MODE == 2 && P == 1) ? 8 : ((sizeof(T) == 4 && (LPR <= 4 || (MODE == 2 && P == 2))) ? 6 : 1))
void vmf_rsample_fwd_kernel(FwdParams<T> p) {
  constexpr bool VEC = MODE == 1 || MODE == 2;
  constexpr bool FAST = MODE == 2;
  constexpr bool CIRCLE = MODE == 3;
  const PhiloxKey key = philox_resolve(p.ka);  // scalar: seed + host offset (+ device-resident offset)
  constexpr int TILE_ROWS = CIRCLE ? VMF_TILE_ROWS_CIRCLE : tile_rows(LPR, P);
  // one array for both slabs, so that phase 2 fetches a row's w and 1 - w^2 with a single two-address LDS read
  __shared__ T s_wc[WAVES][2][TILE_ROWS];
  // fp32 fused log_prob: log C_m(kappa) and kappa of the tile's rows (kappa is read from HBM once per row)
  // (not for the circle: one lane owns a row there, so log p is formed and stored in phase 2 directly)
  constexpr bool LP_SLAB = sizeof(T) == 4 && !CIRCLE;
  __shared__ float s_lc[LP_SLAB ? WAVES : 1][LP_SLAB ? TILE_ROWS : 1];
  __shared__ float s_kp[LP_SLAB ? WAVES : 1][LP_SLAB ? TILE_ROWS : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPI = 64 / LPR;  // rows per wave step
  const int sub = lane % LPR, grp = lane / LPR;
  const int m = FAST ? 4 * LPR * P : (CIRCLE ? 2 : p.m);
  const bool pre = !FAST && !CIRCLE && p.prop_e != nullptr;
  // Circle (m = 2) with counter noise: a proposal needs two uniforms (the Beta(1/2,1/2) angle and the accept draw)
  // and the tangent "direction" is one sign, so one Philox call serves two proposals; the counter layout, and the
  // arithmetic shared with vmf_circle_fwd_kernel, are described with circle_try in vmf_math.h.  The slabs hold w and
  // the signed sqrt(1 - w^2).
  // MODE 3 is this case with m a compile-time 2 and 64-bit accesses; MODE 0 takes it when rows are not 8-byte aligned.
  const bool circle = CIRCLE || (LPR == 1 && MODE == 0 && m == 2 && !pre && p.gauss == nullptr);
  T* const somw = s_wc[wave][0];
  T* const sw = s_wc[wave][1];
  float* const slc = s_lc[LP_SLAB ? wave : 0];
  float* const skp = s_kp[LP_SLAB ? wave : 0];

  {
    const uint32_t bid = xcd_block(blockIdx.x, gridDim.x);
    const int64_t tile = (int64_t)bid * WAVES + wave;
    const int64_t tile0 = tile * TILE_ROWS;  // wave-uniform
    const int64_t left = p.n - tile0;
    const int ntile = left <= 0 ? 0 : (left < TILE_ROWS ? (int)left : TILE_ROWS);
    const T* const kap_t = p.kappa + tile0 * p.kappa_stride;
    const uint64_t grow0 = p.row0 + (uint64_t)tile0;

    // ---- phase 1: rejection sampler, one row per lane; accepted lanes retire their row and are refilled with
    //      the next unassigned row of the tile (ballot + prefix popcount) ----
    // fp32 fused log_prob: log C_m(kappa) of the tile's rows, evaluated in single precision here (two rows per lane)
    // so that phase 2 writes log p once; fp64 outputs get theirs from the double-precision pre-pass kernel instead
    const bool lp32 = LP_SLAB && p.logp != nullptr;
    // (the exact-fit streaming instantiation is kept free of the softplus / kappa_out options, like `normalize`)
    const bool sp = !FAST && p.softplus;
    T* const kout = FAST ? nullptr : p.kappa_out;
    if (lp32 || !pre || kout) {
      for (int i = lane; i < ntile; i += 64) {
        T sig_;
        const T kap = kappa_of<T>(kap_t[i * p.kappa_stride], sp, &sig_);
        if (kout) kout[tile0 + i] = kap;
        if (lp32) {
          skp[i] = (float)kap;
          slc[i] = vmf_log_norm_f32v(p.lc_v, p.lc_lg, p.lc_c0, (float)kap);
        }
        // b(kappa) of every row of the tile, coalesced, parked in the slab that later receives 1 - w^2
        if (!pre) somw[i] = circle ? circle_b<T>(kap) : wood_b<T>(kap, m);
      }
      wave_lds_sync();
    }
    {
      int my = lane, next = 64, k = 0;
      bool active = my < ntile;
      T a = 0, b = 0, d = 0;
      bool neg = false;           // circle: the row's sign bit
      if (active) {
        T sig_;
        if (pre) wood_consts<T>(kappa_of<T>(kap_t[my * p.kappa_stride], sp, &sig_), m, &a, &b, &d);
        else b = somw[my];
      }
      while (__ballot(active) != 0ull) {
        bool acc = false;
        if (active) {
          T w = 0, omw = 0, c2 = 0, pc = 0, ps = 0;  // circle: cos^2 / den and sin^2 / den of the proposal kept
          if (pre) {
            const int64_t row = tile0 + my;
            acc = wood_try<T>(a, b, d, m, p.prop_e[row * p.max_prop + k], p.prop_u[row * p.max_prop + k], &w, &omw);
            c2 = omw * ((T)1 + w);
          } else if (LPR == 1 && circle) {
            const uint64_t grow = grow0 + (uint64_t)my;
            const T hb = circle_hb<T>(b);
            if (k == 0) {
              // the first proposal: half of the call this row shares with its neighbour
              const U4 rc = philox4x32_10(grow >> 1, 0u, key);
              const bool odd = (grow & 1u) != 0u;
              const uint32_t xv = odd ? rc.z : rc.x, xu = odd ? rc.w : rc.y;
              neg = (xv & 1u) != 0u;
              acc = circle_try<T>(b, hb, xv, xu, &pc, &ps);
            } else {
              // k is odd here: both proposals of the row's own call (k - 1) / 2 are evaluated in this iteration and the
              // first accepted one wins (the same decision sequence as one proposal per iteration, in half the calls --
              // lanes of a wave sit at different k, so skipping the call on even k would not shorten the iteration)
              const U4 rc = philox4x32_10(grow, 1u + (uint32_t)(k >> 1), key);
              T pc1, ps1;
              const bool a0 = circle_try<T>(b, hb, rc.x, rc.y, &pc, &ps);
              const bool a1 = circle_try<T>(b, hb, rc.z, rc.w, &pc1, &ps1);
              const bool use0 = a0 || k + 1 >= p.max_prop;
              acc = use0 || a1;
              if (!use0) { pc = pc1; ps = ps1; ++k; }
            }
          } else {
            const U4 r = philox4x32_10(grow0 + (uint64_t)my, (uint32_t)k, key);
            acc = wood_try_counter<T>(b, m, u01<T>(r.x), u01<T>(r.y), u01<T>(r.z), &w, &omw, &c2);
          }
          if (k + 1 >= p.max_prop) acc = true;  // the last proposal allowed is taken regardless
          if (acc) {
            // sqrt((1 - w)(1 + w)) -- 1 + w taken where it is known without cancellation -- is what phase 2 multiplies
            // by: the root is taken here, once per row, instead of by every lane of the row's group at every step
            T parked = F<T>::sqrt(c2 < (T)1e-10 ? (T)1e-10 : c2);
            if (LPR == 1 && circle) circle_finish<T>(b, pc, ps, neg, &w, &omw, &parked);
            sw[my] = w;
            somw[my] = parked;
            if (p.w_out) p.w_out[tile0 + my] = w;
            if (p.omw_out) p.omw_out[tile0 + my] = omw;
            if (p.accept_idx) p.accept_idx[tile0 + my] = k;
          }
          ++k;
        }
        const unsigned long long accm = __ballot(acc);
        if (acc) {
          my = next + __popcll(accm & ((1ull << lane) - 1ull));
          k = 0;
          active = my < ntile;
          if (active) {
            T sig_;
            if (pre) wood_consts<T>(kappa_of<T>(kap_t[my * p.kappa_stride], sp, &sig_), m, &a, &b, &d);
            else b = somw[my];
          }
        }
        next += __popcll(accm);
      }
    }
    wave_lds_sync();

    // ---- phase 2: stream the tile's rows, RPI rows per step ----
    // Two copies of the loop for the one-pass streaming instantiations: a full tile (all but the last of a launch) has a
    // compile-time row count, so the per-lane "row inside the tile" tests, the clamped prefetch address and their
    // execution masks drop out of the step: m = 64, 2^24 rows 1.493 -> 1.481 ms settled, mean of the 20 timed launches
    // of fresh bench processes 1.524 -> 1.511; m = 32 and 128 -1 %; the two- and three-pass shapes measure the same or
    // +0.5 % with it and keep the one general copy (as does everything with VMF_FULL_TILE_COPY=0).
    auto phase2 = [&](auto full_c) {
    constexpr bool FULLT = decltype(full_c)::value;
    const int nt = FULLT ? TILE_ROWS : ntile;
    // every lane group of every step holds a row of the tile (the tile heights of the three-pass shapes are not
    // multiples of the rows per step: their last step is partial in a full tile, too)
    constexpr bool EVEN = FULLT && TILE_ROWS % RPI == 0;
    const T* const mu_t = p.mu + tile0 * p.mu_stride;
    T* const z_t = p.z + tile0 * m;
    const T* const ga_t = (!FAST && !CIRCLE && p.gauss) ? p.gauss + tile0 * m : nullptr;
    // fp32 streaming mode keeps the NEXT step's mu rows in flight while the current step computes and stores.
    constexpr bool PF = FAST && sizeof(T) == 4;
    // ONE register set: a step copies it into its own mu registers first, so the next rows can land in it while the
    // step computes (a second set, copied over after the wait, cost four more 64-bit moves per step)
    f32x4 cur[P];
    if (PF && nt > 0) {
      const int rr0 = (EVEN || grp < nt) ? grp : 0;
#pragma unroll
      for (int j = 0; j < P; ++j)
        prefetch_issue(cur[j], reinterpret_cast<const float*>(mu_t + rr0 * p.mu_stride + 4 * (j * LPR + sub)));
#pragma unroll
      for (int j = 0; j < P; ++j) prefetch_wait(cur[j]);
    }
    const T gmask0 = sub == 0 ? (T)0 : (T)1;
    // element offset of the prefetched row inside the tile, advanced by addition (the product row * stride per step
    // is a quarter-rate instruction)
    int pf_off = (grp + RPI) * p.mu_stride;
    const int pf_step = RPI * p.mu_stride;
    for (int r0 = 0; r0 < nt; r0 += RPI) {
      const int r = r0 + grp;
      const bool valid = EVEN || r < nt;
      const int rr = valid ? r : 0;
      const T* mrow = mu_t + rr * p.mu_stride;
      if (LPR == 1 && circle) {
        // the circle: one lane, one row; w and the signed sqrt(1 - w^2) come from the slabs, the reflection from the
        // function vmf_circle_fwd_kernel uses
        T m0, m1;
        if (CIRCLE) {
          const Pair<T> v = *reinterpret_cast<const Pair<T>*>(mrow);
          m0 = v.a; m1 = v.b;
        } else {
          m0 = mrow[0]; m1 = mrow[1];
        }
        T z0, z1, mz;
        circle_compose<T>(m0, m1, sw[rr], somw[rr], p.normalize != 0, &z0, &z1, &mz);
        if (valid) {
          T* const zrow = z_t + rr * 2;
          if (CIRCLE) {
            Pair<T> o;
            o.a = z0; o.b = z1;
            *reinterpret_cast<Pair<T>*>(zrow) = o;
          } else {
            zrow[0] = z0; zrow[1] = z1;
          }
          if (p.logp) {
            T sig_;
            const T kap = kappa_of<T>(kap_t[rr * p.kappa_stride], sp, &sig_);
            if (sizeof(T) == 4) {
              const float lpv = vfma((float)kap, (float)mz, vmf_log_norm_circle_f32((float)kap));
              if (LP_SLAB) slc[rr] = lpv;  // (replaces the general-m log C the prologue parked; stored after the loop)
              else p.logp[tile0 + rr] = (T)lpv;
            } else {
              p.logp[tile0 + rr] += kap * mz;  // log C_2(kappa) came from the pre-pass
            }
          }
        }
        continue;
      }
      T mu[P][4], g[P][4];
      const bool more = r0 + RPI < nt;  // wave-uniform
      if (PF) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
          mu[j][0] = (T)cur[j].x; mu[j][1] = (T)cur[j].y; mu[j][2] = (T)cur[j].z; mu[j][3] = (T)cur[j].w;
        }
        if (more) {
          const int offn = (EVEN || r + RPI < nt) ? pf_off : 0;
#pragma unroll
          for (int j = 0; j < P; ++j)
            prefetch_issue(cur[j], reinterpret_cast<const float*>(mu_t + (uint32_t)offn + 4 * (j * LPR + sub)));
          pf_off += pf_step;
        }
      } else if (FAST) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(mrow + 4 * (j * LPR + sub));
          mu[j][0] = v.v[0]; mu[j][1] = v.v[1]; mu[j][2] = v.v[2]; mu[j][3] = v.v[3];
        }
      }
      T gg = 0, dd = 0, gm = 0;
      f32x2 gg2 = {0.0f, 0.0f}, dd2 = {0.0f, 0.0f}, gm2 = {0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int chunk = j * LPR + sub;
        if (FAST) {
          philox_normal4<T>(grow0 + (uint64_t)rr, SLOT_GAUSS + (uint32_t)chunk, key, g[j]);
          // element 0 carries no tangent noise; a multiplication by a lane constant (the normals are finite) rather
          // than a select, which the compiler widens to the whole 4-vector
          if (j == 0) g[0][0] *= gmask0;
        } else {
          load4<T, VEC>(mrow, 4 * chunk, m, mu[j]);
          noise4<T, VEC>(ga_t ? ga_t + (int64_t)rr * m : nullptr, grow0 + (uint64_t)rr, chunk, m, key, g[j]);
        }
        if (VMF_PK_SUMS && FAST && sizeof(T) == 4) {
          // the three partial sums in packed pairs (v_pk_mul / v_pk_fma: two lanes of arithmetic per instruction)
          const f32x2 ga = {(float)g[j][0], (float)g[j][1]}, gb = {(float)g[j][2], (float)g[j][3]};
          const f32x2 ma = {(float)mu[j][0], (float)mu[j][1]}, mb = {(float)mu[j][2], (float)mu[j][3]};
          f32x2 mq = ma;  // mu without its first element
          if (j == 0) mq.x = sub == 0 ? 0.0f : ma.x;
          gg2 = ga * ga + gg2; gg2 = gb * gb + gg2;
          dd2 = mq * mq + dd2; dd2 = mb * mb + dd2;
          gm2 = ga * ma + gm2; gm2 = gb * mb + gm2;  // g is zero at element 0
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool first = (j == 0 && i == 0 && sub == 0);
            gg += g[j][i] * g[j][i];
            dd += first ? (T)0 : mu[j][i] * mu[j][i];  // |mu|^2 without its first element
            gm += g[j][i] * mu[j][i];                   // g is zero at element 0
          }
        }
      }
      if (VMF_PK_SUMS && FAST && sizeof(T) == 4) {
        gg = (T)(gg2.x + gg2.y); dd = (T)(dd2.x + dd2.y); gm = (T)(gm2.x + gm2.y);
      }
      group_allreduce_sum3<LPR, T>(gg, dd, gm);
      T mu0 = LPR == 1 ? mu[0][0] : __shfl(mu[0][0], lane - sub, 64);
      // raw mean directions: everything below is written for mu / ||mu||; `inv` carries the scale to the elements
      // (the exact-fit streaming instantiation is kept free of this option: a run-time flag there costs 3 % of the
      //  headline kernel; raw means take the guarded 128-bit instantiation instead, see the launch code)
      const T inv = (!FAST && p.normalize) ? F<T>::rsqrt(dd + mu0 * mu0) : (T)1;
      mu0 *= inv;
      gm *= inv;
      const T d0 = (T)1 - mu0;
      const T tail = dd * (inv * inv);  // |mu|^2 without its first element
      dd = d0 * d0 + tail;              // |e1 - mu|^2
      const T sc2 = somw[rr], w = sw[rr];  // sqrt(1 - w^2), w
      const T f = sc2 * F<T>::rsqrt(gg);
      const T zd = w * d0 - f * gm;
      const T coef = dd > (T)1e-30 ? (T)2 * zd * F<T>::rcp(dd) : (T)0;
      const T coef_mu = coef * inv;
      // retire the prefetch of the next rows BEFORE this step's store is issued (see prefetch_issue); also on the last
      // step, which sent for nothing: a wait every path runs through is one a reader of the disassembly (and the test
      // that checks it) can see without knowing that the two tests of `more` agree
      if (PF) {
#pragma unroll
        for (int jj = 0; jj < P; ++jj) prefetch_wait(cur[jj]);
      }
      if (valid) {
        T* zrow = z_t + rr * m;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const int chunk = j * LPR + sub;
          T out[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool first = (j == 0 && i == 0 && sub == 0);
            out[i] = first ? w - coef * d0 : f * g[j][i] + coef_mu * mu[j][i];
          }
          if (FAST) {
            if (sizeof(T) == 4) {
              // z is written once and not read again by this kernel: stream it past the caches
              f32x4 v4 = {(float)out[0], (float)out[1], (float)out[2], (float)out[3]};
              store_nt(v4, reinterpret_cast<float*>(zrow + 4 * chunk));
            } else {
              Vec4<T> v;
              v.v[0] = out[0]; v.v[1] = out[1]; v.v[2] = out[2]; v.v[3] = out[3];
              *reinterpret_cast<Vec4<T>*>(zrow + 4 * chunk) = v;
            }
          } else {
            store4<T, VEC>(zrow, 4 * chunk, m, out);
          }
        }
        if (p.logp && sub == 0) {
          // <mu, z> = <mu, z'> - coef <mu, d> with <mu, d> = mu0 - |mu|^2 = mu0 (1 - mu0) - tail: written through the
          // two small quantities, because coef ~ 1 / |d|^2 multiplies it and mu0 - |mu|^2 formed from rounded values
          // near 1 loses everything when mu is close to the pole (a circle's mu is, for one row in a few hundred)
          const T mz = w * mu0 + f * gm - coef * (mu0 * d0 - tail);
          if (LP_SLAB) {
            slc[rr] += skp[rr] * (float)mz;  // written out coalesced once the tile is done
          } else {
            T sig_;
            p.logp[tile0 + rr] += kappa_of<T>(kap_t[rr * p.kappa_stride], sp, &sig_) * mz;  // log C_m(kappa) came from the pre-pass
          }
        }
      }
    }
    wave_lds_sync();
    if (LP_SLAB && p.logp)
      for (int i = lane; i < nt; i += 64) p.logp[tile0 + i] = (T)slc[i];
    };
    if constexpr (VMF_FULL_TILE_COPY && FAST && sizeof(T) == 4 && P == 1) {
      if (ntile == TILE_ROWS) phase2(std::true_type{});
      else phase2(std::false_type{});
    } else {
      phase2(std::false_type{});
    }
  }
}

// ---- narrow rows (m = 3 .. 31 except 8; fp32, counter noise): ONE lane owns a row --------------------------------------
// The lane-group mapping above gives rows that do not fill a power-of-two group scalar, guarded element accesses and
// leaves lanes idle (m = 10: four lanes per row, ten of their sixteen slots used).  Here a wave's tile of rows is
// contiguous floats of mu: it moves between HBM and the wave as a flat array, 16 bytes per lane and access, fully
// coalesced, and is parked in a wave-private LDS slab; a lane then reads its row from the slab (row stride m words:
// conflict-free 32-bit reads for odd m, 64-bit reads for even m), keeps the whole row in registers -- no cross-lane sums
// at all -- writes z back over it, and the slab leaves as a flat array again.  (m = 4: a 16-byte unit is a row, so
// there is no slab at all.)  The tile's kappa is loaded first and its mu right behind: the memory counter retires loads
// in order, so phase 1 starts when kappa is there and mu lands while it runs.  kappa and log C_m(kappa) of row
// lane + 64 s stay in the registers of the lane that phase 2 gives that row, so the only LDS besides the mu slab is the
// pair of slabs that carries b(kappa) into the rejection phase and (w, 1 - w^2) out of it -- LDS is what bounds the
// number of resident waves here.  Same counters, same noise, same arithmetic as the kernel above.
// rows per wave: the rejection phase ends in a tail whose cost is per tile, so the narrowest rows take larger tiles (as in
// the kernel above).  2^24 rows, forward, lane-group kernel -> this one as first written -> now (settled, ms):
// m = 3: - / 0.189 / 0.130; 4: 0.163 / - / 0.139; 5: - / 0.190 / 0.163; 10: 0.44 / 0.268 / 0.250; 12: 0.51 / 0.307 /
// 0.287; 16: 0.408 / - / 0.370; 20: 0.75 / 0.45 / 0.45; 31: 1.34 / 0.70 / 0.70
#ifndef VMF_RL_TILE_WIDE
#define VMF_RL_TILE_WIDE 128
#endif
#ifndef VMF_RL_MAX
#define VMF_RL_MAX 31
#endif
#ifndef VMF_RL_TILE_WIDER
#define VMF_RL_TILE_WIDER 64
#endif
#ifndef VMF_RL_KF_MAX
#define VMF_RL_KF_MAX 16  // widths up to this one load kappa ahead of mu
#endif
// waves per workgroup of the forward kernel.  It has no workgroup barrier, so any count works; one-wave workgroups free
// their LDS and registers as soon as that wave's stores are out instead of when the slowest of four is done (the
// rejection phase makes tiles finish at different times): 1-2 % at m <= 16, nothing or slightly worse beyond
#ifndef VMF_RL_WAVES_NARROW
#define VMF_RL_WAVES_NARROW 1
#endif
constexpr int rl_waves(int m) { return m <= 16 ? VMF_RL_WAVES_NARROW : 4; }
#ifndef VMF_RL_TILE4
#define VMF_RL_TILE4 384
#endif
constexpr int rl_tile(int m) { return m == 4 ? VMF_RL_TILE4 : m <= 4 ? 384 : (m <= 7 ? 256 : (m <= 16 ? VMF_RL_TILE_WIDE : VMF_RL_TILE_WIDER)); }
// Forward tiles: the rejection phase runs over the whole wave tile (its tail costs a few wave-iterations per TILE, so
// the tile wants to be large), the mu slab holds one sub-tile at a time (it is what bounds the resident waves, so it
// wants to be small): rl_fwd_tile rows per wave, rl_fwd_sub rows per slab.  Settled times for 2^24 rows, one slab of the
// whole tile (384 rows at m = 3, 256 at 5, 128 from 9 on) -> sub-tiled: m = 3 0.135 -> 0.130 ms (512 / 128: the same),
// 5 0.174 -> 0.163, 10 0.254 -> 0.250 (256 / 64; 384 rows: 0.28); 12 0.287 -> 0.295, 16 0.370 -> 0.397: from m = 12 on
// the whole-tile slab of 128 rows stays.
#ifndef VMF_RLF_TILE_S
#define VMF_RLF_TILE_S 384  // m = 3, 5, 6, 7
#endif
#ifndef VMF_RLF_SUB_S
#define VMF_RLF_SUB_S 128
#endif
#ifndef VMF_RLF_TILE_M
#define VMF_RLF_TILE_M 256  // m = 9, 10, 11
#endif
#ifndef VMF_RLF_SUB_M
#define VMF_RLF_SUB_M 64
#endif
constexpr int rl_fwd_tile(int m) {
  return m == 4 ? VMF_RL_TILE4 : (m <= 7 ? VMF_RLF_TILE_S : (m <= 11 ? VMF_RLF_TILE_M : (m <= 16 ? VMF_RL_TILE_WIDE : VMF_RL_TILE_WIDER)));
}
constexpr int rl_fwd_sub(int m) {
  return m == 4 ? VMF_RL_TILE4 : (m <= 7 ? VMF_RLF_SUB_S : (m <= 11 ? VMF_RLF_SUB_M : (m <= 16 ? VMF_RL_TILE_WIDE : VMF_RL_TILE_WIDER)));
}

template <int M, bool FULL>
__device__ __forceinline__ void rowlane_tile(const FwdParams<float>& p, const PhiloxKey& key, float* const somw,
                                             float* const sw, float* const srow, const int64_t tile0, const int ntile_,
                                             const int lane) {
  using T = float;
  constexpr int RL_TILE = rl_fwd_tile(M), SUB = rl_fwd_sub(M), NSUB = RL_TILE / SUB, SPL = SUB / 64;
  static_assert(RL_TILE % SUB == 0 && SUB % 64 == 0, "a wave tile is a whole number of slabs of whole 64-row steps");
  const int ntile = FULL ? RL_TILE : ntile_;  // a compile-time constant in the full-tile copy
  constexpr int UNITS = SUB * M / 4;      // 16-byte units of a full sub-tile
  constexpr int UPL = (UNITS + 63) / 64;  // per lane
  constexpr int CHUNKS = (M + 3) / 4;
  // m = 4: a 16-byte unit IS a row and unit lane + 64 s is the row phase 2 gives this lane, so the rows stay in
  // registers and no slab exists
  constexpr bool DIRECT = M == 4;
  static_assert(!DIRECT || NSUB == 1, "the slab-free shape keeps its whole tile in registers");
  const T* const kap_t = p.kappa + tile0 * p.kappa_stride;
  const uint64_t grow0 = p.row0 + (uint64_t)tile0;
  const T* const mu_t = p.mu + tile0 * p.mu_stride;
  T* const z_t = p.z + tile0 * M;
  // The tile's concentrations first, then the first sub-tile of mu: the memory counter retires loads in order, so a wait
  // for kappa also waits for every load issued before it -- with kappa in front, phase 1 starts as soon as kappa is
  // there and mu lands while it runs.
  constexpr int KPL = RL_TILE / 64;
  constexpr bool KAPPA_FIRST = M <= VMF_RL_KF_MAX;
  T kraw[KPL];
  if (KAPPA_FIRST) {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const int i = lane + 64 * s;
      kraw[s] = (FULL || i < ntile) ? __builtin_nontemporal_load(kap_t + (int64_t)i * p.kappa_stride) : (T)1;
    }
  }
  u32x4 stage[UPL];
  if (FULL) {
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const int q = lane + 64 * s;
      if (UNITS % 64 == 0 || q < UNITS) stage[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(mu_t) + q);
    }
  }
  // ---- phase 1: rejection sampler, one row per lane with ballot refill (as in vmf_rsample_fwd_kernel) ----
  const bool sp = p.softplus != 0;
  const bool lp = p.logp != nullptr;
  // kappa and log C_m(kappa) of row lane + 64 s stay in this lane's registers: phase 2 gives the lane that same row
  T kap_r[KPL], lc_r[KPL];
#pragma unroll
  for (int s = 0; s < KPL; ++s) {
    const int i = lane + 64 * s;
    kap_r[s] = (T)1;
    lc_r[s] = (T)0;
    if (FULL || i < ntile) {
      T sig_;
      const T kap = kappa_of<T>(KAPPA_FIRST ? kraw[s] : kap_t[(int64_t)i * p.kappa_stride], sp, &sig_);
      if (p.kappa_out) p.kappa_out[tile0 + i] = kap;
      kap_r[s] = kap;
      if (lp) lc_r[s] = vmf_log_norm_f32v(p.lc_v, p.lc_lg, p.lc_c0, kap);
      somw[i] = wood_b<T>(kap, M);
    }
  }
  wave_lds_sync();
#ifdef VMF_ABLATE_PHASE1  // timing ablation only (w fixed)
  for (int i = lane; i < ntile; i += 64) { sw[i] = 0.5f; somw[i] = 0.75f; }
#else
  {
    int my = lane, next = 64, k = 0;
    bool active = my < ntile;
    T b = active ? somw[my] : (T)0;
    while (__ballot(active) != 0ull) {
      bool acc = false;
      if (active) {
        T w, omw, c2;
        const U4 r = philox4x32_10(grow0 + (uint64_t)my, (uint32_t)k, key);
        acc = wood_try_counter<T>(b, M, u01<T>(r.x), u01<T>(r.y), u01<T>(r.z), &w, &omw, &c2);
        if (k + 1 >= p.max_prop) acc = true;  // the last proposal allowed is taken regardless
        if (acc) {
          sw[my] = w;
          somw[my] = c2;
          if (p.w_out) p.w_out[tile0 + my] = w;
          if (p.omw_out) p.omw_out[tile0 + my] = omw;
          if (p.accept_idx) p.accept_idx[tile0 + my] = k;
        }
        ++k;
      }
      const unsigned long long accm = __ballot(acc);
      if (acc) {
        my = next + __popcll(accm & ((1ull << lane) - 1ull));
        k = 0;
        active = my < ntile;
        if (active) b = somw[my];
      }
      next += __popcll(accm);
    }
  }
#endif
  wave_lds_sync();
  // ---- phase 2, one sub-tile at a time: park its mu in the slab, send for the next sub-tile's mu (into the registers
  //      just emptied: it lands while this one computes), one row per lane, the slab out as a flat array ----
#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub) {
    const int rows0 = sub * SUB;
    const int left = ntile - rows0;
    const int nsub = FULL ? SUB : (left < SUB ? left : SUB);
    if (!FULL && nsub <= 0) break;  // wave-uniform
    const T* const mu_s = mu_t + (int64_t)rows0 * p.mu_stride;
    T* const z_s = z_t + rows0 * M;
    if (!DIRECT) {
      if (FULL) {
#pragma unroll
        for (int s = 0; s < UPL; ++s) {
          const int q = lane + 64 * s;
          if (UNITS % 64 == 0 || q < UNITS) reinterpret_cast<u32x4*>(srow)[q] = stage[s];
        }
      } else {
        for (int e = lane; e < nsub * M; e += 64) srow[e] = mu_s[(e / M) * p.mu_stride + e % M];
      }
      wave_lds_sync();
      if (FULL && sub + 1 < NSUB) {
#pragma unroll
        for (int s = 0; s < UPL; ++s) {
          const int q = lane + 64 * s;
          if (UNITS % 64 == 0 || q < UNITS)
            stage[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(mu_s + SUB * M) + q);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const int rl = j * 64 + lane;  // row inside the sub-tile
      const int r = rows0 + rl;      // row inside the tile
      if (FULL || rl < nsub) {
        T mu[M], g[M];
        if (DIRECT) {
          if (FULL) {
            f32x4 v;
            __builtin_memcpy(&v, &stage[j % UPL], 16);
            mu[0] = v.x; mu[1] = v.y; mu[2 % M] = v.z; mu[3 % M] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < M; ++e) mu[e] = mu_s[(int64_t)rl * p.mu_stride + e];
          }
        } else if (M % 2 == 0) {
#pragma unroll
          for (int e = 0; e < M; e += 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(srow + rl * M + e);
            mu[e] = v.x; mu[e + 1] = v.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < M; ++e) mu[e] = srow[rl * M + e];
        }
        // normals of elements 4c .. 4c+3 from Philox(row, SLOT_GAUSS + c), element 0 carries none
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
          const U4 rr = philox4x32_10(grow0 + (uint64_t)r, SLOT_GAUSS + (uint32_t)c, key);
          T n0, n1;
          box_muller<T>(u01<T>(rr.x), u01<T>(rr.y), &n0, &n1);
          if (4 * c < M) g[4 * c] = n0;
          if (4 * c + 1 < M) g[4 * c + 1] = n1;
          if (4 * c + 2 < M) {
            box_muller<T>(u01<T>(rr.z), u01<T>(rr.w), &n0, &n1);
            g[4 * c + 2] = n0;
            if (4 * c + 3 < M) g[4 * c + 3] = n1;
          }
        }
        g[0] = (T)0;
        T gg = 0, tail = 0, gm = 0;
#pragma unroll
        for (int e = 1; e < M; ++e) {
          gg += g[e] * g[e];
          tail += mu[e] * mu[e];
          gm += g[e] * mu[e];
        }
        T mu0 = mu[0];
        const T inv = p.normalize ? F<T>::rsqrt(tail + mu0 * mu0) : (T)1;
        mu0 *= inv;
        gm *= inv;
        const T d0 = (T)1 - mu0;
        tail *= inv * inv;                // |mu|^2 without its first element
        const T dd = d0 * d0 + tail;      // |e1 - mu|^2
        const T w = sw[r];
        T c2 = somw[r];
        c2 = c2 < (T)1e-10 ? (T)1e-10 : c2;
        const T f = F<T>::sqrt(c2) * F<T>::rsqrt(gg);
        const T zd = w * d0 - f * gm;
        const T coef = dd > (T)1e-30 ? (T)2 * zd * F<T>::rcp(dd) : (T)0;
        const T coef_mu = coef * inv;
        T z[M];
        z[0] = w - coef * d0;
#pragma unroll
        for (int e = 1; e < M; ++e) z[e] = f * g[e] + coef_mu * mu[e];
        if (DIRECT) {
          if (FULL) {
            const f32x4 v = {z[0], z[1], z[2 % M], z[3 % M]};
            store_nt(v, z_s + 4 * rl);
          } else {
#pragma unroll
            for (int e = 0; e < M; ++e) z_s[rl * M + e] = z[e];
          }
        } else if (M % 2 == 0) {
#pragma unroll
          for (int e = 0; e < M; e += 2) {
            f32x2 v;
            v.x = z[e]; v.y = z[e + 1];
            *reinterpret_cast<f32x2*>(srow + rl * M + e) = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < M; ++e) srow[rl * M + e] = z[e];
        }
        if (lp) {  // <mu, z> through the two small quantities, as in the kernel above
          const T mz = w * mu0 + f * gm - coef * (mu0 * d0 - tail);
          p.logp[tile0 + r] = lc_r[sub * SPL + j] + kap_r[sub * SPL + j] * mz;
        }
      }
    }
    if (DIRECT) continue;
    wave_lds_sync();
    if (FULL) {
#pragma unroll
      for (int s = 0; s < UPL; ++s) {
        const int q = lane + 64 * s;
        if (UNITS % 64 == 0 || q < UNITS) {
          f32x4 v;
          const u32x4 raw = reinterpret_cast<const u32x4*>(srow)[q];
          __builtin_memcpy(&v, &raw, 16);
          store_nt(v, z_s + 4 * q);
        }
      }
    } else {
      for (int e = lane; e < nsub * M; e += 64) z_s[e] = srow[e];
    }
    wave_lds_sync();  // the slab is rewritten by the next sub-tile
  }
}

// (asking for 8 waves per SIMD -- 64 registers -- spills at m = 10 and measures 2 % slower than the 66 it takes)
template <int M>
__global__ __launch_bounds__(rl_waves(M) * 64) void vmf_rsample_rowlane_kernel(FwdParams<float> p) {
  using T = float;
  constexpr int RL_TILE = rl_fwd_tile(M);
  constexpr int RL_WAVES = rl_waves(M);
  __shared__ T s_wc[RL_WAVES][2][RL_TILE];
  __shared__ __align__(16) T s_rows[RL_WAVES][M == 4 ? 4 : rl_fwd_sub(M) * M];  // (m = 4 keeps its rows in registers)
  const PhiloxKey key = philox_resolve(p.ka);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile0 = ((int64_t)blockIdx.x * RL_WAVES + wave) * RL_TILE;  // wave-uniform
  const int64_t left = p.n - tile0;
  if (left <= 0) return;  // the kernel has no workgroup barrier
  const int ntile = left < RL_TILE ? (int)left : RL_TILE;
  // flat 128-bit moves for full tiles of dense, aligned rows; any other layout (views that start mid-allocation, one
  // shared mean, padded rows) is gathered element by element -- same arithmetic, same bits.  Two copies of the body, so
  // that the full-tile one issues its loads unconditionally and the compiler can count them (see rowlane_tile).
  if (p.dense16 && ntile == RL_TILE)
    rowlane_tile<M, true>(p, key, s_wc[wave][0], s_wc[wave][1], s_rows[wave], tile0, ntile, lane);
  else
    rowlane_tile<M, false>(p, key, s_wc[wave][0], s_wc[wave][1], s_rows[wave], tile0, ntile, lane);
}

// widths that take the one-lane-per-row forward kernel: everything up to VMF_RL_MAX except 8, whose rows fill their pair
// of lanes exactly and measure 3 % faster in the lane-group kernel (2^24 rows, lane-group -> here: m = 4 0.163 -> 0.139 ms,
// 8 0.215 -> 0.222, 16 0.408 -> 0.369).  The backward kernel keeps the lane groups for 4, 8 and 16.
static bool rowlane_ok(int m) { return m >= 3 && m <= VMF_RL_MAX && m != 8; }
static bool rowlane_bwd_ok(int m) { return rowlane_ok(m) && m != 4 && m != 16; }
static void launch_rowlane(const FwdParams<float>& p, hipStream_t s) {
  const int64_t tiles = (p.n + rl_fwd_tile(p.m) - 1) / rl_fwd_tile(p.m);
  const int waves = rl_waves(p.m);
  const dim3 grid((unsigned)((tiles + waves - 1) / waves)), block(waves * 64);
  switch (p.m) {
#define VMF_RL(MM) case MM: hipLaunchKernelGGL((vmf_rsample_rowlane_kernel<MM>), grid, block, 0, s, p); break
    VMF_RL(4); VMF_RL(8); VMF_RL(16);
    VMF_RL(3); VMF_RL(5); VMF_RL(6); VMF_RL(7); VMF_RL(9); VMF_RL(10); VMF_RL(11); VMF_RL(12); VMF_RL(13); VMF_RL(14);
    VMF_RL(15); VMF_RL(17); VMF_RL(18); VMF_RL(19); VMF_RL(20); VMF_RL(21); VMF_RL(22); VMF_RL(23); VMF_RL(24); VMF_RL(25);
    VMF_RL(26); VMF_RL(27); VMF_RL(28); VMF_RL(29); VMF_RL(30); VMF_RL(31);
#undef VMF_RL
    default: break;
  }
}

// ---- the circle (m = 2), dense 16-byte-aligned fp32 rows, counter noise: TWO rows per lane ---------------------------
// A circle row is 8 bytes of mu in and 8 bytes of z out against a rejection sampler, so the kernel is bound by the
// instructions it issues, not by HBM, unless they are cut to the bone.  What this kernel does about it:
//   * a lane owns the rows 2p, 2p+1 of a 128-row step: one 128-bit access moves both rows of mu / z, one 64-bit access
//     both concentrations, and ONE Philox call gives both rows their first proposal (see circle_try in vmf_math.h);
//   * a wave issues every load of its tile (CF_TILE rows) before it computes anything: the kappa loads first (needed
//     first; the memory counter retires in order), then mu, which lands while the rejection phase runs -- phase 2 never
//     waits on memory;
//   * phase 1a evaluates the first proposal of every row, parks (w, signed sqrt(1-w^2)) of the accepted rows in a
//     wave-private LDS slab and compacts the rejected rows (ballot + prefix popcount) into a wave-private queue;
//   * phase 1b works the queue off one row per lane, two proposals per Philox call, retiring accepted lanes and
//     refilling them from the queue through a ballot as the general-m kernel does: its trip count follows the mean
//     number of further proposals of the REJECTED rows only (a fifth of the tile), about one wave-iteration per 100 rows;
//   * phase 2 reflects both rows (circle_compose) and streams z out with one 128-bit non-temporal store per lane.
// Same draws, same bits as the one-row-per-lane instantiations of vmf_rsample_fwd_kernel (MODE 3 / MODE 0), which take
// every other layout: shards that start on an odd global row, views that are not 16-byte aligned, shared parameters.
#ifndef VMF_CF_TILE
#define VMF_CF_TILE 512
#endif
constexpr int CF_TILE = VMF_CF_TILE;
constexpr int CF_STEPS = CF_TILE / 128;
#ifndef VMF_CF_WAVES_PER_SIMD
#define VMF_CF_WAVES_PER_SIMD 8
#endif
#ifndef VMF_CF_XCD
#define VMF_CF_XCD 1
#endif
// tiles per wave; beyond one, the next tile's loads are issued before the current tile is computed
#ifndef VMF_CF_PIPE
#define VMF_CF_PIPE 1
#endif
constexpr int CF_PIPE = VMF_CF_PIPE;
#ifndef VMF_CF_NT
#define VMF_CF_NT 1
#endif
__device__ __forceinline__ int lanes_below(unsigned long long mask) {  // set bits of `mask` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// OPTS: any of softplus / normalize / kappa_out / w_out / omw_out / accept_idx / logp is in use.  FULL: the tile has all
// its CF_TILE rows (the ragged last tile runs the same code with guarded accesses; a separate instantiation, so that the
// full tiles' loads and their waits are straight-line code the compiler can count).
struct CircleRegs {
  f32x2 kap[CF_STEPS];
  f32x4 mu[CF_STEPS];
};
// every load of a tile
template <bool FULL>
__device__ __forceinline__ void circle_load(CircleRegs& r, const FwdParams<float>& p, const int64_t tile0,
                                            const int ntile, const int lane) {
  constexpr bool full = FULL;
  const float* const kap_t = p.kappa + tile0;
  const float* const mu_t = p.mu + 2 * tile0;
  f32x2(&kap)[CF_STEPS] = r.kap;
  f32x4(&mu)[CF_STEPS] = r.mu;
  if (full) {
#pragma unroll
    for (int s = 0; s < CF_STEPS; ++s)
      kap[s] = VMF_CF_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(kap_t) + 64 * s + lane)
                         : reinterpret_cast<const f32x2*>(kap_t)[64 * s + lane];
#pragma unroll
    for (int s = 0; s < CF_STEPS; ++s)
      mu[s] = VMF_CF_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mu_t) + 64 * s + lane)
                        : reinterpret_cast<const f32x4*>(mu_t)[64 * s + lane];
  } else {
#pragma unroll
    for (int s = 0; s < CF_STEPS; ++s) {
      const int rA = 128 * s + 2 * lane;
      kap[s].x = rA < ntile ? kap_t[rA] : 1.0f;
      kap[s].y = rA + 1 < ntile ? kap_t[rA + 1] : 1.0f;
      f32x2 a = {1.0f, 0.0f}, b = {1.0f, 0.0f};
      if (rA < ntile) a = *reinterpret_cast<const f32x2*>(mu_t + 2 * rA);
      if (rA + 1 < ntile) b = *reinterpret_cast<const f32x2*>(mu_t + 2 * rA + 2);
      mu[s] = f32x4{a.x, a.y, b.x, b.y};
    }
  }
}
// PF: the wave has another full tile after this one, at rows next0 ...: its kappa loads are issued once phase 1a has
// consumed this tile's (into the same registers), its mu loads step by step as phase 2 releases this tile's -- so a wave
// always has loads in flight while it computes, at no cost in registers
template <bool OPTS, bool FULL, bool PF>
__device__ __forceinline__ void circle_compute(CircleRegs& r, const FwdParams<float>& p, const PhiloxKey key,
                                               float* const park, uint16_t* const queue, const int64_t tile0,
                                               const int ntile, const int lane, const int64_t next0 = 0) {
  static_assert(!PF || (FULL && !OPTS), "prefetch: full tiles of the plain instantiation only");
  constexpr bool full = FULL;
  float* const z_t = p.z + 2 * tile0;
  const uint64_t grow0 = p.row0 + (uint64_t)tile0;  // even: the launch code sends odd row offsets to the other kernel
  const bool normalize = OPTS && p.normalize != 0;
  f32x2(&kap)[CF_STEPS] = r.kap;
  f32x4(&mu)[CF_STEPS] = r.mu;

  // ---- phase 1a: the first proposal of every row ----
  int qn = 0;  // rows waiting for further proposals (wave-uniform)
  const bool forced0 = p.max_prop <= 1;  // the last proposal allowed is taken regardless
#pragma unroll
  for (int s = 0; s < CF_STEPS; ++s) {
    const int rA = 128 * s + 2 * lane;
    float kA = kap[s].x, kB = kap[s].y;
    if (OPTS) {
      if (p.softplus) {
        float sig_;
        kA = kappa_of<float>(kA, true, &sig_);
        kB = kappa_of<float>(kB, true, &sig_);
        kap[s] = f32x2{kA, kB};
      }
      if (p.kappa_out) {
        if (rA < ntile) p.kappa_out[tile0 + rA] = kA;
        if (rA + 1 < ntile) p.kappa_out[tile0 + rA + 1] = kB;
      }
    }
    const U4 r = philox4x32_10((grow0 >> 1) + (uint64_t)(64 * s + lane), 0u, key);
    const float bA = circle_b<float>(kA), bB = circle_b<float>(kB);
    float pcA, psA, pcB, psB;
    const bool accA = circle_try<float>(bA, circle_hb<float>(bA), r.x, r.y, &pcA, &psA) || forced0;
    const bool accB = circle_try<float>(bB, circle_hb<float>(bB), r.z, r.w, &pcB, &psB) || forced0;
    const bool negA = (r.x & 1u) != 0u, negB = (r.z & 1u) != 0u;
    float wA, omwA, fsA, wB, omwB, fsB;
    circle_finish<float>(bA, pcA, psA, negA, &wA, &omwA, &fsA);
    circle_finish<float>(bB, pcB, psB, negB, &wB, &omwB, &fsB);
    // accepted: (w, signed sqrt(1-w^2)); rejected: (b, the sign) for phase 1b
    const f32x4 pk = {accA ? wA : bA, accA ? fsA : (negA ? -1.0f : 1.0f), accB ? wB : bB,
                      accB ? fsB : (negB ? -1.0f : 1.0f)};
    *reinterpret_cast<f32x4*>(park + 2 * rA) = pk;
    const bool rejA = !accA && rA < ntile, rejB = !accB && rA + 1 < ntile;
    const unsigned long long mA = __ballot(rejA), mB = __ballot(rejB);
    if (rejA) queue[qn + lanes_below(mA)] = (uint16_t)rA;
    qn += __popcll(mA);
    if (rejB) queue[qn + lanes_below(mB)] = (uint16_t)(rA + 1);
    qn += __popcll(mB);
    if (OPTS) {
      if (accA && rA < ntile) {
        if (p.w_out) p.w_out[tile0 + rA] = wA;
        if (p.omw_out) p.omw_out[tile0 + rA] = omwA;
        if (p.accept_idx) p.accept_idx[tile0 + rA] = 0;
      }
      if (accB && rA + 1 < ntile) {
        if (p.w_out) p.w_out[tile0 + rA + 1] = wB;
        if (p.omw_out) p.omw_out[tile0 + rA + 1] = omwB;
        if (p.accept_idx) p.accept_idx[tile0 + rA + 1] = 0;
      }
    }
  }
  if (PF) {
#pragma unroll
    for (int s = 0; s < CF_STEPS; ++s)
      kap[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p.kappa + next0) + 64 * s + lane);
  }
  wave_lds_sync();

  // ---- phase 1b: further proposals of the rejected rows, one row per lane, ballot refill from the queue ----
  {
    int my = lane, next = 64, t = 0, row = 0;
    bool active = my < qn;
    float b = 0.0f, hb = 0.0f, sg = 1.0f;
    if (active) {
      row = queue[my];
      const f32x2 e = *reinterpret_cast<const f32x2*>(park + 2 * row);
      b = e.x; sg = e.y; hb = circle_hb<float>(b);
    }
    while (__ballot(active) != 0ull) {
      bool acc = false;
      if (active) {
        const U4 rc = philox4x32_10(grow0 + (uint64_t)row, 1u + (uint32_t)t, key);
        int k = 1 + 2 * t;
        float pc, ps, pc1, ps1;
        const bool a0 = circle_try<float>(b, hb, rc.x, rc.y, &pc, &ps);
        const bool a1 = circle_try<float>(b, hb, rc.z, rc.w, &pc1, &ps1);
        const bool use0 = a0 || k + 1 >= p.max_prop;
        acc = use0 || a1 || k + 2 >= p.max_prop;
        if (!use0) { pc = pc1; ps = ps1; ++k; }
        if (acc) {
          float w, omw, fs;
          circle_finish<float>(b, pc, ps, sg < 0.0f, &w, &omw, &fs);
          *reinterpret_cast<f32x2*>(park + 2 * row) = f32x2{w, fs};
          if (OPTS) {
            if (p.w_out) p.w_out[tile0 + row] = w;
            if (p.omw_out) p.omw_out[tile0 + row] = omw;
            if (p.accept_idx) p.accept_idx[tile0 + row] = k;
          }
        }
        ++t;
      }
      const unsigned long long accm = __ballot(acc);
      if (acc) {
        my = next + lanes_below(accm);
        t = 0;
        active = my < qn;
        if (active) {
          row = queue[my];
          const f32x2 e = *reinterpret_cast<const f32x2*>(park + 2 * row);
          b = e.x; sg = e.y; hb = circle_hb<float>(b);
        }
      }
      next += __popcll(accm);
    }
  }
  wave_lds_sync();

  // ---- phase 2: reflect and store ----
#pragma unroll
  for (int s = 0; s < CF_STEPS; ++s) {
    const int rA = 128 * s + 2 * lane;
    const f32x4 pk = *reinterpret_cast<const f32x4*>(park + 2 * rA);
    float zA0, zA1, mzA, zB0, zB1, mzB;
    circle_compose<float>(mu[s].x, mu[s].y, pk.x, pk.y, normalize, &zA0, &zA1, &mzA);
    circle_compose<float>(mu[s].z, mu[s].w, pk.z, pk.w, normalize, &zB0, &zB1, &mzB);
    if (full) {
      store_nt(f32x4{zA0, zA1, zB0, zB1}, z_t + 4 * (64 * s + lane));
      if (PF) mu[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.mu + 2 * next0) + 64 * s + lane);
    } else {
      if (rA < ntile) *reinterpret_cast<f32x2*>(z_t + 2 * rA) = f32x2{zA0, zA1};
      if (rA + 1 < ntile) *reinterpret_cast<f32x2*>(z_t + 2 * rA + 2) = f32x2{zB0, zB1};
    }
    if (OPTS && p.logp) {
      const float kA = kap[s].x, kB = kap[s].y;
      if (rA < ntile) p.logp[tile0 + rA] = vfma(kA, mzA, vmf_log_norm_circle_f32(kA));
      if (rA + 1 < ntile) p.logp[tile0 + rA + 1] = vfma(kB, mzB, vmf_log_norm_circle_f32(kB));
    }
  }
  if (CF_PIPE > 1) wave_lds_sync();  // the next tile of this wave reuses the slab and the queue
}
template <bool OPTS>
__global__ __launch_bounds__(WAVES * 64, VMF_CF_WAVES_PER_SIMD) void vmf_circle_fwd_kernel(FwdParams<float> p) {
  __shared__ __align__(16) float s_park[WAVES][2 * CF_TILE];
  __shared__ uint16_t s_queue[WAVES][CF_TILE];
  const PhiloxKey key = philox_resolve(p.ka);
  const int lane = threadIdx.x & 63;
  // the wave index as a scalar: tile bases then live in scalar registers and every access is base + lane offset
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if VMF_CF_XCD
  const uint32_t bid = xcd_block(blockIdx.x, gridDim.x);
#else
  const uint32_t bid = blockIdx.x;
#endif
  constexpr int K = CF_PIPE;
  const int64_t tile0 = ((int64_t)bid * WAVES + wave) * (K * CF_TILE);  // wave-uniform
  const int64_t left = p.n - tile0;
  if (left <= 0) return;  // the kernel has no workgroup barrier
  float* const park = s_park[wave];
  uint16_t* const queue = s_queue[wave];
  const int nfull = left >= (int64_t)K * CF_TILE ? K : (int)(left / CF_TILE);  // full tiles of this wave
  CircleRegs r;
  if (K > 1 && !OPTS && nfull == K) {
    // K full tiles as straight-line code: each tile's loads are issued while the tile before it computes
    circle_load<true>(r, p, tile0, CF_TILE, lane);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      if (i + 1 < K)
        circle_compute<false, true, true>(r, p, key, park, queue, tile0 + (int64_t)i * CF_TILE, CF_TILE, lane,
                                          tile0 + (int64_t)(i + 1) * CF_TILE);
      else
        circle_compute<false, true, false>(r, p, key, park, queue, tile0 + (int64_t)i * CF_TILE, CF_TILE, lane);
    }
    return;
  } else {
#pragma unroll 1
    for (int i = 0; i < nfull; ++i) {
      circle_load<true>(r, p, tile0 + (int64_t)i * CF_TILE, CF_TILE, lane);
      circle_compute<OPTS, true, false>(r, p, key, park, queue, tile0 + (int64_t)i * CF_TILE, CF_TILE, lane);
    }
  }
  const int64_t rem = left - (int64_t)nfull * CF_TILE;
  if (nfull < K && rem > 0) {  // the ragged last tile
    circle_load<false>(r, p, tile0 + (int64_t)nfull * CF_TILE, (int)rem, lane);
    circle_compute<OPTS, false, false>(r, p, key, park, queue, tile0 + (int64_t)nfull * CF_TILE, (int)rem, lane);
  }
}

// Backward: one pass over (grad_z, mu) rows; normals regenerated from the counters unless given.
template <typename T, int LPR, int P, bool VEC>
__global__ __launch_bounds__(WAVES * 64) void vmf_rsample_bwd_kernel(GenParams<T> p) {
  const PhiloxKey key = philox_resolve(p.ka);
  const int lane = threadIdx.x & 63;
  constexpr int RPI = 64 / LPR;
  const int sub = lane % LPR, grp = lane / LPR;
  const int m = p.m;
  const int64_t n_steps = (p.n + RPI - 1) / RPI;
  const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // 1 to WAVES waves per block
  // a wave takes STEPS consecutive steps; the grid covers all of them (neighbouring waves, neighbouring memory)
  constexpr int STEPS = bwd_steps(LPR, P);
  const int64_t step_end = (wave_id + 1) * STEPS < n_steps ? (wave_id + 1) * STEPS : n_steps;
  for (int64_t step = wave_id * STEPS; step < step_end; ++step) {  // wave-uniform
    const int64_t r = step * RPI + grp;
    const bool valid = r < p.n;
    const int64_t row = valid ? r : 0;
    const T* mrow = p.mu + row * p.mu_stride;
    const T* grow_ = p.gz + row * (int64_t)m;
    if (LPR == 1 && m == 2 && p.gauss == nullptr) {
      // the circle, one row per lane: the tangent is the sign bit of the row's first call; the arithmetic is the function
      // vmf_circle_bwd_kernel uses
      if (valid) {
        const uint64_t grow = p.row0 + (uint64_t)row;
        const U4 r0 = philox4x32_10(grow >> 1, 0u, key);
        const bool neg = ((((grow & 1u) != 0u) ? r0.z : r0.x) & 1u) != 0u;
        T ksig, o0, o1, dk;
        const T kap = kappa_of<T>(p.kappa[row * p.kappa_stride], p.softplus != 0, &ksig);
        const T w = p.w_in[row];
        const T omw = p.omw_in ? p.omw_in[row] : (T)1 - w;
        circle_bwd_row<T>(mrow[0], mrow[1], grow_[0], grow_[1], neg, w, omw, kap, p.normalize != 0, &o0, &o1, &dk);
        p.dmu[row * 2] = o0;
        p.dmu[row * 2 + 1] = o1;
        if (p.gkappa) dk += p.gkappa[row];
        p.dkappa[row] = dk * ksig;
      }
      continue;
    }
    T mu[P][4], g[P][4], G[P][4];
    T gg = 0, dd = 0, gm = 0, Gd = 0, Gg = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int chunk = j * LPR + sub;
      load4<T, VEC, VMF_BWD_STREAM>(mrow, 4 * chunk, m, mu[j]);
      load4<T, VEC, VMF_BWD_STREAM>(grow_, 4 * chunk, m, G[j]);
      noise4<T, VEC>(p.gauss ? p.gauss + row * (int64_t)m : nullptr, p.row0 + (uint64_t)row, chunk, m, key, g[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool first = (chunk == 0 && i == 0);
        gg += g[j][i] * g[j][i];
        dd += first ? (T)0 : mu[j][i] * mu[j][i];  // |mu|^2 without its first element
        gm += g[j][i] * mu[j][i];
        Gd += first ? (T)0 : G[j][i] * mu[j][i];   // <G, mu> without the first element
        Gg += G[j][i] * g[j][i];
      }
    }
    group_allreduce_sum5<LPR, T>(gg, dd, gm, Gd, Gg);
    T mu0 = __shfl(mu[0][0], lane - sub, 64);
    const T G0 = __shfl(G[0][0], lane - sub, 64);
    // raw mean directions: work with mu_hat = mu / ||mu|| (inv = 1 / ||mu||), chain through the normalisation below
    const T inv = p.normalize ? F<T>::rsqrt(dd + mu0 * mu0) : (T)1;
    mu0 *= inv;
    gm *= inv;
    Gd = G0 * ((T)1 - mu0) - Gd * inv;               // <G, e1 - mu_hat>
    dd = ((T)1 - mu0) * ((T)1 - mu0) + dd * (inv * inv);
    const T w = p.w_in[row];
    const T omw = p.omw_in ? p.omw_in[row] : (T)1 - w;
    T ksig;
    const T kap = kappa_of<T>(p.kappa[row * p.kappa_stride], p.softplus != 0, &ksig);
    T c2 = omw * ((T)1 + w);
    const bool inside = c2 >= (T)1e-10;
    c2 = inside ? c2 : (T)1e-10;
    // F<T>: the 1-ulp hardware sqrt / reciprocal for fp32 (as in the forward), the accurate library for fp64.
    // With IEEE divisions and square roots the fp32 backward measured 2.70 ms instead of 2.50 (2^24 x 64).
    const T sq = F<T>::sqrt(c2), gn = F<T>::sqrt(gg);
    const T rgn = F<T>::rcp(gn);
    const T f = sq * rgn;
    const T d0 = (T)1 - mu0;
    const T zd = w * d0 - f * gm;
    const bool refl = dd > (T)1e-30;
    const T inv_dd = refl ? F<T>::rcp(dd) : (T)0;
    // gradient rows are formed by every lane (rows past the end recompute row 0 and are not stored), so that the
    // cross-lane sum below runs with the whole wave active
    T out[P][4];
    T dot = 0;  // <mu_hat, d mu_hat>
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int chunk = j * LPR + sub;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool first = (chunk == 0 && i == 0);
        const T zp = first ? w : f * g[j][i];
        const T mh = mu[j][i] * inv;
        const T de = first ? d0 : -mh;
        out[j][i] = (T)2 * (zp * Gd + zd * G[j][i] - (T)2 * zd * Gd * de * inv_dd) * inv_dd;
        dot += out[j][i] * mh;
      }
    }
    if (p.normalize) {  // wave-uniform: d mu = (d mu_hat - mu_hat <mu_hat, d mu_hat>) / ||mu||
      dot = group_allreduce_sum<LPR, T>(dot);
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[j][i] = (out[j][i] - mu[j][i] * inv * dot) * inv;
    }
    if (valid) {
      T* drow = p.dmu + row * (int64_t)m;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        // d mu is written once and not read again here: stream it past the caches (fp32 128-bit path)
        if (VEC && sizeof(T) == 4) {
          if (4 * (j * LPR + sub) < m) {
            f32x4 v4 = {(float)out[j][0], (float)out[j][1], (float)out[j][2], (float)out[j][3]};
            store_nt(v4, reinterpret_cast<float*>(drow + 4 * (j * LPR + sub)));
          }
        } else {
          store4<T, VEC>(drow, 4 * (j * LPR + sub), m, out[j]);
        }
      }
      if (sub == 0) {
        const T Gp0 = G0 - (T)2 * Gd * inv_dd * d0;
        const T S = Gg + (T)2 * Gd * gm * inv_dd;
        const T dw = Gp0 + (inside ? S * rgn * (-w * F<T>::rcp(sq)) : (T)0);
        const T m1 = (T)(m - 1);
        const T c = F<T>::sqrt((T)4 * kap * kap + m1 * m1);
        const T b = m1 * F<T>::rcp((T)2 * kap + c);
        const T e = omw * F<T>::rcp(((T)1 + b) - w * ((T)1 - b));
        const T den = (T)1 - ((T)1 - b) * e;
        const T dw_db = (T)-2 * e * ((T)1 - e) * F<T>::rcp(den * den);
        T dk = dw * dw_db * ((T)-2 * b * F<T>::rcp(c));
        if (p.gkappa) dk += p.gkappa[row];
        p.dkappa[row] = dk * ksig;
      }
    }
  }
}

// Backward of the circle (m = 2, dense rows, regenerated sign): a thread owns the GLOBAL pair of rows (2p, 2p+1) --
// 128-bit accesses to mu / grad z / d mu, and the ONE Philox call whose words hold both rows' tangent signs (one call
// per row, for one bit of it, was a third of the lane-per-row kernel's instructions).  A call whose first or last row
// has no partner (odd row offset, odd count) handles that row alone with 64-bit accesses.
// KP: the per-row arrays (kappa, w, 1 - w, d kappa, grad kappa) of a pair are one aligned 64-bit unit each (even row
// offset, arrays aligned to two elements): one access per pair instead of two half-used ones.
template <typename T, bool KP>
__global__ __launch_bounds__(256) void vmf_circle_bwd_kernel(GenParams<T> p) {
  const PhiloxKey key = philox_resolve(p.ka);
  const uint64_t p0 = p.row0 >> 1;
  const int64_t npairs = (int64_t)(((p.row0 + (uint64_t)p.n + 1u) >> 1) - p0);  // global pairs the call touches
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool norm = p.normalize != 0, sp = p.softplus != 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += stride) {
    const uint64_t P = p0 + (uint64_t)t;
    const int64_t ia = (int64_t)(2 * P - p.row0), ib = ia + 1;  // local indices of the pair's rows (ia may be -1)
    const bool va = ia >= 0, vb = ib < p.n;
    const U4 r = philox4x32_10(P, 0u, key);
    const bool nega = (r.x & 1u) != 0u, negb = (r.z & 1u) != 0u;
    if (va && vb) {
      const RowPair<T> mu = load_rows2_nt<T>(p.mu, ia);
      const RowPair<T> G = load_rows2_nt<T>(p.gz, ia);
      T ra, rb, wa, wb, oa, ob;
      if (KP) {
        const Pair<T> k2 = load_pair_nt<T>(p.kappa, ia >> 1), w2 = load_pair_nt<T>(p.w_in, ia >> 1);
        ra = k2.a; rb = k2.b; wa = w2.a; wb = w2.b;
        oa = (T)1 - wa; ob = (T)1 - wb;
        if (p.omw_in) {
          const Pair<T> o2 = load_pair_nt<T>(p.omw_in, ia >> 1);
          oa = o2.a; ob = o2.b;
        }
      } else {
        ra = __builtin_nontemporal_load(p.kappa + ia); rb = __builtin_nontemporal_load(p.kappa + ib);
        wa = __builtin_nontemporal_load(p.w_in + ia); wb = __builtin_nontemporal_load(p.w_in + ib);
        oa = p.omw_in ? __builtin_nontemporal_load(p.omw_in + ia) : (T)1 - wa;
        ob = p.omw_in ? __builtin_nontemporal_load(p.omw_in + ib) : (T)1 - wb;
      }
      T sa, sb, dka, dkb;
      const T ka_ = kappa_of<T>(ra, sp, &sa), kb_ = kappa_of<T>(rb, sp, &sb);
      RowPair<T> o;
      circle_bwd_row<T>(mu.v[0], mu.v[1], G.v[0], G.v[1], nega, wa, oa, ka_, norm, &o.v[0], &o.v[1], &dka);
      circle_bwd_row<T>(mu.v[2], mu.v[3], G.v[2], G.v[3], negb, wb, ob, kb_, norm, &o.v[2], &o.v[3], &dkb);
      store_rows2_nt<T>(p.dmu, ia, o);
      if (p.gkappa) { dka += p.gkappa[ia]; dkb += p.gkappa[ib]; }
      if (KP) {
        store_pair_nt<T>(p.dkappa, ia >> 1, dka * sa, dkb * sb);
      } else {
        __builtin_nontemporal_store(dka * sa, p.dkappa + ia);
        __builtin_nontemporal_store(dkb * sb, p.dkappa + ib);
      }
    } else {
      const int64_t i = va ? ia : ib;  // the call's first row (odd global index) or its last one (even)
      const Pair<T> mu = load_pair_nt<T>(p.mu, i);
      const Pair<T> G = load_pair_nt<T>(p.gz, i);
      T sg, o0, o1, dk;
      const T kap = kappa_of<T>(p.kappa[i], sp, &sg);
      const T w = p.w_in[i];
      const T omw = p.omw_in ? p.omw_in[i] : (T)1 - w;
      circle_bwd_row<T>(mu.a, mu.b, G.a, G.b, va ? nega : negb, w, omw, kap, norm, &o0, &o1, &dk);
      store_pair_nt<T>(p.dmu, i, o0, o1);
      if (p.gkappa) dk += p.gkappa[i];
      p.dkappa[i] = dk * sg;
    }
  }
}

// Backward for the narrow widths of vmf_rsample_rowlane_kernel (fp32, counter noise): ONE lane owns a row, as there.  The
// tile's mu, then its grad z, move from HBM as flat arrays of 16-byte units through one wave-private LDS slab into the
// lane's registers; d mu goes back through the same slab.  No cross-lane sums; the lane-group kernel above gives these
// rows scalar, guarded element accesses and idle lanes (m = 10: 47 % of the HBM rate).  Same arithmetic as the kernel
// above, sums taken in element order.
template <int M>
__global__ __launch_bounds__(RLB_WAVES * 64) void vmf_rsample_bwd_rowlane_kernel(GenParams<float> p) {
  using T = float;
  constexpr int RL_TILE = rl_tile(M);
  constexpr int UNITS = RL_TILE * M / 4;
  constexpr int UPL = (UNITS + 63) / 64;
  constexpr int CHUNKS = (M + 3) / 4;
  __shared__ __align__(16) T s_rows[RLB_WAVES][RL_TILE * M];
  const PhiloxKey key = philox_resolve(p.ka);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* const srow = s_rows[wave];
  const int64_t tile0 = ((int64_t)blockIdx.x * RLB_WAVES + wave) * RL_TILE;  // wave-uniform
  const int64_t left = p.n - tile0;
  if (left <= 0) return;  // the kernel has no workgroup barrier
  const int ntile = left < RL_TILE ? (int)left : RL_TILE;
  const bool full = p.dense16 && ntile == RL_TILE;
  const T* const mu_t = p.mu + tile0 * p.mu_stride;
  const T* const gz_t = p.gz + tile0 * M;
  T* const dmu_t = p.dmu + tile0 * M;
  const uint64_t grow0 = p.row0 + (uint64_t)tile0;
  u32x4 st_mu[UPL], st_g[UPL];
  if (full) {
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const int q = lane + 64 * s;
      if (UNITS % 64 == 0 || q < UNITS) {
        st_mu[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(mu_t) + q);
        st_g[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gz_t) + q);
      }
    }
  }
  // a lane's rows: lane, lane + 64, ... (RL_TILE / 64 of them), held in registers across the two slab passes
  constexpr int SUBS = RL_TILE / 64;
  T mu[SUBS][M], G[SUBS][M];
  auto park = [&](const u32x4 (&st)[UPL], const T* src, int64_t stride) {
    if (full) {
#pragma unroll
      for (int s = 0; s < UPL; ++s) {
        const int q = lane + 64 * s;
        if (UNITS % 64 == 0 || q < UNITS) reinterpret_cast<u32x4*>(srow)[q] = st[s];
      }
    } else {
      for (int e = lane; e < ntile * M; e += 64) srow[e] = src[(e / M) * stride + e % M];
    }
    wave_lds_sync();
  };
  auto take = [&](T (&dst)[SUBS][M]) {
#pragma unroll
    for (int sub = 0; sub < SUBS; ++sub) {
      const int r = sub * 64 + lane;
      if (r < ntile) {
        if (M % 2 == 0) {
#pragma unroll
          for (int e = 0; e < M; e += 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(srow + r * M + e);
            dst[sub][e] = v.x; dst[sub][e + 1] = v.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < M; ++e) dst[sub][e] = srow[r * M + e];
        }
      }
    }
    wave_lds_sync();
  };
  park(st_mu, mu_t, p.mu_stride);
  take(mu);
  park(st_g, gz_t, M);
  take(G);
  const bool sp = p.softplus != 0;
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    const int r = sub * 64 + lane;
    if (r < ntile) {
      const int64_t row = tile0 + r;
      T g[M];
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c) {
        const U4 rr = philox4x32_10(grow0 + (uint64_t)r, SLOT_GAUSS + (uint32_t)c, key);
        T n0, n1;
        box_muller<T>(u01<T>(rr.x), u01<T>(rr.y), &n0, &n1);
        if (4 * c < M) g[4 * c] = n0;
        if (4 * c + 1 < M) g[4 * c + 1] = n1;
        if (4 * c + 2 < M) {
          box_muller<T>(u01<T>(rr.z), u01<T>(rr.w), &n0, &n1);
          g[4 * c + 2] = n0;
          if (4 * c + 3 < M) g[4 * c + 3] = n1;
        }
      }
      g[0] = (T)0;
      T gg = 0, dd = 0, gm = 0, Gd = 0, Gg = 0;
#pragma unroll
      for (int e = 1; e < M; ++e) {
        gg += g[e] * g[e];
        dd += mu[sub][e] * mu[sub][e];
        gm += g[e] * mu[sub][e];
        Gd += G[sub][e] * mu[sub][e];
        Gg += G[sub][e] * g[e];
      }
      T mu0 = mu[sub][0];
      const T G0 = G[sub][0];
      const T inv = p.normalize ? F<T>::rsqrt(dd + mu0 * mu0) : (T)1;
      mu0 *= inv;
      gm *= inv;
      Gd = G0 * ((T)1 - mu0) - Gd * inv;
      dd = ((T)1 - mu0) * ((T)1 - mu0) + dd * (inv * inv);
      const T w = p.w_in[row];
      const T omw = p.omw_in ? p.omw_in[row] : (T)1 - w;
      T ksig;
      const T kap = kappa_of<T>(p.kappa[row * p.kappa_stride], sp, &ksig);
      T c2 = omw * ((T)1 + w);
      const bool inside = c2 >= (T)1e-10;
      c2 = inside ? c2 : (T)1e-10;
      const T sq = F<T>::sqrt(c2), gn = F<T>::sqrt(gg);
      const T rgn = F<T>::rcp(gn);
      const T f = sq * rgn;
      const T d0 = (T)1 - mu0;
      const T zd = w * d0 - f * gm;
      const bool refl = dd > (T)1e-30;
      const T inv_dd = refl ? F<T>::rcp(dd) : (T)0;
      T out[M];
      T dot = 0;
#pragma unroll
      for (int e = 0; e < M; ++e) {
        const T zp = e == 0 ? w : f * g[e];
        const T mh = mu[sub][e] * inv;
        const T de = e == 0 ? d0 : -mh;
        out[e] = (T)2 * (zp * Gd + zd * G[sub][e] - (T)2 * zd * Gd * de * inv_dd) * inv_dd;
        dot += out[e] * mh;
      }
      if (p.normalize) {
#pragma unroll
        for (int e = 0; e < M; ++e) out[e] = (out[e] - mu[sub][e] * inv * dot) * inv;
      }
      if (M % 2 == 0) {
#pragma unroll
        for (int e = 0; e < M; e += 2) {
          f32x2 v;
          v.x = out[e]; v.y = out[e + 1];
          *reinterpret_cast<f32x2*>(srow + r * M + e) = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < M; ++e) srow[r * M + e] = out[e];
      }
      const T Gp0 = G0 - (T)2 * Gd * inv_dd * d0;
      const T S = Gg + (T)2 * Gd * gm * inv_dd;
      const T dw = Gp0 + (inside ? S * rgn * (-w * F<T>::rcp(sq)) : (T)0);
      const T m1 = (T)(M - 1);
      const T c = F<T>::sqrt((T)4 * kap * kap + m1 * m1);
      const T b = m1 * F<T>::rcp((T)2 * kap + c);
      const T e_ = omw * F<T>::rcp(((T)1 + b) - w * ((T)1 - b));
      const T den = (T)1 - ((T)1 - b) * e_;
      const T dw_db = (T)-2 * e_ * ((T)1 - e_) * F<T>::rcp(den * den);
      T dk = dw * dw_db * ((T)-2 * b * F<T>::rcp(c));
      if (p.gkappa) dk += p.gkappa[row];
      p.dkappa[row] = dk * ksig;
    }
  }
  wave_lds_sync();
  if (full) {
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const int q = lane + 64 * s;
      if (UNITS % 64 == 0 || q < UNITS) {
        f32x4 v;
        const u32x4 raw = reinterpret_cast<const u32x4*>(srow)[q];
        __builtin_memcpy(&v, &raw, 16);
        store_nt(v, dmu_t + 4 * q);
      }
    }
  } else {
    for (int e = lane; e < ntile * M; e += 64) dmu_t[e] = srow[e];
  }
}

static void launch_bwd_rowlane(const GenParams<float>& p, hipStream_t s) {
  const int64_t tiles = (p.n + rl_tile(p.m) - 1) / rl_tile(p.m);
  const dim3 grid((unsigned)((tiles + RLB_WAVES - 1) / RLB_WAVES)), block(RLB_WAVES * 64);
  switch (p.m) {
#define VMF_RL(MM) case MM: hipLaunchKernelGGL((vmf_rsample_bwd_rowlane_kernel<MM>), grid, block, 0, s, p); break
    VMF_RL(3); VMF_RL(5); VMF_RL(6); VMF_RL(7); VMF_RL(9); VMF_RL(10); VMF_RL(11); VMF_RL(12); VMF_RL(13); VMF_RL(14);
    VMF_RL(15); VMF_RL(17); VMF_RL(18); VMF_RL(19); VMF_RL(20); VMF_RL(21); VMF_RL(22); VMF_RL(23); VMF_RL(24); VMF_RL(25);
    VMF_RL(26); VMF_RL(27); VMF_RL(28); VMF_RL(29); VMF_RL(30); VMF_RL(31);
#undef VMF_RL
    default: break;
  }
}

// log C_m(kappa_i) for every row: pre-pass of the fused log_prob (keeps the Bessel code, and its registers,
// out of the streaming kernel)
template <typename T>
__global__ __launch_bounds__(256) void vmf_log_norm_kernel(const T* __restrict__ kappa, int64_t kappa_stride,
                                                           T* __restrict__ out, int64_t n, VmfDim dim, VmfDimTab tab,
                                                           int softplus) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T sig_;
    const T kap = kappa_of<T>(kappa[i * kappa_stride], softplus != 0, &sig_);
    if (sizeof(T) == 4) out[i] = (T)vmf_log_norm_f32(dim, (float)kap);
    else out[i] = (T)vmf_log_norm_tab(dim, tab, (double)kap);
  }
}

// ---- launch plumbing ---------------------------------------------------------------------------------
// Lanes per row x passes for a row of `chunks` 16-byte chunks: the smallest power of two, or three times a power of two,
// that holds the row (widths of 3 x 2^k elements -- 96, 192, 384, 768 -- fit exactly; before the three-pass shapes
// existed they ran with a quarter of their lanes idle, and a row just above a power of two with half of them:
// 2^21 rows of m = 768 2.52 -> 2.02 ms, 2^23 rows of m = 96 1.43 -> 1.20, of m = 130 3.04 -> 2.03).
static void pick_shape(int chunks, int* lpr, int* passes) {
  // (four lanes x three passes for 9-12 chunks measured slower than sixteen lanes with idle ones: m = 40 backward 2x)
  static const int table[][3] = {{1, 1, 1},   {2, 2, 1},   {4, 4, 1},   {8, 8, 1},    {16, 16, 1},  {24, 8, 3},   {32, 32, 1},
                                 {48, 16, 3}, {64, 64, 1}, {96, 32, 3}, {128, 64, 2}, {192, 64, 3}, {256, 64, 4}, {512, 64, 8}};
  for (const auto& t : table)
    if (chunks <= t[0]) {
      *lpr = t[1];
      *passes = t[2];
      return;
    }
  *lpr = 64;
  *passes = 8;
}
#define VMF_SHAPE_SWITCH(chunks, CALL)                                \
  do {                                                                \
    int lpr_, passes_;                                                \
    pick_shape((chunks), &lpr_, &passes_);                            \
    switch (lpr_ * 16 + passes_) {                                    \
      case 1 * 16 + 1: CALL(1, 1); break;                             \
      case 2 * 16 + 1: CALL(2, 1); break;                             \
      case 4 * 16 + 1: CALL(4, 1); break;                             \
      case 8 * 16 + 1: CALL(8, 1); break;                             \
      case 16 * 16 + 1: CALL(16, 1); break;                           \
      case 8 * 16 + 3: CALL(8, 3); break;                             \
      case 32 * 16 + 1: CALL(32, 1); break;                           \
      case 16 * 16 + 3: CALL(16, 3); break;                           \
      case 64 * 16 + 1: CALL(64, 1); break;                           \
      case 32 * 16 + 3: CALL(32, 3); break;                           \
      case 64 * 16 + 2: CALL(64, 2); break;                           \
      case 64 * 16 + 3: CALL(64, 3); break;                           \
      case 64 * 16 + 4: CALL(64, 4); break;                           \
      default: CALL(64, 8); break;                                    \
    }                                                                 \
  } while (0)

template <typename T, int MODE>
static void launch_fwd(const FwdParams<T>& p, int grid, hipStream_t s) {
  const int chunks = (p.m + 3) / 4;
#define VMF_CALL(L, PP) \
  hipLaunchKernelGGL((vmf_rsample_fwd_kernel<T, L, PP, MODE>), dim3(grid), dim3(WAVES * 64), 0, s, p)
  VMF_SHAPE_SWITCH(chunks, VMF_CALL);
#undef VMF_CALL
}

template <typename T, bool VEC>
static void launch_bwd(const GenParams<T>& p, int grid, int block_waves, hipStream_t s) {
  const int chunks = (p.m + 3) / 4;
#define VMF_CALL(L, PP) \
  hipLaunchKernelGGL((vmf_rsample_bwd_kernel<T, L, PP, VEC>), dim3(grid), dim3(block_waves * 64), 0, s, p)
  VMF_SHAPE_SWITCH(chunks, VMF_CALL);
#undef VMF_CALL
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <typename T>
static int rsample_fwd_t(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, void* z,
                         void* w_out, void* omw_out, int32_t* accept_idx, void* logp_out, void* kappa_out, int64_t n,
                         int m, const void* prop_e, const void* prop_u, int n_prop, const void* gauss, int max_prop,
                         uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int flags,
                         hipStream_t s) {
  FwdParams<T> p = {};
  p.kappa_out = (T*)kappa_out;
  p.softplus = (flags & VMF_FLAG_SOFTPLUS_KAPPA) ? 1 : 0;
  p.mu = (const T*)mu; p.kappa = (const T*)kappa; p.z = (T*)z;
  p.w_out = (T*)w_out; p.omw_out = (T*)omw_out; p.accept_idx = accept_idx; p.logp = (T*)logp_out;
  p.prop_e = (const T*)prop_e; p.prop_u = (const T*)prop_u; p.gauss = (const T*)gauss;
  p.n = n; p.row0 = row0; p.mu_stride = (int)mu_stride; p.kappa_stride = (int)kappa_stride; p.m = m;
  p.max_prop = prop_e ? n_prop : max_prop;
  p.normalize = (flags & VMF_FLAG_NORMALIZE_MU) ? 1 : 0;
  p.ka = make_philox_arg(seed, offset, offset_dev);
  const bool vec = (m % 4 == 0) && (mu_stride % 4 == 0) && aligned16(mu) && aligned16(z) && (!gauss || aligned16(gauss));
  const int chunks = (m + 3) / 4;
  int lpr, kpasses;  // the instantiation VMF_SHAPE_SWITCH picks: it sets the tile size
  pick_shape(chunks, &lpr, &kpasses);
  const bool full = vec && m == 4 * lpr * kpasses;
  const int64_t n_tiles = (n + tile_rows(lpr, kpasses) - 1) / tile_rows(lpr, kpasses);
  int64_t gb = (n_tiles + WAVES - 1) / WAVES;
  // One tile per wave, every tile its own wave: workgroups are dispatched in index order and the kernel maps them
  // to tiles so that the waves in flight on one XCD work on neighbouring tiles.  A persistent grid of 2048 workgroups
  // walking strided tiles measured 13 % slower on the same kernel body.
  VMF_REQUIRE(gb <= 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows need more than 2^31 workgroups", (long long)n);
  const int grid = (int)(gb < 1 ? 1 : gb);
  {
    const VmfDim dim = make_vmf_dim(m);
    p.lc_v = (float)dim.v; p.lc_lg = (float)dim.lg_v1; p.lc_c0 = (float)dim.half_m_log2pi;
  }
  if (logp_out && sizeof(T) == 8)
    hipLaunchKernelGGL(vmf_log_norm_kernel<T>, dim3(vmf_grid_for(n, 256)), dim3(256), 0, s, (const T*)kappa,
                       kappa_stride, (T*)logp_out, n, make_vmf_dim(m), make_vmf_dim_tab(make_vmf_dim(m)), p.softplus);
  const bool pair_ok = m == 2 && !prop_e && !gauss && mu_stride % 2 == 0 && ((uintptr_t)mu % (2 * sizeof(T))) == 0 &&
                       ((uintptr_t)z % (2 * sizeof(T))) == 0;
  if constexpr (sizeof(T) == 4) {
    // narrow rows, counter noise: one lane per row whatever the layout, so that a draw does not depend on where a
    // shard's rows start
    if (rowlane_ok(m) && !prop_e) {
      VMF_REQUIRE((n + rl_fwd_tile(m) - 1) / rl_fwd_tile(m) / rl_waves(m) < 0x7fffffff, VMF_ERR_UNSUPPORTED,
                  "n=%lld rows need more than 2^31 workgroups", (long long)n);
      p.dense16 = mu_stride == m && aligned16(mu) && aligned16(z);
      launch_rowlane(p, s);
      return VMF_OK;
    }
  }
  if constexpr (sizeof(T) == 4) {
    // dense aligned circles whose pairs of rows (2p, 2p+1) in the GLOBAL numbering are 16-byte units: two rows per lane
    if (pair_ok && mu_stride == 2 && kappa_stride == 1 && (row0 & 1u) == 0 && aligned16(mu) && aligned16(z) &&
        ((uintptr_t)kappa & 7u) == 0) {
      const int64_t nt = (n + CF_PIPE * CF_TILE - 1) / (CF_PIPE * CF_TILE);
      const int64_t nb = (nt + WAVES - 1) / WAVES;
      VMF_REQUIRE(nb <= 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows need more than 2^31 workgroups", (long long)n);
      const bool opts = p.softplus || p.normalize || kappa_out || w_out || omw_out || accept_idx || logp_out;
      if (opts) hipLaunchKernelGGL((vmf_circle_fwd_kernel<true>), dim3((unsigned)nb), dim3(WAVES * 64), 0, s, p);
      else hipLaunchKernelGGL((vmf_circle_fwd_kernel<false>), dim3((unsigned)nb), dim3(WAVES * 64), 0, s, p);
      return VMF_OK;
    }
  }
  if (pair_ok) {
    const int64_t nt = (n + VMF_TILE_ROWS_CIRCLE - 1) / VMF_TILE_ROWS_CIRCLE;
    hipLaunchKernelGGL((vmf_rsample_fwd_kernel<T, 1, 1, 3>), dim3((unsigned)((nt + WAVES - 1) / WAVES)), dim3(WAVES * 64), 0,
                       s, p);
  } else if (full && !prop_e && !p.normalize && !p.softplus && !kappa_out) launch_fwd<T, 2>(p, grid, s);
  else if (vec) launch_fwd<T, 1>(p, grid, s);
  else launch_fwd<T, 0>(p, grid, s);
  return VMF_OK;
}

extern "C" int vmf_rsample_fwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, void* z,
                               void* w_out, void* omw_out, int32_t* accept_idx, void* logp_out, void* kappa_out,
                               int64_t n, int m, const void* prop_e, const void* prop_u, int n_prop, const void* gauss,
                               int max_prop, uint64_t row0, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                               int flags, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_rsample_fwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  VMF_REQUIRE((flags & ~(VMF_FLAG_NORMALIZE_MU | VMF_FLAG_SOFTPLUS_KAPPA)) == 0, VMF_ERR_BAD_ARG,
              "vmf_rsample_fwd: unknown flags %d", flags);
  VMF_REQUIRE(!kappa_out || kappa_stride == 1, VMF_ERR_BAD_ARG, "vmf_rsample_fwd: kappa_out needs kappa_stride 1");
  VMF_REQUIRE(m >= 2, VMF_ERR_BAD_ARG, "m=%d < 2", m);
  VMF_REQUIRE(m <= 2048, VMF_ERR_UNSUPPORTED, "m=%d > 2048 is not supported", m);
  VMF_REQUIRE(mu_stride == 0 || mu_stride >= m, VMF_ERR_BAD_ARG, "mu_stride=%lld", (long long)mu_stride);
  VMF_REQUIRE(kappa_stride == 0 || kappa_stride == 1, VMF_ERR_BAD_ARG, "kappa_stride must be 0 or 1");
  const bool pre = prop_e || prop_u || gauss;
  if (pre) {
    VMF_REQUIRE(prop_e && prop_u && gauss, VMF_ERR_BAD_ARG, "pre-drawn mode needs prop_e, prop_u and gauss");
    VMF_REQUIRE(n_prop >= 1 && n_prop <= 65536, VMF_ERR_BAD_ARG, "n_prop=%d outside [1, 65536]", n_prop);
  } else {
    // proposal k draws from Philox slot k and the Gaussians from slots 0x10000 + chunk: more than 65536 proposals would
    // reuse counters (and a NaN kappa, which never accepts, would spin its wave up to the cap)
    VMF_REQUIRE(max_prop >= 1 && max_prop <= 65536, VMF_ERR_BAD_ARG, "max_prop=%d outside [1, 65536]", max_prop);
  }
  VMF_REQUIRE(mu_stride <= 0x7fffffff, VMF_ERR_UNSUPPORTED, "mu_stride=%lld does not fit 32 bits", (long long)mu_stride);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mu && kappa && z, VMF_ERR_BAD_ARG, "vmf_rsample_fwd: null mu/kappa/z");
  int rc;
  VMF_DISPATCH_DTYPE(dtype, rc = rsample_fwd_t<T>(mu, mu_stride, kappa, kappa_stride, z, w_out, omw_out, accept_idx,
                                                  logp_out, kappa_out, n, m, prop_e, prop_u, n_prop, gauss, max_prop, row0,
                                                  seed, offset, offset_dev, flags, (hipStream_t)stream));
  if (rc) return rc;
  return vmf_check_launch("vmf_rsample_fwd");
}

template <typename T>
static int rsample_bwd_t(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, const void* w,
                         const void* omw, const void* gz, const void* gauss, const void* gkappa, void* dmu,
                         void* dkappa, int64_t n, int m, uint64_t row0, uint64_t seed, uint64_t offset,
                         const uint64_t* offset_dev, int flags, hipStream_t s) {
  GenParams<T> p = {};
  p.softplus = (flags & VMF_FLAG_SOFTPLUS_KAPPA) ? 1 : 0;
  p.gkappa = (const T*)gkappa;
  p.normalize = (flags & VMF_FLAG_NORMALIZE_MU) ? 1 : 0;
  p.mu = (const T*)mu; p.mu_stride = mu_stride; p.kappa = (const T*)kappa; p.kappa_stride = kappa_stride;
  p.gauss = (const T*)gauss; p.n = n; p.m = m; p.row0 = row0; p.ka = make_philox_arg(seed, offset, offset_dev);
  p.dim = make_vmf_dim(m);
  p.w_in = (const T*)w; p.omw_in = (const T*)omw; p.gz = (const T*)gz; p.dmu = (T*)dmu; p.dkappa = (T*)dkappa;
  const bool vec = (m % 4 == 0) && (mu_stride % 4 == 0) && aligned16(mu) && aligned16(gz) && aligned16(dmu) &&
                   (!gauss || aligned16(gauss));
  const int chunks = (m + 3) / 4;
  int lpr, kpasses;  // the instantiation VMF_SHAPE_SWITCH picks: it sets the steps per wave
  pick_shape(chunks, &lpr, &kpasses);
  const int64_t steps = (n + (64 / lpr) - 1) / (64 / lpr);
  const int spw = bwd_steps(lpr, kpasses);
  const int64_t waves = (steps + spw - 1) / spw;
  const int bw = bwd_waves(lpr, kpasses);
  int64_t gb = (waves + bw - 1) / bw;
  VMF_REQUIRE(gb <= 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows need more than 2^31 workgroups", (long long)n);
  const int grid = (int)(gb < 1 ? 1 : gb);
  // dense circles with the regenerated sign: a thread per global pair of rows
  if (m == 2 && !gauss && mu_stride == 2 && kappa_stride == 1 && ((uintptr_t)mu % (2 * sizeof(T))) == 0 &&
      ((uintptr_t)gz % (2 * sizeof(T))) == 0 && ((uintptr_t)dmu % (2 * sizeof(T))) == 0) {
    const auto al2 = [](const void* q) { return ((uintptr_t)q % (2 * sizeof(T))) == 0; };
    const bool kp = (row0 & 1u) == 0 && al2(kappa) && al2(w) && al2(omw) && al2(dkappa);
    const dim3 cgrid(vmf_grid_for(n / 2 + 1, 256, 1 << 22));
    if (kp) hipLaunchKernelGGL((vmf_circle_bwd_kernel<T, true>), cgrid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((vmf_circle_bwd_kernel<T, false>), cgrid, dim3(256), 0, s, p);
    return VMF_OK;
  }
  if constexpr (sizeof(T) == 4) {
    // narrow rows of a width that is not a multiple of four, regenerated noise: one lane per row whatever the layout
    if (rowlane_bwd_ok(m) && !gauss) {
      VMF_REQUIRE((n + rl_tile(m) - 1) / rl_tile(m) / RLB_WAVES < 0x7fffffff, VMF_ERR_UNSUPPORTED,
                  "n=%lld rows need more than 2^31 workgroups", (long long)n);
      p.dense16 = mu_stride == m && aligned16(mu) && aligned16(gz) && aligned16(dmu);
      launch_bwd_rowlane(p, s);
      return VMF_OK;
    }
  }
  if (vec) launch_bwd<T, true>(p, grid, bw, s);
  else launch_bwd<T, false>(p, grid, bw, s);
  return VMF_OK;
}

extern "C" int vmf_rsample_bwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride,
                               const void* w, const void* omw, const void* grad_z, const void* gauss,
                               const void* grad_kappa, void* dmu, void* dkappa, int64_t n, int m, uint64_t row0,
                               uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int flags, int dtype,
                               vmf_stream_t stream) {
  VMF_RANGE("vmf_rsample_bwd");
  VMF_REQUIRE(n >= 0, VMF_ERR_BAD_ARG, "n < 0");
  VMF_REQUIRE((flags & ~(VMF_FLAG_NORMALIZE_MU | VMF_FLAG_SOFTPLUS_KAPPA)) == 0, VMF_ERR_BAD_ARG,
              "vmf_rsample_bwd: unknown flags %d", flags);
  VMF_REQUIRE(!grad_kappa || kappa_stride == 1, VMF_ERR_BAD_ARG, "vmf_rsample_bwd: grad_kappa needs kappa_stride 1");
  VMF_REQUIRE(m >= 2, VMF_ERR_BAD_ARG, "m=%d < 2", m);
  VMF_REQUIRE(m <= 2048, VMF_ERR_UNSUPPORTED, "m=%d > 2048 is not supported", m);
  VMF_REQUIRE(mu_stride == 0 || mu_stride >= m, VMF_ERR_BAD_ARG, "mu_stride=%lld", (long long)mu_stride);
  VMF_REQUIRE(kappa_stride == 0 || kappa_stride == 1, VMF_ERR_BAD_ARG, "kappa_stride must be 0 or 1");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mu && kappa && w && grad_z && dmu && dkappa, VMF_ERR_BAD_ARG, "vmf_rsample_bwd: null pointer");
  int rc;
  VMF_DISPATCH_DTYPE(dtype, rc = rsample_bwd_t<T>(mu, mu_stride, kappa, kappa_stride, w, omw, grad_z, gauss, grad_kappa,
                                                  dmu, dkappa, n, m, row0, seed, offset, offset_dev, flags,
                                                  (hipStream_t)stream));
  if (rc) return rc;
  return vmf_check_launch("vmf_rsample_bwd");
}

// ---- log_prob, entropy, KL ----------------------------------------------------------------------------
// log_prob runs in two passes so that the Bessel code (double arithmetic, one evaluation per ROW) never idles the
// lanes that stream the rows: pass 1 is elementwise over kappa (log C_m, or -g A_m for the backward), pass 2 uses the
// sampler's row mapping (LPR lanes x 4 elements, DPP reduction) for <mu, x> and adds kappa <mu, x>.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void vmf_log_prob_rowterm_kernel(const T* __restrict__ kappa, int64_t kappa_stride,
                                                                    const T* __restrict__ gl, T* __restrict__ out,
                                                                    int64_t n, VmfDim dim, VmfDimTab tab) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double k = (double)kappa[i * kappa_stride];
    if (!BWD) out[i] = sizeof(T) == 4 ? (T)(dim.m == 2 ? vmf_log_norm_circle_f32((float)k) : vmf_log_norm_f32(dim, (float)k))
                                      : (T)vmf_log_norm_tab(dim, tab, k);
    else out[i] = (T)(-(double)gl[i] * vmf_mean_resultant_tab(dim, tab, k));  // d log C_m / dk = -A_m(k)
  }
}

template <typename T>
struct LogProbParams {
  const T* mu; const T* kappa; const T* x; const T* gl;
  T* logp; T* dmu; T* dx; T* dkappa;
  int64_t n; int64_t mu_stride; int kappa_stride, m;
};

template <typename T, int LPR, int P, bool VEC, bool BWD>
__global__ __launch_bounds__(WAVES * 64) void vmf_log_prob_dot_kernel(LogProbParams<T> p) {
  const int lane = threadIdx.x & 63;
  constexpr int RPI = 64 / LPR;
  const int sub = lane % LPR, grp = lane / LPR;
  const int m = p.m;
  const int64_t n_steps = (p.n + RPI - 1) / RPI;
  const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // 1 to WAVES waves per block
  constexpr int LP_STEPS = lp_steps(BWD, LPR);
  const int64_t step_end = (wave_id + 1) * LP_STEPS < n_steps ? (wave_id + 1) * LP_STEPS : n_steps;
  for (int64_t step = wave_id * LP_STEPS; step < step_end; ++step) {  // wave-uniform
    const int64_t r = step * RPI + grp;
    const bool valid = r < p.n;
    const int64_t row = valid ? r : 0;
    const T* mrow = p.mu + row * p.mu_stride;
    const T* xrow = p.x + row * (int64_t)m;
    T mu[P][4], x[P][4];
    T dot = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int chunk = j * LPR + sub;
      load4<T, VEC, VMF_LP_STREAM>(mrow, 4 * chunk, m, mu[j]);
      load4<T, VEC, VMF_LP_STREAM>(xrow, 4 * chunk, m, x[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) dot += mu[j][i] * x[j][i];
    }
    dot = group_allreduce_sum<LPR, T>(dot);
    if (!valid) continue;
    const T k = p.kappa[row * p.kappa_stride];
    if (!BWD) {
      if (sub == 0) p.logp[row] += k * dot;  // log C_m(kappa) was stored by pass 1
    } else {
      const T gk = p.gl[row] * k;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int chunk = j * LPR + sub;
        T a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = gk * x[j][i]; b[i] = gk * mu[j][i]; }
        if (p.dmu) store4<T, VEC, true>(p.dmu + row * (int64_t)m, 4 * chunk, m, a);
        if (p.dx) store4<T, VEC, true>(p.dx + row * (int64_t)m, 4 * chunk, m, b);
      }
      if (p.dkappa && sub == 0) p.dkappa[row] += p.gl[row] * dot;  // -g A_m(kappa) was stored by pass 1
    }
  }
}

template <typename T, bool VEC, bool BWD>
static void launch_log_prob(const LogProbParams<T>& p, int grid, hipStream_t s) {
  const int chunks = (p.m + 3) / 4;
#define VMF_CALL(L, PP) \
  hipLaunchKernelGGL((vmf_log_prob_dot_kernel<T, L, PP, VEC, BWD>), dim3(grid), dim3(lp_waves(BWD) * 64), 0, s, p)
  VMF_SHAPE_SWITCH(chunks, VMF_CALL);
#undef VMF_CALL
}

// log_prob for the narrow fp32 widths (3 to 32), dense rows: ONE lane owns a row, as in the rowlane sampler kernels.  The
// tile's mu, then its x, move from HBM as flat arrays of 16-byte units through one wave-private LDS slab; the Bessel term
// of the row is evaluated in the lane that owns it (forward: log C_m(kappa) in single precision; backward: -g A_m(kappa)
// in double precision through the coefficient tables), so each direction is ONE launch and the double arithmetic of one
// wave hides behind the loads of the others.  d mu and d x go back through the slab.  Measured, 2^24 rows of 10 floats:
// forward 0.468 -> 0.233 ms, backward 0.886 -> 0.511 (two passes, lane groups with guarded scalar accesses before);
// 2^24 rows of 8: 0.285 -> 0.188 and 0.761 -> 0.495; 2^23 rows of 16: 0.227 -> 0.185 and 0.564 -> 0.430.
#ifndef VMF_LPR_FUSE_A
#define VMF_LPR_FUSE_A 1
#endif
template <int M, bool BWD>
__global__ __launch_bounds__(LPR_WAVES * 64) void vmf_log_prob_rowlane_kernel(LogProbParams<float> p, float lc_v, float lc_lg,
                                                                          float lc_c0, VmfDim dim, VmfDimTab tab) {
  using T = float;
  constexpr int RL_TILE = rl_tile(M);
  constexpr int UNITS = RL_TILE * M / 4;
  constexpr int UPL = (UNITS + 63) / 64;
  constexpr int SUBS = RL_TILE / 64;
  // a lane reads its row from the slab in 16-byte pieces when the width allows; rows of 8, 16 or 24 floats would then
  // start 8, 4 or 8 to a bank group, so those are kept 16 bytes apart (a pitch of an odd number of 16-byte units)
  constexpr int PITCH = M % 8 == 0 ? M + 4 : M;
  constexpr int UPR = M / 4;  // (used only when M % 8 == 0)
  __shared__ __align__(16) T s_rows[LPR_WAVES][RL_TILE * PITCH];
  auto unit_at = [](int q) { return M % 8 == 0 ? q + q / UPR : q; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* const srow = s_rows[wave];
  const int64_t tile0 = ((int64_t)blockIdx.x * LPR_WAVES + wave) * RL_TILE;  // wave-uniform
  const int64_t left = p.n - tile0;
  if (left <= 0) return;  // the kernel has no workgroup barrier
  const int ntile = left < RL_TILE ? (int)left : RL_TILE;
  const bool full = ntile == RL_TILE;
  const T* const mu_t = p.mu + tile0 * M;
  const T* const x_t = p.x + tile0 * M;
  T kap[SUBS], gl[SUBS];
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    const int r = sub * 64 + lane;
    kap[sub] = r < ntile ? p.kappa[(tile0 + r) * p.kappa_stride] : (T)1;
    gl[sub] = BWD && r < ntile ? p.gl[tile0 + r] : (T)0;
  }
  u32x4 st_mu[UPL], st_x[UPL];
  if (full) {
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const int q = lane + 64 * s;
      if (UNITS % 64 == 0 || q < UNITS) {
        st_mu[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(mu_t) + q);
        st_x[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x_t) + q);
      }
    }
  }
  auto park = [&](const u32x4 (&st)[UPL], const T* src) {
    if (full) {
#pragma unroll
      for (int s = 0; s < UPL; ++s) {
        const int q = lane + 64 * s;
        if (UNITS % 64 == 0 || q < UNITS) reinterpret_cast<u32x4*>(srow)[unit_at(q)] = st[s];
      }
    } else {
      for (int e = lane; e < ntile * M; e += 64) srow[(e / M) * PITCH + e % M] = src[e];
    }
    wave_lds_sync();
  };
  auto flush = [&](T* dst) {
    wave_lds_sync();
    if (full) {
#pragma unroll
      for (int s = 0; s < UPL; ++s) {
        const int q = lane + 64 * s;
        if (UNITS % 64 == 0 || q < UNITS) {
          f32x4 v;
          const u32x4 raw = reinterpret_cast<const u32x4*>(srow)[unit_at(q)];
          __builtin_memcpy(&v, &raw, 16);
          store_nt(v, dst + 4 * q);
        }
      }
    } else {
      for (int e = lane; e < ntile * M; e += 64) dst[e] = srow[(e / M) * PITCH + e % M];
    }
    wave_lds_sync();
  };
  auto row_get = [&](int r, T (&dst)[M]) {
    if (M % 4 == 0) {
#pragma unroll
      for (int e = 0; e < M; e += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(srow + r * PITCH + e);
        dst[e] = v.x; dst[e + 1] = v.y; dst[e + 2] = v.z; dst[e + 3] = v.w;
      }
    } else if (M % 2 == 0) {
#pragma unroll
      for (int e = 0; e < M; e += 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(srow + r * PITCH + e);
        dst[e] = v.x; dst[e + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < M; ++e) dst[e] = srow[r * PITCH + e];
    }
  };
  auto row_put = [&](int r, const T (&src)[M], T scale) {
    if (M % 4 == 0) {
#pragma unroll
      for (int e = 0; e < M; e += 4) {
        f32x4 v;
        v.x = scale * src[e]; v.y = scale * src[e + 1]; v.z = scale * src[e + 2]; v.w = scale * src[e + 3];
        *reinterpret_cast<f32x4*>(srow + r * PITCH + e) = v;
      }
    } else if (M % 2 == 0) {
#pragma unroll
      for (int e = 0; e < M; e += 2) {
        f32x2 v;
        v.x = scale * src[e]; v.y = scale * src[e + 1];
        *reinterpret_cast<f32x2*>(srow + r * PITCH + e) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < M; ++e) srow[r * PITCH + e] = scale * src[e];
    }
  };
  T mu[SUBS][M];
  park(st_mu, mu_t);
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    const int r = sub * 64 + lane;
    if (r < ntile) row_get(r, mu[sub]);
  }
  wave_lds_sync();
  park(st_x, x_t);
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    const int r = sub * 64 + lane;
    if (r < ntile) {
      const int64_t row = tile0 + r;
      T x[M];
      row_get(r, x);
      T dot = 0;
#pragma unroll
      for (int e = 0; e < M; ++e) dot += mu[sub][e] * x[e];
      if (!BWD) {
        p.logp[row] = vmf_log_norm_f32v(lc_v, lc_lg, lc_c0, kap[sub]) + kap[sub] * dot;
      } else {
        if (p.dkappa) {
          // -g A_m(kappa) in double precision, rounded, then + g <mu, x>: the two-pass kernels' bits
          const T first = VMF_LPR_FUSE_A ? (T)(-(double)gl[sub] * vmf_mean_resultant_tab(dim, tab, (double)kap[sub]))
                                         : p.dkappa[row];
          p.dkappa[row] = first + gl[sub] * dot;
        }
        if (p.dmu) row_put(r, x, gl[sub] * kap[sub]);
      }
    }
  }
  if (BWD) {
    T* const dmu_t = p.dmu ? p.dmu + tile0 * M : nullptr;
    T* const dx_t = p.dx ? p.dx + tile0 * M : nullptr;
    if (dmu_t) flush(dmu_t);
    if (dx_t) {
      if (!dmu_t) wave_lds_sync();
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const int r = sub * 64 + lane;
        if (r < ntile) row_put(r, mu[sub], gl[sub] * kap[sub]);
      }
      flush(dx_t);
    }
  }
}

static bool log_prob_rowlane_ok(int m) { return m >= 3 && m <= 32; }
template <bool BWD>
static void launch_log_prob_rowlane(const LogProbParams<float>& p, const VmfDim& dim, hipStream_t s) {
  const int64_t tiles = (p.n + rl_tile(p.m) - 1) / rl_tile(p.m);
  const dim3 grid((unsigned)((tiles + LPR_WAVES - 1) / LPR_WAVES)), block(LPR_WAVES * 64);
  const float v = (float)dim.v, lg = (float)dim.lg_v1, c0 = (float)dim.half_m_log2pi;
  const VmfDimTab tab = BWD ? make_vmf_dim_tab(dim) : VmfDimTab{};
  switch (p.m) {
#define VMF_RL(MM) \
  case MM: hipLaunchKernelGGL((vmf_log_prob_rowlane_kernel<MM, BWD>), grid, block, 0, s, p, v, lg, c0, dim, tab); break
    VMF_RL(3); VMF_RL(4); VMF_RL(5); VMF_RL(6); VMF_RL(7); VMF_RL(8); VMF_RL(9); VMF_RL(10); VMF_RL(11); VMF_RL(12);
    VMF_RL(13); VMF_RL(14); VMF_RL(15); VMF_RL(16); VMF_RL(17); VMF_RL(18); VMF_RL(19); VMF_RL(20); VMF_RL(21); VMF_RL(22);
    VMF_RL(23); VMF_RL(24); VMF_RL(25); VMF_RL(26); VMF_RL(27); VMF_RL(28); VMF_RL(29); VMF_RL(30); VMF_RL(31);
    VMF_RL(32);
#undef VMF_RL
    default: break;
  }
}

template <typename T, bool BWD>
static int log_prob_t(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride, const void* x,
                      const void* gl, void* logp, void* dmu, void* dx, void* dkappa, int64_t n, int m, hipStream_t s) {
  LogProbParams<T> p = {};
  p.mu = (const T*)mu; p.kappa = (const T*)kappa; p.x = (const T*)x; p.gl = (const T*)gl;
  p.logp = (T*)logp; p.dmu = (T*)dmu; p.dx = (T*)dx; p.dkappa = (T*)dkappa;
  p.n = n; p.mu_stride = mu_stride; p.kappa_stride = (int)kappa_stride; p.m = m;
  VmfDim dim = make_vmf_dim(m);
  T* rowterm = BWD ? (T*)dkappa : (T*)logp;
  bool rowlane = false;
  if constexpr (sizeof(T) == 4) {
    rowlane = log_prob_rowlane_ok(m) && mu_stride == m && aligned16(mu) && aligned16(x) && (!dmu || aligned16(dmu)) &&
              (!dx || aligned16(dx));
    if (rowlane)
      VMF_REQUIRE((n + rl_tile(m) - 1) / rl_tile(m) / LPR_WAVES < 0x7fffffff, VMF_ERR_UNSUPPORTED,
                  "n=%lld rows need more than 2^31 workgroups", (long long)n);
  }
  if (rowterm && !(rowlane && (!BWD || VMF_LPR_FUSE_A)))
    hipLaunchKernelGGL((vmf_log_prob_rowterm_kernel<T, BWD>), dim3(vmf_grid_for(n, 256)), dim3(256), 0, s, (const T*)kappa,
                       kappa_stride, (const T*)gl, rowterm, n, dim, make_vmf_dim_tab(dim));
  if constexpr (sizeof(T) == 4) {
    if (rowlane) {
      launch_log_prob_rowlane<BWD>(p, dim, s);
      return VMF_OK;
    }
  }
  const bool vec = (m % 4 == 0) && (mu_stride % 4 == 0) && aligned16(mu) && aligned16(x) && (!dmu || aligned16(dmu)) &&
                   (!dx || aligned16(dx));
  const int chunks = (m + 3) / 4;
  int lpr, kpasses;
  pick_shape(chunks, &lpr, &kpasses);
  const int64_t steps = (n + (64 / lpr) - 1) / (64 / lpr);
  const int64_t waves = (steps + lp_steps(BWD, lpr) - 1) / lp_steps(BWD, lpr);
  int64_t gb = (waves + lp_waves(BWD) - 1) / lp_waves(BWD);
  VMF_REQUIRE(gb <= 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows need more than 2^31 workgroups", (long long)n);
  const int grid = (int)(gb < 1 ? 1 : gb);
  if (vec) launch_log_prob<T, true, BWD>(p, grid, s);
  else launch_log_prob<T, false, BWD>(p, grid, s);
  return VMF_OK;
}

extern "C" int vmf_log_prob_fwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride,
                                const void* x, void* logp, int64_t n, int m, int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_log_prob_fwd");
  VMF_REQUIRE(n >= 0 && m >= 2, VMF_ERR_BAD_ARG, "bad n/m");
  VMF_REQUIRE(m <= 2048, VMF_ERR_UNSUPPORTED, "m=%d > 2048 is not supported", m);
  VMF_REQUIRE(kappa_stride == 0 || kappa_stride == 1, VMF_ERR_BAD_ARG, "kappa_stride must be 0 or 1");
  VMF_REQUIRE(mu_stride == 0 || mu_stride >= m, VMF_ERR_BAD_ARG, "mu_stride=%lld", (long long)mu_stride);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mu && kappa && x && logp, VMF_ERR_BAD_ARG, "vmf_log_prob_fwd: null pointer");
  int rc;
  VMF_DISPATCH_DTYPE(dtype, rc = log_prob_t<T, false>(mu, mu_stride, kappa, kappa_stride, x, nullptr, logp, nullptr,
                                                      nullptr, nullptr, n, m, (hipStream_t)stream));
  if (rc) return rc;
  return vmf_check_launch("vmf_log_prob_fwd");
}

extern "C" int vmf_log_prob_bwd(const void* mu, int64_t mu_stride, const void* kappa, int64_t kappa_stride,
                                const void* x, const void* gl, void* dmu, void* dx, void* dkappa, int64_t n, int m,
                                int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_log_prob_bwd");
  VMF_REQUIRE(n >= 0 && m >= 2, VMF_ERR_BAD_ARG, "bad n/m");
  VMF_REQUIRE(m <= 2048, VMF_ERR_UNSUPPORTED, "m=%d > 2048 is not supported", m);
  VMF_REQUIRE(kappa_stride == 0 || kappa_stride == 1, VMF_ERR_BAD_ARG, "kappa_stride must be 0 or 1");
  VMF_REQUIRE(mu_stride == 0 || mu_stride >= m, VMF_ERR_BAD_ARG, "mu_stride=%lld", (long long)mu_stride);
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(mu && kappa && x && gl, VMF_ERR_BAD_ARG, "vmf_log_prob_bwd: null pointer");
  int rc;
  VMF_DISPATCH_DTYPE(dtype, rc = log_prob_t<T, true>(mu, mu_stride, kappa, kappa_stride, x, gl, nullptr, dmu, dx, dkappa,
                                                     n, m, (hipStream_t)stream));
  if (rc) return rc;
  return vmf_check_launch("vmf_log_prob_bwd");
}

template <typename T>
__global__ __launch_bounds__(256) void vmf_entropy_kl_kernel(const T* __restrict__ kappa, T* __restrict__ entropy,
                                                             T* __restrict__ kl, T* __restrict__ dkl, int64_t n,
                                                             VmfDim dim, VmfDimTab tab) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double k = (double)kappa[i];
    double logc, A;
    vmf_log_norm_and_resultant_tab(dim, tab, k, &logc, &A);
    const double H = -logc - k * A;
    if (entropy) entropy[i] = (T)H;
    if (kl) kl[i] = (T)(-H + dim.log_area);
    // dKL/dk = k A'(k),  A' = 1 - A^2 - (m-1) A / k
    if (dkl) dkl[i] = (T)(k * (1.0 - A * A) - (double)(dim.m - 1) * A);
  }
}

extern "C" int vmf_entropy_kl_fwd(const void* kappa, void* entropy, void* kl, void* dkl_dk, int64_t n, int m,
                                  int dtype, vmf_stream_t stream) {
  VMF_RANGE("vmf_entropy_kl_fwd");
  VMF_REQUIRE(n >= 0 && m >= 2, VMF_ERR_BAD_ARG, "bad n/m");
  if (n == 0) return VMF_OK;
  VMF_REQUIRE(kappa, VMF_ERR_BAD_ARG, "vmf_entropy_kl_fwd: null kappa");
  VmfDim dim = make_vmf_dim(m);
  int grid = vmf_grid_for(n, 256);
  VMF_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(vmf_entropy_kl_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)kappa, (T*)entropy, (T*)kl,
                                               (T*)dkl_dk, n, dim, make_vmf_dim_tab(dim)));
  return vmf_check_launch("vmf_entropy_kl_fwd");
}
