// vmf3_fast.hip -- the two shapes VonMisesFisher3D.rsample (sampler/von_mises_fisher.py:66-86) takes in the reference's
// models, each with a kernel of its own; gfx950 only.
//
// (1) ROWS: scale (B,) -> z (B, 3) (CNN_experiment/circular_VAE.py:74-76).  A row of loc / eps / z is 12 bytes (fp32),
//     so a thread-per-row kernel reads and writes dwords at stride 12.  Here a wave owns a tile of 256 consecutive rows:
//     every (256, 3) piece moves between HBM and the wave as three coalesced 1 KiB accesses (16 bytes per lane) and is
//     re-dealt through a wave-private LDS slab so that lane l ends up with rows 4l .. 4l+3 -- 48 contiguous bytes, three
//     conflict-free ds_read_b128.  scale / u / w are four consecutive values per lane and need no re-deal.
//     Algorithmic bytes per row, fp32: forward 28 (loc 12, scale 4, z 12) + 16 with given noise (u 4, eps 12); backward
//     44 (loc, scale, grad z in; dloc, dscale out) + 16.
//
// (2) OUTER: scale (B, 1) -> z (B, B, 3), the broadcast the Spherical and Toroidal VAEs produce
//     (model/spherical_vae.py:81-83): z[i, j] = w_i l_j + f_i v_j.  Flattened over (j, c) this is a rank-one update per
//     row i, out[i, e] = w_i L[e] + f_i V[e] with L, V the flattened (n_j, 3) arrays of l and v -- no row structure
//     is left in the body.  A block computes L, V for its 256 loc rows and (w_i, f_i) for its i rows once (LDS), then
//     every lane keeps its 12 values of L and V in registers and writes 12 outputs per row: two multiply-adds per
//     component, 3 KiB contiguous per wave and row.  Algorithmic bytes: 12 per output row (fp32), write only.
//     Backward: grad z is read once.  G = grad z as an (n_i, 3 n_j) matrix gives dL/dw_i = <G_i, L> + c_i <G_i, V>
//     (row dot products) and dl = G^T w, dv = G^T f (weighted column sums); a block reduces its (rows x 768 columns)
//     tile in registers and writes one partial per row and per column into a caller workspace, and a finishing launch
//     adds the partials in a fixed order (same bits every run) and runs the per-i and per-j chain rules.
#include "vmf3_fast.h"

#include "vmf_math.h"

constexpr int V3_TILE_ROWS = 256;  // rows of a wave's tile: lane l owns rows 4 l .. 4 l + 3
// 16-byte units per lane of a (256, 3) tile: 3 for fp32, 6 for fp64
template <typename T>
constexpr int v3_units() { return 3 * (int)sizeof(T) / 4; }
template <typename T>
constexpr int v3_unit_elems() { return 16 / (int)sizeof(T); }

#ifndef VMF3_STAGE_LDS
#define VMF3_STAGE_LDS 1  // 0: every lane fetches its own 48 contiguous bytes (three 16-byte accesses at stride 48)
#endif

__device__ __forceinline__ u32x4 ld16_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st16_nt(u32x4 v, u32x4* p) {
  f32x4 f;
  __builtin_memcpy(&f, &v, 16);
  store_nt(f, reinterpret_cast<float*>(p));
}

// fetch a (256, 3) tile: U coalesced 16-byte loads per lane (unit lane + 64 s)
template <typename T>
__device__ __forceinline__ void tile_fetch(const T* tile, int lane, u32x4 (&r)[v3_units<T>()]) {
  constexpr int U = v3_units<T>();
  const u32x4* src = reinterpret_cast<const u32x4*>(tile);
#pragma unroll
  for (int s = 0; s < U; ++s) r[s] = ld16_nt(src + (VMF3_STAGE_LDS ? lane + 64 * s : U * lane + s));
}
// re-deal fetched units through the wave's slab: afterwards `rows` holds rows 4 lane .. 4 lane + 3 (12 values)
template <typename T>
__device__ __forceinline__ void tile_deal_in(u32x4* slab, int lane, u32x4 (&r)[v3_units<T>()], T (&rows)[12]) {
  constexpr int U = v3_units<T>();
#if VMF3_STAGE_LDS
#pragma unroll
  for (int s = 0; s < U; ++s) slab[lane + 64 * s] = r[s];
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < U; ++s) r[s] = slab[U * lane + s];
  wave_lds_sync();
#endif
  __builtin_memcpy(rows, r, sizeof(T) * 12);
}
// the way back: rows 4 lane .. 4 lane + 3 -> coalesced streaming stores
template <typename T>
__device__ __forceinline__ void tile_store(T* tile, u32x4* slab, int lane, const T (&rows)[12]) {
  constexpr int U = v3_units<T>();
  u32x4 r[U];
  __builtin_memcpy(r, rows, sizeof(T) * 12);
  u32x4* dst = reinterpret_cast<u32x4*>(tile);
#if VMF3_STAGE_LDS
#pragma unroll
  for (int s = 0; s < U; ++s) slab[U * lane + s] = r[s];
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < U; ++s) r[s] = slab[lane + 64 * s];
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < U; ++s) st16_nt(r[s], dst + lane + 64 * s);
#else
#pragma unroll
  for (int s = 0; s < U; ++s) st16_nt(r[s], dst + U * lane + s);
#endif
}
// four consecutive per-row scalars of the lane (scale, u, w, dscale ...)
template <typename T>
__device__ __forceinline__ void quad_load(const T* tile, int lane, T (&q)[4]) {
  const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(tile + 4 * lane);
  q[0] = v.v[0]; q[1] = v.v[1]; q[2] = v.v[2]; q[3] = v.v[3];
}
template <typename T>
__device__ __forceinline__ void quad_store(T* tile, int lane, const T (&q)[4]) {
  Vec4<T> v;
  v.v[0] = q[0]; v.v[1] = q[1]; v.v[2] = q[2]; v.v[3] = q[3];
  *reinterpret_cast<Vec4<T>*>(tile + 4 * lane) = v;
}

// one row of the forward pass, everything in registers
template <typename T, typename MP>
__device__ __forceinline__ void v3_forward_row(const T r[3], T k, T u, const T e[3], T z[3], T* w_out) {
  T l[3], t[3], v[3], n, p, s, L, x2;
  bool inside;
  v3_normalise_loc<T, MP>(r, l, &n);
  const T w = v3_w<T, MP>(u, k, &L, &x2);
  v3_tangent<T, MP>(e, l, t, v, &p, &s);
  const T f = v3_factor<T, MP>(w, &inside);
  z[0] = w * l[0] + f * v[0];
  z[1] = w * l[1] + f * v[1];
  z[2] = w * l[2] + f * v[2];
  *w_out = w;
}

// ---- (1) ROWS ----------------------------------------------------------------------------------------------------------
// VEC: every base pointer is 16-byte aligned, so full tiles take the staged path; partial tiles (the last one) and
// unaligned calls walk their rows with scalar accesses.
template <typename T, typename MP, bool BWD>
__global__ __launch_bounds__(256) void vmf3_rows_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                        const T* __restrict__ u, const T* __restrict__ eps,
                                                        const T* __restrict__ gz, const T* __restrict__ gkappa,
                                                        T* __restrict__ out3, T* __restrict__ out1,
                                                        T* __restrict__ kappa_out, int64_t n, int vec, int softplus,
                                                        PhiloxArg ka) {
  // out3 / out1: forward z (n, 3) and w (n) (nullable); backward dloc (n, 3) and dscale (n)
  __shared__ u32x4 s_slab[4][64 * v3_units<T>()];
  const PhiloxKey key = philox_resolve(ka);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * V3_TILE_ROWS;
  if (row0 >= n) return;  // wave-uniform; the kernel has no workgroup barrier
  u32x4* const slab = s_slab[wave];
  if (vec && row0 + V3_TILE_ROWS <= n) {
    constexpr int U = v3_units<T>();
    u32x4 fl[U], fe[U], fg[U];
    T r[12], e[12], g[12], k[4], uu[4], gk[4], o3[12], o1[4], kap[4];
    tile_fetch<T>(loc + 3 * row0, lane, fl);
    if (eps) tile_fetch<T>(eps + 3 * row0, lane, fe);
    if (BWD) tile_fetch<T>(gz + 3 * row0, lane, fg);
    quad_load<T>(scale + row0, lane, k);
    if (u) quad_load<T>(u + row0, lane, uu);
    if (BWD && gkappa) quad_load<T>(gkappa + row0, lane, gk);
    tile_deal_in<T>(slab, lane, fl, r);
    if (eps) tile_deal_in<T>(slab, lane, fe, e);
    if (BWD) tile_deal_in<T>(slab, lane, fg, g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t row = row0 + 4 * lane + q;
      T sig = (T)1;
      kap[q] = softplus ? v3_softplus_kappa<T, MP>(k[q], &sig) : k[q];
      T uq = uu[q];
      if (!u || !eps) {  // in-kernel noise: one Philox call per row serves u and the eps row
        const U4 pr = philox4x32_10(key.i0 + (uint64_t)row, 0u, key);
        if (!u) uq = u01<T>(pr.x);
        if (!eps) v3_eps_of<T>(pr, &e[3 * q]);
      }
      if (!BWD) {
        v3_forward_row<T, MP>(&r[3 * q], kap[q], uq, &e[3 * q], &o3[3 * q], &o1[q]);
      } else {
        T dk;
        v3_backward_row<T, MP>(&r[3 * q], kap[q], uq, &e[3 * q], &g[3 * q], &o3[3 * q], &dk);
        if (gkappa) dk += gk[q];
        o1[q] = dk * sig;
      }
    }
    tile_store<T>(out3 + 3 * row0, slab, lane, o3);
    if (out1) quad_store<T>(out1 + row0, lane, o1);
    if (!BWD && kappa_out) quad_store<T>(kappa_out + row0, lane, kap);
    return;
  }
  for (int q = 0; q < 4; ++q) {
    const int64_t row = row0 + lane + 64 * q;
    if (row >= n) break;
    const T rr[3] = {loc[3 * row], loc[3 * row + 1], loc[3 * row + 2]};
    T sig = (T)1;
    const T kq = softplus ? v3_softplus_kappa<T, MP>(scale[row], &sig) : scale[row];
    T uq, ee[3], o[3], s1;
    v3_draw_row<T>(u, eps, row, key, &uq, ee);
    if (!BWD) {
      v3_forward_row<T, MP>(rr, kq, uq, ee, o, &s1);
      if (kappa_out) kappa_out[row] = kq;
    } else {
      const T gg[3] = {gz[3 * row], gz[3 * row + 1], gz[3 * row + 2]};
      v3_backward_row<T, MP>(rr, kq, uq, ee, gg, o, &s1);
      if (gkappa) s1 += gkappa[row];
      s1 *= sig;
    }
    out3[3 * row] = o[0]; out3[3 * row + 1] = o[1]; out3[3 * row + 2] = o[2];
    if (out1) out1[row] = s1;
  }
}

static bool v3_aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <typename T, typename MP, bool BWD>
static void launch_rows(const V3Call& c, int64_t n) {
  const int64_t tiles = (n + V3_TILE_ROWS - 1) / V3_TILE_ROWS;
  const int64_t blocks = (tiles + 3) / 4;
  const bool vec = v3_aligned16(c.loc) && v3_aligned16(c.scale) && v3_aligned16(c.u) && v3_aligned16(c.eps) &&
                   (BWD ? v3_aligned16(c.gz) && v3_aligned16(c.gkappa) && v3_aligned16(c.dloc) && v3_aligned16(c.dscale)
                        : v3_aligned16(c.z) && v3_aligned16(c.w_out) && v3_aligned16(c.kappa_out));
  hipLaunchKernelGGL((vmf3_rows_kernel<T, MP, BWD>), dim3((unsigned)blocks), dim3(256), 0, c.stream, (const T*)c.loc,
                     (const T*)c.scale, (const T*)c.u, (const T*)c.eps, (const T*)c.gz, (const T*)c.gkappa,
                     (T*)(BWD ? c.dloc : c.z), (T*)(BWD ? c.dscale : c.w_out), (T*)c.kappa_out, n, vec ? 1 : 0,
                     c.softplus, c.ka);
}

int vmf3_fwd_rows(const V3Call& c, int64_t n) {
  VMF_REQUIRE((n + V3_TILE_ROWS - 1) / V3_TILE_ROWS / 4 < 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows exceed the grid",
              (long long)n);
  if (c.dtype == VMF_F32) launch_rows<float, MFast, false>(c, n);
  else launch_rows<double, M<double>, false>(c, n);
  return VMF_OK;
}
int vmf3_bwd_rows(const V3Call& c, int64_t n) {
  VMF_REQUIRE((n + V3_TILE_ROWS - 1) / V3_TILE_ROWS / 4 < 0x7fffffff, VMF_ERR_UNSUPPORTED, "n=%lld rows exceed the grid",
              (long long)n);
  if (c.dtype == VMF_F32) launch_rows<float, MFast, true>(c, n);
  else launch_rows<double, M<double>, true>(c, n);
  return VMF_OK;
}

// ---- (2) OUTER ---------------------------------------------------------------------------------------------------------
constexpr int V3_OUTER_J = 256;            // loc rows per block: 768 flattened columns, 12 per lane
constexpr int V3_OUTER_E = 3 * V3_OUTER_J;
constexpr int V3_OUTER_MAX_RPW = 16;       // rows i per wave (4 waves per block)
// smallest grid worth keeping when a larger tile is possible.  The backward wants large tiles (its column partials shrink
// with the tile height: (4096, 4096, 3) ran 0.057 / 0.072 / 0.127 ms at 1024 / 2048 / 4096 blocks).  The forward only
// amortises a prologue over its tile and runs best with the smallest one, one row per wave: (4096, 4096, 3) 0.0412 /
// 0.0387 / 0.0359 / 0.0328 ms at 1024 / 4096 / 8192 / 16384 blocks, (12288, 12288, 3) 0.338 -> 0.310 ms -- hence a floor
// no problem reaches.
#ifndef VMF3_OUTER_MIN_BLOCKS_FWD
#define VMF3_OUTER_MIN_BLOCKS_FWD (1 << 30)
#endif
#ifndef VMF3_OUTER_MIN_BLOCKS_BWD
#define VMF3_OUTER_MIN_BLOCKS_BWD 1024
#endif

// Rows per wave: large tiles amortise the block's prologue (256 direction rows + its w's: 2 % of the work at 16 rows per
// wave) and shrink the backward's column partials; small problems take small tiles so that the grid still fills the chip.
static int v3_outer_rpw(int64_t n_i, int64_t n_j, int64_t min_blocks) {
  const int64_t ntj = (n_j + V3_OUTER_J - 1) / V3_OUTER_J;
  for (int rpw = V3_OUTER_MAX_RPW; rpw > 1; rpw >>= 1) {
    const int64_t nti = (n_i + 4 * rpw - 1) / (4 * rpw);
    if (nti * ntj >= min_blocks) return rpw;
  }
  return 1;
}

// prologue shared by the forward and backward kernels: flattened l and v of the block's loc rows, (w, f) of its i rows
template <typename T, typename MP>
__device__ __forceinline__ void outer_prologue(const T* __restrict__ loc, const T* __restrict__ scale,
                                               const T* __restrict__ u, const T* __restrict__ eps, int64_t n_i,
                                               int64_t n_j, int64_t i0, int64_t j0, int rows_i, int softplus,
                                               PhiloxKey key, T* sL, T* sV, T* sW, T* sF, T* w_out, T* kappa_out) {
  const int t = threadIdx.x;
  {
    const int64_t j = j0 + t;
    T l[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    if (j < n_j) {
      const T r[3] = {loc[3 * j], loc[3 * j + 1], loc[3 * j + 2]};
      T e[3], tt[3], n, p, s;
      v3_draw_eps<T>(eps, j, key, e);
      v3_normalise_loc<T, MP>(r, l, &n);
      v3_tangent<T, MP>(e, l, tt, v, &p, &s);
    }
    sL[3 * t] = l[0]; sL[3 * t + 1] = l[1]; sL[3 * t + 2] = l[2];
    sV[3 * t] = v[0]; sV[3 * t + 1] = v[1]; sV[3 * t + 2] = v[2];
  }
  if (t < rows_i) {
    const int64_t i = i0 + t;
    T w = 0, f = 0;
    if (i < n_i) {
      T sig, L, x2;
      bool inside;
      const T k = softplus ? v3_softplus_kappa<T, MP>(scale[i], &sig) : scale[i];
      w = v3_w<T, MP>(v3_draw_u<T>(u, i, key), k, &L, &x2);
      f = v3_factor<T, MP>(w, &inside);
      if (w_out) w_out[i] = w;          // only the blocks of the first column tile pass these
      if (kappa_out) kappa_out[i] = k;
    }
    sW[t] = w;
    sF[t] = f;
  }
  __syncthreads();
}

template <typename T, typename MP, bool VEC>
__global__ __launch_bounds__(256) void vmf3_outer_fwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                             const T* __restrict__ u, const T* __restrict__ eps,
                                                             T* __restrict__ z, T* __restrict__ w_out,
                                                             T* __restrict__ kappa_out, int64_t n_i, int64_t n_j,
                                                             int ntj, int rpw, int softplus, PhiloxArg ka) {
  constexpr int U = v3_units<T>(), EU = v3_unit_elems<T>();
  __shared__ __align__(16) T sL[V3_OUTER_E], sV[V3_OUTER_E];
  __shared__ T sW[4 * V3_OUTER_MAX_RPW], sF[4 * V3_OUTER_MAX_RPW];
  const PhiloxKey key = philox_resolve(ka);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tj = blockIdx.x % ntj;
  const int64_t ti = blockIdx.x / ntj;
  const int64_t i0 = ti * 4 * rpw, j0 = (int64_t)tj * V3_OUTER_J;
  outer_prologue<T, MP>(loc, scale, u, eps, n_i, n_j, i0, j0, 4 * rpw, softplus, key, sL, sV, sW, sF,
                        tj == 0 ? w_out : nullptr, tj == 0 ? kappa_out : nullptr);
  const int64_t left_j = n_j - j0;
  const int ncol = 3 * (int)(left_j < V3_OUTER_J ? left_j : V3_OUTER_J);  // flattened columns of this tile
  T Lr[U * EU], Vr[U * EU];
#pragma unroll
  for (int s = 0; s < U; ++s) {
    const u32x4 a = reinterpret_cast<const u32x4*>(sL)[lane + 64 * s];
    const u32x4 b = reinterpret_cast<const u32x4*>(sV)[lane + 64 * s];
    __builtin_memcpy(&Lr[EU * s], &a, 16);
    __builtin_memcpy(&Vr[EU * s], &b, 16);
  }
  for (int r = 0; r < rpw; ++r) {
    const int li = wave * rpw + r;
    const int64_t i = i0 + li;
    if (i >= n_i) break;  // wave-uniform
    const T w = sW[li], f = sF[li];
    T* const zrow = z + (i * n_j + j0) * 3;
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int c0 = EU * (lane + 64 * s);
      T o[EU];
#pragma unroll
      for (int x = 0; x < EU; ++x) o[x] = w * Lr[EU * s + x] + f * Vr[EU * s + x];
      if (VEC) {  // 3 n_j is a multiple of the unit: a unit is inside the row or outside it
        if (c0 < ncol) {
          u32x4 pk;
          __builtin_memcpy(&pk, o, 16);
          st16_nt(pk, reinterpret_cast<u32x4*>(zrow + c0));
        }
      } else {
#pragma unroll
        for (int x = 0; x < EU; ++x)
          if (c0 + x < ncol) zrow[c0 + x] = o[x];
      }
    }
  }
}

// Backward, first launch: one pass over grad z.  rowp (ntj, n_i, 2): <G_i, L>, <G_i, V> over the block's columns;
// colp (nti, 3 n_j, 2): sum_i w_i G[i, e], sum_i f_i G[i, e] over the block's rows.
template <typename T, typename MP, bool VEC>
__global__ __launch_bounds__(256) void vmf3_outer_bwd_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                             const T* __restrict__ u, const T* __restrict__ eps,
                                                             const T* __restrict__ gz, T* __restrict__ rowp,
                                                             T* __restrict__ colp, int64_t n_i, int64_t n_j, int ntj,
                                                             int rpw, int softplus, PhiloxArg ka) {
  constexpr int U = v3_units<T>(), EU = v3_unit_elems<T>();
  __shared__ __align__(16) T sL[V3_OUTER_E], sV[V3_OUTER_E];
  __shared__ T sW[4 * V3_OUTER_MAX_RPW], sF[4 * V3_OUTER_MAX_RPW];
  __shared__ __align__(16) T sC[4][V3_OUTER_E];
  const PhiloxKey key = philox_resolve(ka);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tj = blockIdx.x % ntj;
  const int64_t ti = blockIdx.x / ntj;
  const int64_t i0 = ti * 4 * rpw, j0 = (int64_t)tj * V3_OUTER_J;
  outer_prologue<T, MP>(loc, scale, u, eps, n_i, n_j, i0, j0, 4 * rpw, softplus, key, sL, sV, sW, sF, (T*)nullptr,
                        (T*)nullptr);
  const int64_t left_j = n_j - j0;
  const int ncol = 3 * (int)(left_j < V3_OUTER_J ? left_j : V3_OUTER_J);
  T Lr[U * EU], Vr[U * EU], DL[U * EU], DV[U * EU];
#pragma unroll
  for (int s = 0; s < U; ++s) {
    const u32x4 a = reinterpret_cast<const u32x4*>(sL)[lane + 64 * s];
    const u32x4 b = reinterpret_cast<const u32x4*>(sV)[lane + 64 * s];
    __builtin_memcpy(&Lr[EU * s], &a, 16);
    __builtin_memcpy(&Vr[EU * s], &b, 16);
  }
#pragma unroll
  for (int x = 0; x < U * EU; ++x) DL[x] = DV[x] = (T)0;
  for (int r = 0; r < rpw; ++r) {
    const int li = wave * rpw + r;
    const int64_t i = i0 + li;
    if (i >= n_i) break;  // wave-uniform
    const T w = sW[li], f = sF[li];
    const T* const grow = gz + (i * n_j + j0) * 3;
    T g[U * EU];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int c0 = EU * (lane + 64 * s);
      if (VEC) {
        u32x4 pk = {0u, 0u, 0u, 0u};
        if (c0 < ncol) pk = ld16_nt(reinterpret_cast<const u32x4*>(grow + c0));
        __builtin_memcpy(&g[EU * s], &pk, 16);
      } else {
#pragma unroll
        for (int x = 0; x < EU; ++x) g[EU * s + x] = c0 + x < ncol ? grow[c0 + x] : (T)0;
      }
    }
    T a = 0, b = 0;
#pragma unroll
    for (int x = 0; x < U * EU; ++x) {
      a += g[x] * Lr[x];
      b += g[x] * Vr[x];
      DL[x] += w * g[x];
      DV[x] += f * g[x];
    }
    a = group_allreduce_sum<64, T>(a);
    b = group_allreduce_sum<64, T>(b);
    if (lane == 0) {
      Pair<T> pr;
      pr.a = a; pr.b = b;
      reinterpret_cast<Pair<T>*>(rowp)[(int64_t)tj * n_i + i] = pr;
    }
  }
  // column partials of the block: the four waves' sums added in wave order, one (dl, dv) pair per column
  Pair<T>* const cp = reinterpret_cast<Pair<T>*>(colp) + (ti * n_j + j0) * 3;  // (ti, 3 j0) of the (nti, 3 n_j, 2) array
  T dl_sum[3] = {0, 0, 0};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int s = 0; s < U; ++s) {
      u32x4 pk;
      __builtin_memcpy(&pk, pass == 0 ? &DL[EU * s] : &DV[EU * s], 16);
      reinterpret_cast<u32x4*>(sC[wave])[lane + 64 * s] = pk;
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const int e = threadIdx.x + 256 * x;
      const T sum = ((sC[0][e] + sC[1][e]) + sC[2][e]) + sC[3][e];
      if (pass == 0) {
        dl_sum[x] = sum;
      } else if (e < ncol) {
        Pair<T> pr;
        pr.a = dl_sum[x]; pr.b = sum;
        cp[e] = pr;
      }
    }
    __syncthreads();
  }
}

constexpr int V3_FIN_IDX = 32, V3_FIN_SUBS = 8;  // finish kernel: 256 threads = 32 indices x 8 lanes

// Backward, second launch: partials added in a fixed order, then the chain rules of sampler/von_mises_fisher.py:58-63
// (per i) and :30, :77-79 (per j).
template <typename T, typename MP>
__global__ __launch_bounds__(256) void vmf3_outer_bwd_finish_kernel(const T* __restrict__ loc,
                                                                    const T* __restrict__ scale,
                                                                    const T* __restrict__ u, const T* __restrict__ eps,
                                                                    const T* __restrict__ gkappa,
                                                                    const T* __restrict__ rowp,
                                                                    const T* __restrict__ colp, T* __restrict__ dloc,
                                                                    T* __restrict__ dscale, int64_t n_i, int64_t n_j,
                                                                    int ntj, int64_t nti, int softplus, PhiloxArg ka) {
  // 32 indices per block, eight lanes each: lane `sub` adds tiles sub, sub + 8, ... in rising order, the eight sums are
  // then added in lane order -- the same bits every run.  One lane per index left this launch on sixteen compute units
  // with a chain of dependent loads each.
  __shared__ T sP[V3_FIN_SUBS][V3_FIN_IDX][8];
  const PhiloxKey key = philox_resolve(ka);
  const int il = threadIdx.x & (V3_FIN_IDX - 1), sub = threadIdx.x / V3_FIN_IDX;
  const int64_t idx = (int64_t)blockIdx.x * V3_FIN_IDX + il;
  T acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // a, b | dl0, dv0, dl1, dv1, dl2, dv2
  if (idx < n_i) {
    for (int t0 = sub; t0 < ntj; t0 += 4 * V3_FIN_SUBS) {
      Pair<T> pr[4];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (t0 + x * V3_FIN_SUBS < ntj)
          pr[x] = reinterpret_cast<const Pair<T>*>(rowp)[(int64_t)(t0 + x * V3_FIN_SUBS) * n_i + idx];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (t0 + x * V3_FIN_SUBS < ntj) { acc[0] += pr[x].a; acc[1] += pr[x].b; }
    }
  }
  if (idx < n_j) {
    for (int64_t t0 = sub; t0 < nti; t0 += 4 * V3_FIN_SUBS) {
      Pair<T> pr[4][3];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (t0 + x * V3_FIN_SUBS < nti) {
          const Pair<T>* p = reinterpret_cast<const Pair<T>*>(colp) + ((t0 + x * V3_FIN_SUBS) * n_j + idx) * 3;
          pr[x][0] = p[0]; pr[x][1] = p[1]; pr[x][2] = p[2];
        }
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (t0 + x * V3_FIN_SUBS < nti) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { acc[2 + 2 * c] += pr[x][c].a; acc[3 + 2 * c] += pr[x][c].b; }
        }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) sP[sub][il][c] = acc[c];
  __syncthreads();
  if (sub == 0 && idx < n_i) {  // wave 0 finishes the rows, wave 1 the columns
    T a = 0, b = 0;
#pragma unroll
    for (int q = 0; q < V3_FIN_SUBS; ++q) { a += sP[q][il][0]; b += sP[q][il][1]; }
    T sig = (T)1;
    const T k = softplus ? v3_softplus_kappa<T, MP>(scale[idx], &sig) : scale[idx];
    T dk = v3_bwd_scale<T, MP>(k, v3_draw_u<T>(u, idx, key), a, b);
    if (gkappa) dk += gkappa[idx];
    dscale[idx] = dk * sig;
  }
  if (sub == 2 && idx < n_j) {
    T dl[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < V3_FIN_SUBS; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) { dl[c] += sP[q][il][2 + 2 * c]; dv[c] += sP[q][il][3 + 2 * c]; }
    const T r[3] = {loc[3 * idx], loc[3 * idx + 1], loc[3 * idx + 2]};
    T e[3], dr[3];
    v3_draw_eps<T>(eps, idx, key, e);
    v3_bwd_loc<T, MP>(r, e, dl, dv, dr);
    dloc[3 * idx] = dr[0]; dloc[3 * idx + 1] = dr[1]; dloc[3 * idx + 2] = dr[2];
  }
}

static bool outer_vec_ok(const void* p, int64_t n_j, int dtype) {
  const int64_t row_bytes = 3 * n_j * (dtype == VMF_F32 ? 4 : 8);
  return v3_aligned16(p) && row_bytes % 16 == 0;
}

static int outer_grid(int64_t n_i, int64_t n_j, int64_t min_blocks, int* ntj_out, int64_t* nti_out, int* rpw_out,
                      const char* who) {
  const int rpw = v3_outer_rpw(n_i, n_j, min_blocks);
  const int64_t ntj = (n_j + V3_OUTER_J - 1) / V3_OUTER_J, nti = (n_i + 4 * rpw - 1) / (4 * rpw);
  VMF_REQUIRE(ntj < (1 << 24) && nti * ntj < 0x7fffffff, VMF_ERR_UNSUPPORTED, "%s: (%lld, %lld, 3) exceeds the grid", who,
              (long long)n_i, (long long)n_j);
  *ntj_out = (int)ntj; *nti_out = nti; *rpw_out = rpw;
  return VMF_OK;
}

template <typename T, typename MP>
static void launch_outer_fwd(const V3Call& c, int64_t n_i, int64_t n_j, int ntj, int64_t nti, int rpw) {
  const unsigned grid = (unsigned)(nti * ntj);
#define V3_GO(VEC)                                                                                                       \
  hipLaunchKernelGGL((vmf3_outer_fwd_kernel<T, MP, VEC>), dim3(grid), dim3(256), 0, c.stream, (const T*)c.loc,           \
                     (const T*)c.scale, (const T*)c.u, (const T*)c.eps, (T*)c.z, (T*)c.w_out, (T*)c.kappa_out, n_i, n_j, \
                     ntj, rpw, c.softplus, c.ka)
  if (outer_vec_ok(c.z, n_j, c.dtype)) V3_GO(true);
  else V3_GO(false);
#undef V3_GO
}

int vmf3_fwd_outer(const V3Call& c, int64_t n_i, int64_t n_j) {
  int ntj, rpw;
  int64_t nti;
  int rc = outer_grid(n_i, n_j, VMF3_OUTER_MIN_BLOCKS_FWD, &ntj, &nti, &rpw, "vmf3_rsample_fwd");
  if (rc) return rc;
  if (c.dtype == VMF_F32) launch_outer_fwd<float, MFast>(c, n_i, n_j, ntj, nti, rpw);
  else launch_outer_fwd<double, M<double>>(c, n_i, n_j, ntj, nti, rpw);
  return VMF_OK;
}

int64_t vmf3_bwd_outer_workspace_bytes(int64_t n_i, int64_t n_j, int dtype) {
  const int rpw = v3_outer_rpw(n_i, n_j, VMF3_OUTER_MIN_BLOCKS_BWD);
  const int64_t ntj = (n_j + V3_OUTER_J - 1) / V3_OUTER_J, nti = (n_i + 4 * rpw - 1) / (4 * rpw);
  return (ntj * n_i * 2 + nti * n_j * 6) * (dtype == VMF_F32 ? 4 : 8);
}

template <typename T, typename MP>
static void launch_outer_bwd(const V3Call& c, int64_t n_i, int64_t n_j, int ntj, int64_t nti, int rpw, void* ws) {
  T* rowp = (T*)ws;
  T* colp = rowp + (int64_t)ntj * n_i * 2;
  const unsigned grid = (unsigned)(nti * ntj);
#define V3_GO(VEC)                                                                                                    \
  hipLaunchKernelGGL((vmf3_outer_bwd_kernel<T, MP, VEC>), dim3(grid), dim3(256), 0, c.stream, (const T*)c.loc,        \
                     (const T*)c.scale, (const T*)c.u, (const T*)c.eps, (const T*)c.gz, rowp, colp, n_i, n_j, ntj, rpw, \
                     c.softplus, c.ka)
  if (outer_vec_ok(c.gz, n_j, c.dtype)) V3_GO(true);
  else V3_GO(false);
#undef V3_GO
  const int64_t nmax = n_i > n_j ? n_i : n_j;
  hipLaunchKernelGGL((vmf3_outer_bwd_finish_kernel<T, MP>), dim3((unsigned)((nmax + V3_FIN_IDX - 1) / V3_FIN_IDX)), dim3(256), 0,
                     c.stream, (const T*)c.loc, (const T*)c.scale, (const T*)c.u, (const T*)c.eps, (const T*)c.gkappa, rowp, colp,
                     (T*)c.dloc, (T*)c.dscale, n_i, n_j, ntj, nti, c.softplus, c.ka);
}

int vmf3_bwd_outer(const V3Call& c, int64_t n_i, int64_t n_j, void* workspace, int64_t workspace_bytes) {
  int ntj, rpw;
  int64_t nti;
  int rc = outer_grid(n_i, n_j, VMF3_OUTER_MIN_BLOCKS_BWD, &ntj, &nti, &rpw, "vmf3_rsample_bwd_outer");
  if (rc) return rc;
  const int64_t need = vmf3_bwd_outer_workspace_bytes(n_i, n_j, c.dtype);
  VMF_REQUIRE(workspace && workspace_bytes >= need && v3_aligned16(workspace), VMF_ERR_BAD_ARG,
              "vmf3_rsample_bwd_outer: needs a 16-byte aligned workspace of %lld bytes, got %lld", (long long)need,
              (long long)workspace_bytes);
  if (c.dtype == VMF_F32) launch_outer_bwd<float, MFast>(c, n_i, n_j, ntj, nti, rpw, workspace);
  else launch_outer_bwd<double, M<double>>(c, n_i, n_j, ntj, nti, rpw, workspace);
  return VMF_OK;
}

// ---- (3) log_prob with the reference's sign, whole tiles -----------------------------------------------------------------
// The thread-per-row kernels read and write (n, 3) arrays as 4-byte elements 12 bytes apart from lane to lane -- six to
// fifteen such accesses per row, and the address unit's time for them is those kernels' time.  Here every (256, 3)
// operand moves as coalesced 16-byte accesses and is re-dealt through the wave's LDS slab (lane l owns rows 4 l ..
// 4 l + 3), per-row scalars as one 16-byte access per lane.  Same row functions, same bits.
template <typename T>
__global__ __launch_bounds__(256) void vmf3_log_prob_ref_tile_kernel(const T* __restrict__ loc,
                                                                     const T* __restrict__ scale,
                                                                     const T* __restrict__ x, T* __restrict__ logp,
                                                                     T* __restrict__ lognorm, int64_t ntiles) {
  constexpr int U = v3_units<T>();
  __shared__ u32x4 s_slab[4][64 * U];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= ntiles) return;  // wave-uniform; no workgroup barrier
  const int64_t row0 = t * V3_TILE_ROWS;
  u32x4* const slab = s_slab[wave];
  T k[4], lp[4], ln[4];
  quad_load<T>(scale + row0, lane, k);
  if (logp) {
    u32x4 fl[U], fx[U];
    T r[12], xv[12];
    tile_fetch<T>(loc + 3 * row0, lane, fl);
    tile_fetch<T>(x + 3 * row0, lane, fx);
    tile_deal_in<T>(slab, lane, fl, r);
    tile_deal_in<T>(slab, lane, fx, xv);
#pragma unroll
    for (int j = 0; j < 4; ++j) lp[j] = v3_log_prob_ref_row<T>(r + 3 * j, k[j], xv + 3 * j, &ln[j]);
    quad_store<T>(logp + row0, lane, lp);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v3_log_norm_ref<T>(k[j], &ln[j], (T*)nullptr);
  }
  if (lognorm) quad_store<T>(lognorm + row0, lane, ln);
}

template <typename T>
__global__ __launch_bounds__(256) void vmf3_log_prob_ref_bwd_tile_kernel(const T* __restrict__ loc,
                                                                         const T* __restrict__ scale,
                                                                         const T* __restrict__ x,
                                                                         const T* __restrict__ gl, T* __restrict__ dloc,
                                                                         T* __restrict__ dscale, T* __restrict__ dx,
                                                                         int64_t ntiles) {
  constexpr int U = v3_units<T>();
  __shared__ u32x4 s_slab[4][64 * U];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= ntiles) return;
  const int64_t row0 = t * V3_TILE_ROWS;
  u32x4* const slab = s_slab[wave];
  u32x4 fl[U], fx[U];
  T r[12], xv[12], k[4], g[4], dl[12], dxr[12], dk[4];
  tile_fetch<T>(loc + 3 * row0, lane, fl);
  tile_fetch<T>(x + 3 * row0, lane, fx);
  quad_load<T>(scale + row0, lane, k);
  quad_load<T>(gl + row0, lane, g);
  tile_deal_in<T>(slab, lane, fl, r);
  tile_deal_in<T>(slab, lane, fx, xv);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    v3_log_prob_ref_bwd_row<T>(r + 3 * j, k[j], xv + 3 * j, g[j], dl + 3 * j, &dk[j], dxr + 3 * j);
  if (dloc) tile_store<T>(dloc + 3 * row0, slab, lane, dl);
  if (dx) tile_store<T>(dx + 3 * row0, slab, lane, dxr);
  if (dscale) quad_store<T>(dscale + row0, lane, dk);
}

static bool lp_tiles_ok(int64_t n, std::initializer_list<const void*> ptrs) {
  if (n < V3_TILE_ROWS || (n / V3_TILE_ROWS + 3) / 4 >= 0x7fffffff) return false;
  for (const void* q : ptrs)
    if (!v3_aligned16(q)) return false;  // (a null pointer counts as aligned)
  return true;
}

int64_t vmf3_log_prob_ref_tiles(const void* loc, const void* scale, const void* x, void* logp, void* lognorm, int64_t n,
                                int dtype, hipStream_t stream) {
  if (!lp_tiles_ok(n, {loc, scale, x, logp, lognorm})) return 0;
  const int64_t ntiles = n / V3_TILE_ROWS;
  const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  if (dtype == VMF_F32)
    hipLaunchKernelGGL(vmf3_log_prob_ref_tile_kernel<float>, grid, block, 0, stream, (const float*)loc,
                       (const float*)scale, (const float*)x, (float*)logp, (float*)lognorm, ntiles);
  else
    hipLaunchKernelGGL(vmf3_log_prob_ref_tile_kernel<double>, grid, block, 0, stream, (const double*)loc,
                       (const double*)scale, (const double*)x, (double*)logp, (double*)lognorm, ntiles);
  return ntiles * V3_TILE_ROWS;
}

int64_t vmf3_log_prob_ref_bwd_tiles(const void* loc, const void* scale, const void* x, const void* gl, void* dloc,
                                    void* dscale, void* dx, int64_t n, int dtype, hipStream_t stream) {
  if (!lp_tiles_ok(n, {loc, scale, x, gl, dloc, dscale, dx})) return 0;
  const int64_t ntiles = n / V3_TILE_ROWS;
  const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  if (dtype == VMF_F32)
    hipLaunchKernelGGL(vmf3_log_prob_ref_bwd_tile_kernel<float>, grid, block, 0, stream, (const float*)loc,
                       (const float*)scale, (const float*)x, (const float*)gl, (float*)dloc, (float*)dscale, (float*)dx,
                       ntiles);
  else
    hipLaunchKernelGGL(vmf3_log_prob_ref_bwd_tile_kernel<double>, grid, block, 0, stream, (const double*)loc,
                       (const double*)scale, (const double*)x, (const double*)gl, (double*)dloc, (double*)dscale,
                       (double*)dx, ntiles);
  return ntiles * V3_TILE_ROWS;
}

// ---- (4) the mean direction, whole tiles ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vmf3_mean_tile_kernel(const T* __restrict__ loc, const T* __restrict__ scale,
                                                             T* __restrict__ mean, int64_t ntiles, BesselOrder o32,
                                                             BesselOrder o12) {
  constexpr int U = v3_units<T>();
  __shared__ u32x4 s_slab[4][64 * U];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= ntiles) return;  // wave-uniform; no workgroup barrier
  const int64_t row0 = t * V3_TILE_ROWS;
  u32x4* const slab = s_slab[wave];
  u32x4 fl[U];
  T r[12], k[4], o[12];
  tile_fetch<T>(loc + 3 * row0, lane, fl);
  quad_load<T>(scale + row0, lane, k);
  tile_deal_in<T>(slab, lane, fl, r);
#pragma unroll
  for (int j = 0; j < 4; ++j) v3_mean_row<T>(r + 3 * j, k[j], o32, o12, o + 3 * j);
  tile_store<T>(mean + 3 * row0, slab, lane, o);
}

int64_t vmf3_mean_tiles(const void* loc, const void* scale, void* mean, int64_t n, int dtype, hipStream_t stream) {
  if (!lp_tiles_ok(n, {loc, scale, mean})) return 0;
  const int64_t ntiles = n / V3_TILE_ROWS;
  const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  const BesselOrder o32 = make_bessel_order(1.5, 1), o12 = make_bessel_order(0.5, 1);
  if (dtype == VMF_F32)
    hipLaunchKernelGGL(vmf3_mean_tile_kernel<float>, grid, block, 0, stream, (const float*)loc, (const float*)scale,
                       (float*)mean, ntiles, o32, o12);
  else
    hipLaunchKernelGGL(vmf3_mean_tile_kernel<double>, grid, block, 0, stream, (const double*)loc, (const double*)scale,
                       (double*)mean, ntiles, o32, o12);
  return ntiles * V3_TILE_ROWS;
}
