// vmf3_fast.h -- launchers of the tiled (row-per-row) and outer-product kernels of VonMisesFisher3D.rsample
// (vmf3_fast.hip), called by the C entry points in vmf3.hip once the broadcast pattern of a call is known.
#pragma once
#include "common.h"

struct V3Call {
  const void* loc; const void* scale; const void* u; const void* eps;  // u / eps null: in-kernel Philox
  void* z; void* w_out; void* kappa_out;                                // forward outputs (w_out / kappa_out nullable)
  const void* gz; const void* gkappa; void* dloc; void* dscale;         // backward (gkappa nullable)
  int softplus;                                                         // scale holds raw values: kappa = softplus(raw) + 1
  int dtype;
  PhiloxArg ka;
  hipStream_t stream;
};

// every output row has its own loc row, scale element, u and eps row: n rows
int vmf3_fwd_rows(const V3Call& c, int64_t n);
int vmf3_bwd_rows(const V3Call& c, int64_t n);
// z[i, j] = w_i l_j + f_i v_j: u / scale have n_i elements, loc / eps n_j rows, z is (n_i, n_j, 3)
int vmf3_fwd_outer(const V3Call& c, int64_t n_i, int64_t n_j);
int vmf3_bwd_outer(const V3Call& c, int64_t n_i, int64_t n_j, void* workspace, int64_t workspace_bytes);
int64_t vmf3_bwd_outer_workspace_bytes(int64_t n_i, int64_t n_j, int dtype);
// CNN_experiment/von_mises_fisher.py:88-108 on whole tiles of 256 rows (every pointer given 16-byte aligned): the
// number of leading rows taken; the caller's thread-per-row kernel takes the rest
int64_t vmf3_log_prob_ref_tiles(const void* loc, const void* scale, const void* x, void* logp, void* lognorm, int64_t n,
                                int dtype, hipStream_t stream);
int64_t vmf3_log_prob_ref_bwd_tiles(const void* loc, const void* scale, const void* x, const void* gl, void* dloc,
                                    void* dscale, void* dx, int64_t n, int dtype, hipStream_t stream);
// sampler/von_mises_fisher.py:42 (the mean direction) likewise
int64_t vmf3_mean_tiles(const void* loc, const void* scale, void* mean, int64_t n, int dtype, hipStream_t stream);

// Noise of index i comes from ONE call, Philox(i, slot 0) = (x, y, z, w) -- u01 reads the upper 23 bits of a word, so
// five uniforms fit into the four words without sharing a bit:
//   u    = u01(x)                                    the inverse-CDF uniform of element i of `u`
//   eps  = Box-Muller(u01(y), u01(z)) -> eps0, eps1;  Box-Muller(u01(w), u01(t)) -> eps2 (its cosine branch),
//          t = the lower 9 bits of x, y and z side by side (the sine branch of the second pair is not used)
// A row-per-row call (u index = eps index) therefore pays for one Philox call per row; the two-call layout it replaces
// (u from slot 0, eps from slot 1) made the kernel instruction-bound at half the memory rate.
__device__ __forceinline__ uint32_t v3_fifth_word(const U4& r) {
  return ((r.x & 0x1FFu) << 23) | ((r.y & 0x1FFu) << 14) | ((r.z & 0x1FFu) << 5);
}
template <typename T>
__device__ __forceinline__ void v3_eps_of(const U4& r, T e[3]) {
  T unused;
  box_muller<T>(u01<T>(r.y), u01<T>(r.z), &e[0], &e[1]);
  box_muller<T>(u01<T>(r.w), u01<T>(v3_fifth_word(r)), &e[2], &unused);
}
template <typename T>
__device__ __forceinline__ T v3_draw_u(const T* u, int64_t wi, PhiloxKey key) {
  if (u) return u[wi];
  return u01<T>(philox4x32_10(key.i0 + (uint64_t)wi, 0u, key).x);
}
template <typename T>
__device__ __forceinline__ void v3_philox_eps(int64_t ei, PhiloxKey key, T e[3]) {
  v3_eps_of<T>(philox4x32_10(key.i0 + (uint64_t)ei, 0u, key), e);
}
template <typename T>
__device__ __forceinline__ void v3_draw_eps(const T* eps, int64_t ei, PhiloxKey key, T e[3]) {
  if (eps) {
    e[0] = eps[3 * ei]; e[1] = eps[3 * ei + 1]; e[2] = eps[3 * ei + 2];
    return;
  }
  v3_philox_eps<T>(ei, key, e);
}
// u and the eps row of the same index: one call serves both
template <typename T>
__device__ __forceinline__ void v3_draw_row(const T* u, const T* eps, int64_t row, PhiloxKey key, T* uq, T e[3]) {
  if (u && eps) {
    *uq = u[row];
    e[0] = eps[3 * row]; e[1] = eps[3 * row + 1]; e[2] = eps[3 * row + 2];
    return;
  }
  const U4 r = philox4x32_10(key.i0 + (uint64_t)row, 0u, key);
  *uq = u ? u[row] : u01<T>(r.x);
  if (eps) {
    e[0] = eps[3 * row]; e[1] = eps[3 * row + 1]; e[2] = eps[3 * row + 2];
  } else {
    v3_eps_of<T>(r, e);
  }
}
