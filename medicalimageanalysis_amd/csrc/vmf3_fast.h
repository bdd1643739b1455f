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

// noise of one row: u from Philox(index wi, slot 0), the eps row from Philox(index ei, slot 1)
template <typename T>
__device__ __forceinline__ T v3_draw_u(const T* u, int64_t wi, PhiloxKey key) {
  if (u) return u[wi];
  return u01<T>(philox4x32_10((uint64_t)wi, 0u, key).x);
}
template <typename T>
__device__ __forceinline__ void v3_philox_eps(int64_t ei, PhiloxKey key, T e[3]) {
  T n4[4];
  philox_normal4<T>((uint64_t)ei, 1u, key, n4);
  e[0] = n4[0]; e[1] = n4[1]; e[2] = n4[2];
}
template <typename T>
__device__ __forceinline__ void v3_draw_eps(const T* eps, int64_t ei, PhiloxKey key, T e[3]) {
  if (eps) {
    e[0] = eps[3 * ei]; e[1] = eps[3 * ei + 1]; e[2] = eps[3 * ei + 2];
    return;
  }
  v3_philox_eps<T>(ei, key, e);
}
