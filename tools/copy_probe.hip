// copy_probe.hip -- what a 1:1 read/write stream of the headline's size reaches on the GPU at hand, by access shape.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/copy_probe tools/copy_probe.hip && tools/bin/copy_probe
// The sampler kernels read a row tile and write a row tile of the same size; the yardstick for "memory-bound" is the
// best plain copy of the same bytes, and which shape of copy gets there says how a tile should move:
//   grid-stride (every access of the grid adjacent), wave tiles (a wave owns T contiguous KiB, all loads first, then the
//   stores), cache modifiers (plain / nt / nt + sc0 sc1 stores), workgroup size, XCD-aware tile order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MOD>
__device__ __forceinline__ f32x4 ld(const f32x4* p) {
  f32x4 v;
  if (MOD == 0) v = *p;
  else v = __builtin_nontemporal_load(p);
  return v;
}
template <int MOD>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if (MOD == 0) *p = v;
  else if (MOD == 1) __builtin_nontemporal_store(v, p);
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n) {
  // consecutive tiles on one XCD: blocks are dealt round-robin over 8 XCDs
  const uint32_t per = n / 8, rem = n % 8, x = b % 8, q = b / 8;
  return q + x * per + (x < rem ? x : rem);
}

// grid-stride: U accesses per thread per step, every access of the grid adjacent
template <int LM, int SM, int U>
__global__ __launch_bounds__(256) void copy_stride(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) v[u] = ld<LM>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) st<SM>(out + i + u * stride, v[u]);
  }
}

// wave tiles: a wave owns U KiB (U accesses of 1 KiB), all loads first, then all stores; one tile per wave
template <int LM, int SM, int U, int WAVES, bool XCD>
__global__ __launch_bounds__(WAVES * 64) void copy_tile(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
  const uint32_t b = XCD ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const size_t t0 = ((size_t)b * WAVES + (threadIdx.x >> 6)) * (size_t)(64 * U) + (threadIdx.x & 63);
  f32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (t0 + 64 * u < n4) v[u] = ld<LM>(in + t0 + 64 * u);
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (t0 + 64 * u < n4) st<SM>(out + t0 + 64 * u, v[u]);
}


// The sampler's software pipeline without its arithmetic: a wave owns 32 KiB = 32 steps of 1 KiB, D loads in flight
// (issued from inline asm, waits placed by hand: the memory counter retires loads and stores in issue order), one store
// per step.  OLD = the wait the sampler uses today (D = 1: vmcnt(0) between a step's arithmetic and its store, so the
// previous step's store is waited for as well); otherwise the wait sits at the top of the step and names the number of
// younger operations: min(t, D) stores + min(D - 1, 31 - t) loads.  FILL dependent multiply-adds per step and PRE before
// the first step stand in for the composition and the rejection phase.
__device__ __forceinline__ void issue(f32x4& dst, const f32x4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_n(f32x4& dst, int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst) : : "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(dst) : : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(dst) : : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(dst) : : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(dst) : : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(dst) : : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(dst) : : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" : "+v"(dst) : : "memory"); break;
  }
}
template <int D, int FILL, int PRE, bool OLD>
__global__ __launch_bounds__(256, 8) void copy_stream(const f32x4* __restrict__ in, f32x4* __restrict__ out, float a, float b) {
  constexpr int STEPS = 32;
  const uint32_t blk = xcd_block(blockIdx.x, gridDim.x);
  const size_t t0 = ((size_t)blk * 4 + (threadIdx.x >> 6)) * (size_t)(64 * STEPS) + (threadIdx.x & 63);
  f32x4 cur[D];
#pragma unroll
  for (int d = 0; d < D; ++d) issue(cur[d], in + t0 + 64 * d);
  float acc = a;
  for (int i = 0; i < PRE; ++i) acc = __builtin_fmaf(acc, a, b);
  if (OLD) wait_n(cur[0], 0);
#pragma unroll
  for (int t = 0; t < STEPS; ++t) {
    if (!OLD) wait_n(cur[t % D], (t < D ? t : D) + (D - 1 < STEPS - 1 - t ? D - 1 : STEPS - 1 - t));
    f32x4 v = cur[t % D];
    if (t + D < STEPS) issue(cur[t % D], in + t0 + 64 * (t + D));
    if (t == 0) v.x += acc * b;
    for (int i = 0; i < FILL / 4; ++i) {
      v.x = __builtin_fmaf(v.x, a, b); v.y = __builtin_fmaf(v.y, a, b);
      v.z = __builtin_fmaf(v.z, a, b); v.w = __builtin_fmaf(v.w, a, b);
    }
    if (OLD && t + 1 < STEPS) wait_n(cur[0], 0);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(out + t0 + 64 * t), "v"(v) : "memory");
  }
}

struct Result {
  const char* name;
  double best, med;
};

template <typename F>
static Result timeit(const char* name, F launch, size_t bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 30; ++i) launch();
  std::vector<float> ms;
  for (int i = 0; i < 30; ++i) {
    hipEventRecord(a, 0);
    launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float t;
    hipEventElapsedTime(&t, a, b);
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  Result r{name, 2.0 * bytes / ms[0] / 1e6, 2.0 * bytes / ms[ms.size() / 2] / 1e6};
  printf("%-44s best %7.0f GB/s  median %7.0f GB/s  (%.3f ms)\n", name, r.best, r.med, ms[ms.size() / 2]);
  fflush(stdout);
  return r;
}

int main() {
  const size_t bytes = (size_t)1 << 32;  // 4 GiB in, 4 GiB out: the headline's 2^24 rows x 256 B
  const size_t n4 = bytes / 16;
  f32x4 *in, *out;
  if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) return 1;
  hipMemset(in, 1, bytes);
  hipMemset(out, 0, bytes);
  hipDeviceSynchronize();
#define STREAM(D, FILL, PRE, OLD)                                                                          \
  timeit("stream D" #D " fill " #FILL " pre " #PRE " old-wait " #OLD, [&] {                                  \
    hipLaunchKernelGGL((copy_stream<D, FILL, PRE, OLD>), dim3((unsigned)(n4 / (64 * 32) / 4)), dim3(256), 0, 0, in, out, \
                       1.0f, 0.0f);                                                                          \
  }, bytes)
  if (getenv("PROBE_STREAM")) {
    STREAM(1, 0, 0, true);
    STREAM(1, 0, 0, false);
    STREAM(2, 0, 0, false);
    STREAM(3, 0, 0, false);
    STREAM(4, 0, 0, false);
    STREAM(1, 128, 700, true);
    STREAM(1, 128, 700, false);
    STREAM(2, 128, 700, false);
    STREAM(3, 128, 700, false);
    STREAM(4, 128, 700, false);
    STREAM(1, 256, 1400, true);
    STREAM(2, 256, 1400, false);
    STREAM(4, 256, 1400, false);
    return 0;
  }
  timeit("hipMemcpyAsync device-to-device", [&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
#define STRIDE(LM, SM, U, BLOCKS)                                                                        \
  timeit("stride ld" #LM " st" #SM " U" #U " blocks " #BLOCKS,                                             \
         [&] { hipLaunchKernelGGL((copy_stride<LM, SM, U>), dim3(BLOCKS), dim3(256), 0, 0, in, out, n4); }, bytes)
  STRIDE(0, 0, 4, 2048);
  STRIDE(1, 1, 4, 2048);
  STRIDE(1, 2, 4, 2048);
  STRIDE(1, 2, 4, 4096);
  STRIDE(1, 2, 4, 8192);
  STRIDE(1, 2, 8, 2048);
  STRIDE(1, 2, 8, 4096);
  STRIDE(1, 2, 2, 8192);
  STRIDE(1, 2, 1, 16384);
#define TILE(LM, SM, U, WAVES, XCD)                                                                         \
  timeit("tile ld" #LM " st" #SM " " #U " KiB/wave " #WAVES " waves xcd" #XCD, [&] {                          \
    const size_t tiles = (n4 + 64 * U - 1) / (64 * U);                                                       \
    hipLaunchKernelGGL((copy_tile<LM, SM, U, WAVES, XCD>), dim3((unsigned)((tiles + WAVES - 1) / WAVES)),     \
                       dim3(WAVES * 64), 0, 0, in, out, n4);                                                 \
  }, bytes)
  TILE(0, 0, 16, 4, false);
  TILE(1, 1, 16, 4, false);
  TILE(1, 2, 16, 4, false);
  TILE(1, 2, 16, 4, true);
  TILE(1, 2, 8, 4, true);
  TILE(1, 2, 4, 4, true);
  TILE(1, 2, 32, 4, true);
  TILE(1, 2, 16, 1, true);
  TILE(1, 2, 16, 2, true);
  TILE(1, 2, 16, 8, true);
  TILE(1, 2, 16, 16, true);
  TILE(1, 2, 8, 8, true);
  TILE(1, 2, 4, 16, true);
  return 0;
}
