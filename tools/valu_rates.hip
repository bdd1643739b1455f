// valu_rates.hip -- issue cost of the vector instructions the samplers are made of, on the GPU at hand.
//   hipcc -O2 --offload-arch=gfx950 -o valu_rates tools/valu_rates.hip && ./valu_rates
// Every kernel runs a loop of 8 independent chains of ONE instruction; a wave stamps the shader clock around the loop.
// With W waves resident per SIMD (grid = 256 CUs x W blocks of 4 waves) the SIMD issues W x N instructions in the
// stamped interval, so cycles per wave-instruction = median interval / (W x N).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

constexpr int ITERS = 512, CHAINS = 8;

#define CHAIN8(INS)                                                                                        \
  asm volatile(INS(0) "\n\t" INS(1) "\n\t" INS(2) "\n\t" INS(3) "\n\t" INS(4) "\n\t" INS(5) "\n\t" INS(6) \
               "\n\t" INS(7)                                                                               \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(x), "v"(y))
#define CHAIN8_64(INS)                                                                                     \
  asm volatile(INS(0) "\n\t" INS(1) "\n\t" INS(2) "\n\t" INS(3) "\n\t" INS(4) "\n\t" INS(5) "\n\t" INS(6) \
               "\n\t" INS(7)                                                                               \
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
               : "v"(x), "v"(y), "v"(dx))

#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8"
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8"
#define I_MOV(i) "v_mov_b32 %" #i ", %8"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8"
#define I_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 9"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define I_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define I_SIN(i) "v_sin_f32 %" #i ", %" #i
#define I_COS(i) "v_cos_f32 %" #i ", %" #i
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i
#define I_LOG(i) "v_log_f32 %" #i ", %" #i
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i
#define I_RSQ(i) "v_rsq_f32 %" #i ", %" #i
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i
#define I_CVTU(i) "v_cvt_f32_u32 %" #i ", %" #i
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %10"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %10"
#define I_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %10, %10"
#define I_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %10"
#define I_ADD64(i) "v_add_f64 %" #i ", %" #i ", %10"
#define I_RCP64(i) "v_rcp_f64 %" #i ", %" #i
#define I_RSQ64(i) "v_rsq_f64 %" #i ", %" #i
#define I_MADU64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i
#define I_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %10"
#define I_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", %8"

template <int WHICH>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* cycles, uint32_t* sink, uint32_t seed) {
  uint32_t a[CHAINS];
  uint64_t d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    a[i] = seed * (threadIdx.x + 1) + i * 77u + 0x3f000000u;
    d[i] = ((uint64_t)(0x3ff00000u + i) << 32) | (seed + threadIdx.x);
  }
  const uint32_t x = 0x3f7ff000u + seed, y = 0x3a000000u + seed;
  const uint64_t dx = 0x3feff00000000000ull + seed;
  __builtin_amdgcn_s_barrier();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    if (WHICH == 0) CHAIN8(I_FMA);
    if (WHICH == 1) CHAIN8(I_MUL);
    if (WHICH == 2) CHAIN8(I_ADD);
    if (WHICH == 3) CHAIN8(I_MOV);
    if (WHICH == 4) CHAIN8(I_CND);
    if (WHICH == 5) CHAIN8(I_ADDU);
    if (WHICH == 6) CHAIN8(I_XOR);
    if (WHICH == 7) CHAIN8(I_BITOP3);
    if (WHICH == 8) CHAIN8(I_ALIGN);
    if (WHICH == 9) CHAIN8(I_MULLO);
    if (WHICH == 10) CHAIN8(I_MULHI);
    if (WHICH == 11) CHAIN8(I_MUL24);
    if (WHICH == 12) CHAIN8(I_MULHI24);
    if (WHICH == 13) CHAIN8(I_MAD24);
    if (WHICH == 14) CHAIN8(I_SIN);
    if (WHICH == 15) CHAIN8(I_COS);
    if (WHICH == 16) CHAIN8(I_EXP);
    if (WHICH == 17) CHAIN8(I_LOG);
    if (WHICH == 18) CHAIN8(I_RCP);
    if (WHICH == 19) CHAIN8(I_RSQ);
    if (WHICH == 20) CHAIN8(I_SQRT);
    if (WHICH == 21) CHAIN8(I_CVTU);
    if (WHICH == 22) CHAIN8_64(I_PKFMA);
    if (WHICH == 23) CHAIN8_64(I_PKMUL);
    if (WHICH == 24) CHAIN8_64(I_FMA64);
    if (WHICH == 25) CHAIN8_64(I_MUL64);
    if (WHICH == 26) CHAIN8_64(I_ADD64);
    if (WHICH == 27) CHAIN8_64(I_RCP64);
    if (WHICH == 28) CHAIN8_64(I_RSQ64);
    if (WHICH == 29) CHAIN8_64(I_MADU64);
    if (WHICH == 30) CHAIN8_64(I_LSHLADD64);
    if (WHICH == 31) CHAIN8_64(I_LDEXP64);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t acc = 0;
  for (int i = 0; i < CHAINS; ++i) acc ^= a[i] ^ (uint32_t)d[i] ^ (uint32_t)(d[i] >> 32);
  if (acc == 0x12345678u) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* NAMES[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_cndmask_b32", "v_add_u32", "v_xor_b32",
                              "v_bitop3_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
                              "v_mul_hi_u32_u24", "v_mad_u32_u24", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32",
                              "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_cvt_f32_u32", "v_pk_fma_f32", "v_pk_mul_f32",
                              "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_mad_u64_u32",
                              "v_lshl_add_u64", "v_ldexp_f64"};

template <int WHICH>
static void run(uint64_t* dcyc, uint32_t* dsink, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  std::vector<uint64_t> h(blocks * 4);
  for (int rep = 0; rep < 3; ++rep)
    hipLaunchKernelGGL((rate_kernel<WHICH>), dim3(blocks), dim3(256), 0, 0, dcyc, dsink, 3u + rep);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-18s W=%d  %.2f cycles per wave-instruction\n", NAMES[WHICH], waves_per_simd,
         med / ((double)waves_per_simd * ITERS * CHAINS));
}

template <int WHICH>
static void run_all(uint64_t* dcyc, uint32_t* dsink) {
  run<WHICH>(dcyc, dsink, 1);
  run<WHICH>(dcyc, dsink, 2);
  run<WHICH>(dcyc, dsink, 8);
  if constexpr (WHICH + 1 < 32) run_all<WHICH + 1>(dcyc, dsink);
}

int main() {
  uint64_t* dcyc;
  uint32_t* dsink;
  if (hipMalloc(&dcyc, 256 * 8 * 4 * 8) != hipSuccess || hipMalloc(&dsink, 64) != hipSuccess) return 1;
  run_all<0>(dcyc, dsink);
  return 0;
}
