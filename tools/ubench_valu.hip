// tools/ubench_valu.hip -- issue cost of the individual VALU instructions the butterflies are made of
// (gfx950).  Every kernel runs 8 independent register chains per lane, 8 waves per SIMD, no memory
// traffic; reported: cycles per wave-instruction per SIMD at the device clock.  Pairs that talk
// through VCC (add_co/addc) are timed as a pair and reported per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 1024, UNR = 8;

// one asm statement per chain; X = 64-bit chain register pair, S = wave-uniform 64-bit constant
#define CHAIN8(STMT) STMT(x0) STMT(x1) STMT(x2) STMT(x3) STMT(x4) STMT(x5) STMT(x6) STMT(x7)

#define K(NAME, STMT)                                                                      \
  __global__ __launch_bounds__(256) void NAME(u64 *out, u64 seed, u64 sc) {                \
    u64 x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11,      \
        x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;                                          \
    u64 z = 0;                                                                             \
    asm volatile("" : "+v"(z));                                                            \
    for (int it = 0; it < ITERS; it++) {                                                   \
      _Pragma("unroll") for (int u = 0; u < UNR; u++) { CHAIN8(STMT) }                     \
    }                                                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ z;       \
  }

#define K32(NAME, STMT)                                                                    \
  __global__ __launch_bounds__(256) void NAME(u64 *out, u64 seed, u64 sc) {                \
    u32 x0 = (u32)seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, \
        x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;                                          \
    u32 y0 = x0 ^ 1, y1 = x1 ^ 2, y2 = x2 ^ 3, y3 = x3 ^ 4, y4 = x4 ^ 5, y5 = x5 ^ 6, y6 = x6 ^ 7, y7 = x7 ^ 8; \
    const u32 s = (u32)sc;                                                                 \
    for (int it = 0; it < ITERS; it++) {                                                   \
      _Pragma("unroll") for (int u = 0; u < UNR; u++) {                                    \
        STMT(x0, y0) STMT(x1, y1) STMT(x2, y2) STMT(x3, y3) STMT(x4, y4) STMT(x5, y5) STMT(x6, y6) STMT(x7, y7) } \
    }                                                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = (u64)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) << 32 | (y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7); \
  }

#define I_ADD(x, y)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "s"(s));
#define I_ADD3(x, y)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(s));
#define I_MOV(x, y)      asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(y));
#define I_XOR(x, y)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_CNDMASK(x, y)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
#define I_MULLO(x, y)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(s));
#define I_MULLO_V(x, y)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_MULHI(x, y)    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "s"(s));
#define I_MUL24(x, y)    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_MULHI24(x, y)  asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_MAD24(x, y)    asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(y));
#define I_ADDCO_PAIR(x, y) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(s) : "vcc");
#define I_SUBCO_PAIR(x, y) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(s) : "vcc");
#define I_ADDCO_S(x, y)  asm volatile("v_add_co_u32 %0, s[20:21], %0, %2\n\tv_addc_co_u32 %1, s[20:21], %1, %2, s[20:21]" : "+v"(x), "+v"(y) : "v"(s) : "s20", "s21");
#define I_PERM(x, y)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(s));
#define I_ALIGNBIT(x, y) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(y));
#define I_LSHLADD32(x, y) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(y));
#define I_DOT2(x, y)     asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(x) : "v"(y));
#define I_DOT4(x, y)     asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(y));
#define I_CMP_CND(x, y)  asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
#define I_MIN(x, y)      asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(y));
#define I_SUBREV_MIN(x, y) asm volatile("v_sub_u32 %1, %0, %2\n\tv_min_u32 %0, %0, %1" : "+v"(x), "+v"(y) : "s"(s));

K32(k_add, I_ADD) K32(k_add3, I_ADD3) K32(k_mov, I_MOV) K32(k_xor, I_XOR) K32(k_cndmask, I_CNDMASK)
K32(k_mullo, I_MULLO) K32(k_mullo_v, I_MULLO_V) K32(k_mulhi, I_MULHI) K32(k_mul24, I_MUL24) K32(k_mulhi24, I_MULHI24)
K32(k_mad24, I_MAD24) K32(k_addco, I_ADDCO_PAIR) K32(k_subco, I_SUBCO_PAIR) K32(k_addco_s, I_ADDCO_S)
K32(k_perm, I_PERM) K32(k_alignbit, I_ALIGNBIT) K32(k_lshladd32, I_LSHLADD32) K32(k_dot2, I_DOT2) K32(k_dot4, I_DOT4)
K32(k_cmpcnd, I_CMP_CND) K32(k_min, I_MIN) K32(k_submin, I_SUBREV_MIN)

#define J_LSHLADD64(x)   asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x) : "v"(z));
#define J_LSHLADD64_S(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "s"(sc));
#define J_MAD(x)         asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(x) : "v"((u32)z), "s"((u32)sc) : "s20", "s21");
#define J_MAD_VV(x)      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %1, %0" : "+v"(x) : "v"((u32)z) : "s20", "s21");
#define J_MAD_0(x)       asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "+v"(x) : "v"((u32)z), "s"((u32)sc) : "s20", "s21");
#define J_MADI(x)        asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(x) : "v"((u32)z), "s"((u32)sc) : "s20", "s21");
#define J_MOV64(x)       asm volatile("v_mov_b64 %0, %1" : "+v"(x) : "v"(z));
#define J_LSHR64(x)      asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(x));
#define J_LSHL64(x)      asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x));
#define J_FMA64(x)       asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(z));
#define J_MUL64F(x)      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(z));
#define J_PKFMA32(x)     asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(z));
#define J_PKMUL32(x)     asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(z));

K(k_lshladd64, J_LSHLADD64) K(k_lshladd64_s, J_LSHLADD64_S) K(k_mad, J_MAD) K(k_mad_vv, J_MAD_VV) K(k_mad_0, J_MAD_0) K(k_madi, J_MADI)
K(k_mov64, J_MOV64) K(k_lshr64, J_LSHR64) K(k_lshl64, J_LSHL64) K(k_fma64, J_FMA64) K(k_mul64f, J_MUL64F) K(k_pkfma32, J_PKFMA32) K(k_pkmul32, J_PKMUL32)

typedef void (*kern_t)(u64 *, u64, u64);
struct Row { const char *name; kern_t k; int insts; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clock %.2f GHz; 8 chains/lane, 8 waves/SIMD\n", prop.name, cus, ghz);
  const int blocks = cus * 8;
  u64 *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Row rows[] = {
    {"v_add_u32 (sgpr)", k_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_mov_b32", k_mov, 1}, {"v_xor_b32", k_xor, 1}, {"v_cndmask_b32", k_cndmask, 1},
    {"v_min_u32", k_min, 1}, {"v_sub_u32 + v_min_u32", k_submin, 2}, {"v_cmp_gt_u32 + v_cndmask", k_cmpcnd, 2},
    {"v_perm_b32", k_perm, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_lshl_add_u32", k_lshladd32, 1},
    {"v_mul_lo_u32 (sgpr)", k_mullo, 1}, {"v_mul_lo_u32 (vgpr)", k_mullo_v, 1}, {"v_mul_hi_u32 (sgpr)", k_mulhi, 1},
    {"v_mul_u32_u24", k_mul24, 1}, {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_mad_u32_u24", k_mad24, 1},
    {"v_dot2_u32_u16", k_dot2, 1}, {"v_dot4_u32_u8", k_dot4, 1},
    {"v_add_co + v_addc_co (vcc)", k_addco, 2}, {"v_sub_co + v_subb_co (vcc)", k_subco, 2}, {"v_add_co + v_addc_co (sgpr pair)", k_addco_s, 2},
    {"v_lshl_add_u64 (vgpr)", k_lshladd64, 1}, {"v_lshl_add_u64 (sgpr)", k_lshladd64_s, 1},
    {"v_mad_u64_u32 (v,s,acc)", k_mad, 1}, {"v_mad_u64_u32 (v,v,acc)", k_mad_vv, 1}, {"v_mad_u64_u32 (v,s,0)", k_mad_0, 1}, {"v_mad_i64_i32", k_madi, 1},
    {"v_mov_b64", k_mov64, 1}, {"v_lshrrev_b64", k_lshr64, 1}, {"v_lshlrev_b64", k_lshl64, 1},
    {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64f, 1}, {"v_pk_fma_f32", k_pkfma32, 1}, {"v_pk_mul_f32", k_pkmul32, 1},
  };
  for (auto &r : rows) {
    hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, d, 12345ull, 0x0123456789abcdefull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, d, 12345ull, 0x0123456789abcdefull);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double sec = ms / 1e3 / 5;
    const double wave_insts = (double)blocks * 4 * 8.0 * UNR * ITERS * r.insts;     // 4 waves per block
    const double per_simd_clk = wave_insts / sec / (cus * 4) / (ghz * 1e9);
    printf("%-34s %8.3f ms   %.2f cycles per wave-instruction per SIMD\n", r.name, sec * 1e3, 1.0 / per_simd_clk);
  }
  return 0;
}
