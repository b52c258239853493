// tools/ubench_intmul.hip -- issue-rate micro-benchmark for the integer multiplier on gfx950.
// Answers SURVEY 7 "hard part 1": is the 64-bit modular butterfly ALU-bound before HBM?
// Each kernel runs NCH independent dependency chains per lane for ITERS iterations with no
// memory traffic; reported: wave-instructions retired per cycle per CU (2.4 GHz assumed) and
// the derived cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../aloha_amd/csrc/modarith.hpp"
using namespace aloha;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NCH = 8;
constexpr int ITERS = 2048;

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u64 seed, u64 q, u64 w, u64 wp, u64 nq) {
  u64 x[NCH];
  for (int i = 0; i < NCH; i++) x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      if (OP == 0) x[i] = mad32_s(lo32(x[i]), lo32(w), x[i]);                       // v_mad_u64_u32
      if (OP == 1) x[i] = pack64(mullo32_s(lo32(x[i]), lo32(w)), hi32(x[i]));               // v_mul_lo_u32
      if (OP == 2) x[i] = pack64(mulhi32_s(lo32(x[i]), lo32(w)), hi32(x[i]));       // v_mul_hi_u32
      if (OP == 3) x[i] = pack64(__umul24(lo32(x[i]), lo32(w)), hi32(x[i]));      // v_mul_u32_u24
      if (OP == 4) x[i] = x[i] * w;                                               // 64-bit low mul
      if (OP == 5) x[i] = __umul64hi(x[i], wp) + 1;                               // 64-bit high mul
      if (OP == 6) x[i] = x[i] + w;                                               // 64-bit add (2 VALU)
      if (OP == 7) x[i] = mul_shoup_2q(x[i], w, wp, q);                           // exact Shoup
      if (OP == 8) x[i] = mul_shoup_4q<true>(x[i], w, wp, nq);                          // approx Shoup
      if (OP == 9) x[i] = pack64(lo32(x[i]) + lo32(w), hi32(x[i]));               // v_add_u32
    }
    if (OP == 10) {  // lazy CT butterfly pairs: (u,v) -> (u + v*w, u - v*w + 2q) with u < 4q kept by one csub(.,2q)
#pragma unroll
      for (int i = 0; i < NCH; i += 2) {
        u64 u = x[i], v = x[i + 1];
        u = u >= 2 * q ? u - 2 * q : u;
        u64 t = mul_shoup_2q(v, w, wp, q);
        x[i] = u + t;
        x[i + 1] = u - t + 2 * q;
      }
    }
    if (OP >= 12 && OP <= 17) {  // relaxed butterflies on pairs (x[i], x[i+1]); no conditionals
#pragma unroll
      for (int i = 0; i < NCH; i += 2) {
        if (OP == 12) ct_bfly<true, true>(x[i], x[i + 1], w, wp, nq, 2 * q);
        if (OP == 13) ct_bfly<true, false>(x[i], x[i + 1], w, wp, nq, 4 * q);
        if (OP == 14) ct_bfly_p32<true, true>(x[i], x[i + 1], w, wp, nq, 2 * q);
        if (OP == 15) ct_bfly_p32<true, false>(x[i], x[i + 1], w, wp, nq, 4 * q);
        if (OP == 16) ct_bfly<false, true>(x[i], x[i + 1], w ^ x[(i + 2) % NCH], wp, nq, 2 * q);
        if (OP == 17) { ct_bfly<true, true>(x[i], x[i + 1], w, wp, nq, 2 * q); if ((it & 3) == 3) x[i] = csub_borrow(x[i], 8 * q); }
      }
    }
    if (OP == 11) {  // same with the approximate quotient (t < 4q): values kept in [0, 8q)
#pragma unroll
      for (int i = 0; i < NCH; i += 2) {
        u64 u = x[i], v = x[i + 1];
        u = u >= 4 * q ? u - 4 * q : u;
        u64 t = mul_shoup_4q<true>(v, w, wp, nq);
        x[i] = u + t;
        x[i + 1] = u - t + 4 * q;
      }
    }
  }
  u64 acc = 0;
  for (int i = 0; i < NCH; i++) acc ^= x[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Row { const char* name; int op; double ops_per_iter; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, cus, ghz);
  const int blocks = cus * 8;
  u64* d;
  CK(hipMalloc(&d, (size_t)blocks * 256 * 8));
  const u64 q = 0x0800001100000001ull, w = 0x0123456789abcdefull % q;
  const u64 wp = (u64)((((unsigned __int128)w) << 64) / q), nq = 0 - q;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Row rows[] = {{"v_mad_u64_u32", 0, 1}, {"v_mul_lo_u32", 1, 1}, {"v_mul_hi_u32", 2, 1}, {"v_mul_u32_u24", 3, 1},
                {"mul64 lo", 4, 1}, {"mul64 hi (+1)", 5, 1}, {"add64", 6, 1}, {"shoup exact [0,2q)", 7, 1},
                {"shoup approx [0,4q)", 8, 1}, {"v_add_u32", 9, 1}, {"CT butterfly exact-q", 10, 0.5},
                {"CT butterfly approx-q", 11, 0.5},
                {"relaxed exact-q  C=2q", 12, 0.5}, {"relaxed approx-q C=4q", 13, 0.5},
                {"relaxed p32 exact", 14, 0.5}, {"relaxed p32 approx", 15, 0.5},
                {"relaxed exact, VGPR tw", 16, 0.5}, {"relaxed exact + corr/4", 17, 0.5}};
  for (auto& r : rows) {
    auto launch = [&]() {
      switch (r.op) {
#define C(n) case n: hipLaunchKernelGGL(k<n>, dim3(blocks), dim3(256), 0, 0, d, 12345ull, q, w, wp, nq); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
#undef C
      }
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; rep++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double sec = ms / 1e3 / 5;
    double lane_ops = (double)blocks * 256 * NCH * ITERS * r.ops_per_iter;   // per-lane ops
    double wave_ops = lane_ops / 64;
    double per_cu_per_clk = wave_ops / sec / cus / (ghz * 1e9);
    printf("%-24s %8.3f ms  %9.2f Gop/s(lane)  %.4f wave-op/clk/CU  => %.1f cyc per wave-op per SIMD\n", r.name,
           sec * 1e3, lane_ops / sec / 1e9, per_cu_per_clk, 4.0 / per_cu_per_clk);
  }
  // forward-NTT equivalent: N/2*log2 N = 524288 butterflies at N = 2^16
  return 0;
}
