// tools/ubench_nop.hip -- what the s_nop 0 cost that LLVM puts between inline-asm instructions on gfx950 (round 4).
// Every v_mad_u64_u32 / v_mul_hi_u32 of the butterflies is pinned by inline asm (modarith.hpp); on gfx940+ the hazard recogniser
// assumes a "dst forwarding" hazard between ANY inline asm and a following instruction that touches one of its registers
// (GCNHazardRecognizer: checkInlineAsmHazards / checkVALUHazards, hasDstSelForwardingHazard) and pads with s_nop 0: ~25 % of the
// instruction stream of every transform kernel.  The same butterflies written in plain C get no padding but ~11 % more VALU
// instructions (the compiler narrows the cross-term chain to v_mul_lo_u32 + v_add3_u32).  Which is faster, at which occupancy?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_nop.hip -o tools/ubench_nop && tools/ubench_nop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../aloha_amd/csrc/modarith.hpp"
using namespace aloha;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u64 cmad(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ void ct_c(u64 &X, u64 &Y, u64 w, u64 wp, u64 nq, u64 C) {
  u32 a0 = lo32(Y), a1 = hi32(Y), p0 = lo32(wp), p1 = hi32(wp);
  u32 c = __umulhi(a1, p0), d = __umulhi(a0, p1);
  u64 qh = cmad(a1, p1, (u64)c + d);
  u32 w0 = lo32(w), w1 = hi32(w), b0 = lo32(qh), b1 = hi32(qh), n0 = lo32(nq), n1 = hi32(nq);
  u64 t = cmad(a1, w0, 0); t = cmad(a0, w1, t); t = cmad(b1, n0, t); t = cmad(b0, n1, t);
  u64 acc = cmad(a0, w0, X); acc = cmad(b0, n0, acc);
  u64 o0 = ((u64)(hi32(acc) + lo32(t)) << 32) | lo32(acc);
  u64 o1 = (X << 1) + C - o0;
  X = o0; Y = o1;
}
template <bool ASM>
__global__ __launch_bounds__(64) void k(u64 *p, u64 w, u64 wp, u64 nq, u64 C, int iters) {
  u64 v[16];
  for (int i = 0; i < 16; i++) v[i] = p[(blockIdx.x * 64 + threadIdx.x) * 16 + i];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int half = 8 >> s;
#pragma unroll
      for (int g = 0; g < (1 << s); g++)
#pragma unroll
        for (int j = 0; j < half; j++) {
          u64 &X = v[g * 2 * half + j], &Y = v[g * 2 * half + j + half];
          if (ASM) ct_bfly<true, false>(X, Y, w + s + g, wp + s * 3 + g, nq, C);
          else ct_c(X, Y, w + s + g, wp + s * 3 + g, nq, C);
        }
    }
  }
  for (int i = 0; i < 16; i++) p[(blockIdx.x * 64 + threadIdx.x) * 16 + i] = v[i];
}
int main() {
  u64 *d; CK(hipMalloc(&d, 8192ull * 64 * 16 * 8)); CK(hipMemset(d, 1, 8192ull * 64 * 16 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  for (int wps : {1, 2, 4, 5, 8}) {
    const int grid = 1024 * wps;
    double t[2];
    for (int a = 0; a < 2; a++) {
      std::vector<float> ts;
      for (int r = 0; r < 9; r++) {
        CK(hipEventRecord(e0));
        if (a) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(64), 0, 0, d, 0x123456789abcdefull, 0xfedcba987654321ull, 0xf7ffffeeffffffffull, 0x2000004400000004ull, iters);
        else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(64), 0, 0, d, 0x123456789abcdefull, 0xfedcba987654321ull, 0xf7ffffeeffffffffull, 0x2000004400000004ull, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end()); t[a] = ts[4];
    }
    const double bf = 32.0 * iters;                      // butterflies per lane
    printf("%d waves/SIMD: plain C %.3f ms (%.1f cycles per butterfly per wave-slot at 2.1 GHz)  pinned asm %.3f ms (%.1f)\n", wps, t[0],
           t[0] * 1e-3 * 2.1e9 / bf / wps, t[1], t[1] * 1e-3 * 2.1e9 / bf / wps);
  }
  return 0;
}
