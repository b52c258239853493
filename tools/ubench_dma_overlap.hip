// tools/ubench_dma_overlap.hip -- would persistent waves with LDS-DMA double buffering overlap the row pass's memory
// traffic and its VALU work better than today's launch shape?  (the two-pass kernels stand at 1.13x of
// max(memory floor, VALU floor) per pass; an explicit prefetch pipeline is the one lever left inside that design.)
//
// Skeleton of the ROW pass only: a tile = 4 lines of 256 words (the same row of 4 limbs, 512 KiB apart), read once and
// written once, with `work` rounds of 16 independent 64-bit multiply-adds per lane and tile standing in for the
// butterflies (1 round = 16 v_mad_u64_u32; the real row tile is ~1300 VALU instructions per lane = ~80 rounds).
//   form A (today's shape): one wave per workgroup and tile, global -> VGPR loads (8 B per lane), one tile per wave,
//           occupancy from an LDS allocation of 8960 B (5 waves per SIMD);
//   form B: persistent waves (2 per SIMD), every wave loops over its tiles; tile t+1 travels global -> LDS by
//           global_load_lds_dwordx4 (16 B per lane, 8 instructions per tile) while tile t is read from LDS into
//           registers, worked on and stored; line pitch 272 words keeps the 64-bit LDS reads conflict-free.
// Output per `work`: microseconds per 4096 limbs for both forms, and the two floors (work alone, traffic alone).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
typedef AS1 u64 *gptr;

__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "s94", "s95");
  return r;
}
__device__ __forceinline__ void work_on(u64 (&v)[16], u32 rounds, u32 y) {
  for (u32 r = 0; r < rounds; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = mad32((u32)v[i], y, v[i]);
  }
}

// tile -> (limb group of 4, row): 4 lines = row `row` of limbs 4g .. 4g+3
__device__ __forceinline__ gptr line_ptr(u64 *data, u32 tile, u32 line) {
  const u32 g = tile >> 8, row = tile & 255;
  return (gptr)data + (((size_t)(4 * g + line)) << 16) + ((size_t)row << 8);
}

template <bool MEM, bool WORK>
__global__ __launch_bounds__(64) void k_plain(u64 *data, u32 rounds, u32 y) {
  extern __shared__ u64 pad[];
  if (y == 0xdeadbeefu) pad[threadIdx.x] = 1;                  // (keeps the occupancy-setting allocation alive)
  const u32 lane = threadIdx.x, pl = lane & 15, lw = lane >> 4, tile = blockIdx.x;
  gptr p = line_ptr(data, tile, lw) + pl;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = MEM ? __builtin_nontemporal_load(p + 16 * k) : (u64)(lane * 977u + k);
  if (WORK) work_on(v, rounds, y);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (MEM) __builtin_nontemporal_store(v[k] + 1, p + 16 * k);
    else if (v[k] == 0x123456789abcull) p[16 * k] = v[k];
  }
}

constexpr u32 LP = 272;                                          // LDS words per line (256 + 16: bank shift of 32 per line)
template <bool WORK>
__global__ __launch_bounds__(64) void k_dma(u64 *data, u32 ntiles, u32 rounds, u32 y) {
  __shared__ u64 buf[2][4 * LP];
  const u32 lane = threadIdx.x, pl = lane & 15, lw = lane >> 4;
  const u32 stride = gridDim.x;
  auto prefetch = [&](u32 tile, u32 b) {                         // 8 x 1 KiB: line = i >> 1, half = i & 1; lane -> 16 bytes
#pragma unroll
    for (int i = 0; i < 8; i++) {
      gptr src = line_ptr(data, tile, i >> 1) + (i & 1) * 128 + lane * 2;
      __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(&buf[b][(i >> 1) * LP + (i & 1) * 128]), 16, 0, 2);
    }
  };
  u32 tile = blockIdx.x, b = 0;
  if (tile < ntiles) prefetch(tile, 0);
  for (; tile < ntiles; tile += stride, b ^= 1) {
    const u32 next = tile + stride;
    if (next < ntiles) {
      prefetch(next, b ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // everything older than the 8 DMAs just issued has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = buf[b][lw * LP + pl + 16 * k];
    if (WORK) work_on(v, rounds, y);
    gptr p = line_ptr(data, tile, lw) + pl;
#pragma unroll
    for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, p + 16 * k);
    // (the stores of this tile count in vmcnt too: the wait above uses vmcnt(8) only when exactly the 8 DMAs are younger
    //  than what it needs -- the stores were issued before them)
  }
}

int main(int argc, char **argv) {
  const u32 nlimbs = argc > 1 ? atoi(argv[1]) : 4096;
  const int reps = 4;
  const size_t n = (size_t)nlimbs << 16;
  const u32 ntiles = nlimbs / 4 * 256;
  u64 *d; CK(hipMalloc(&d, n * 8));
  CK(hipMemset(d, 1, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch) {
    float best = 1e9;
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    return best * 1e3f;
  };
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const u32 cus = prop.multiProcessorCount;
  printf("row-pass skeleton, %u limbs (%u tiles of 8 KiB); times in us per step\n", nlimbs, ntiles);
  printf("%6s %12s %12s %12s %12s %14s %14s\n", "rounds", "plain", "plain:work", "plain:mem", "dma 2w/simd", "dma 3w/simd", "dma 4w/simd");
  for (u32 rounds : {0u, 20u, 40u, 60u, 80u, 100u}) {
    const float a = time_it([&] { hipLaunchKernelGGL((k_plain<true, true>), dim3(ntiles), dim3(64), 8960, 0, d, rounds, 12345u); });
    const float aw = time_it([&] { hipLaunchKernelGGL((k_plain<false, true>), dim3(ntiles), dim3(64), 8960, 0, d, rounds, 12345u); });
    const float am = time_it([&] { hipLaunchKernelGGL((k_plain<true, false>), dim3(ntiles), dim3(64), 8960, 0, d, rounds, 12345u); });
    float dm[3];
    int i = 0;
    for (u32 wps : {2u, 3u, 4u})
      dm[i++] = time_it([&] { hipLaunchKernelGGL((k_dma<true>), dim3(cus * 4 * wps), dim3(64), 0, 0, d, ntiles, rounds, 12345u); });
    printf("%6u %12.1f %12.1f %12.1f %12.1f %14.1f %14.1f\n", rounds, a, aw, am, dm[0], dm[1], dm[2]);
    fflush(stdout);
  }
  return 0;
}
