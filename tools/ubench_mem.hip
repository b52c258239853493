// tools/ubench_mem.hip -- HBM floor of the NTT tile access patterns vs a plain streaming copy.
// 1 GiB buffer (in-place read+write => 2 GiB of traffic per launch), nontemporal accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// (0) plain copy, 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void k_copy16(u64x2* p, size_t n2) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    u64x2 v = __builtin_nontemporal_load(&p[i]); v.x += 1; __builtin_nontemporal_store(v, &p[i]);
  }
}
// (1) plain copy, 8 B per lane
__global__ __launch_bounds__(256) void k_copy8(u64* p, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    u64 v = __builtin_nontemporal_load(&p[i]); __builtin_nontemporal_store(v + 1, &p[i]);
  }
}
// (2) column-tile pattern: block = (limb, 16 columns); thread (col = tid&15, slow = tid>>4) touches rows slow+16k
__global__ __launch_bounds__(256) void k_col(u64* p) {
  const unsigned limb = blockIdx.x >> 4, cg = blockIdx.x & 15, col = threadIdx.x & 15, slow = threadIdx.x >> 4;
  u64* base = p + ((size_t)limb << 16) + cg * 16 + col + (slow << 8);
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(base + ((size_t)(16 * k) << 8));
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, base + ((size_t)(16 * k) << 8));
}
// (5) column-tile pattern, XCD-affine block order: all 16 column groups of a limb on one XCD
__global__ __launch_bounds__(256) void k_col_xcd(u64* p) {
  const unsigned b = blockIdx.x, limb = (b >> 7) * 8 + (b & 7), cg = (b >> 3) & 15, col = threadIdx.x & 15, slow = threadIdx.x >> 4;
  u64* base = p + ((size_t)limb << 16) + cg * 16 + col + (slow << 8);
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(base + ((size_t)(16 * k) << 8));
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, base + ((size_t)(16 * k) << 8));
}
// (6) 32-column tiles (256-byte segments): block = (limb, 8 groups x 2 row halves), thread (col = tid&31, slow = tid>>5)
__global__ __launch_bounds__(256) void k_col32(u64* p) {
  const unsigned limb = blockIdx.x >> 4, cg = (blockIdx.x >> 1) & 7, half = blockIdx.x & 1, col = threadIdx.x & 31, slow = threadIdx.x >> 5;
  u64* base = p + ((size_t)limb << 16) + cg * 32 + col + ((size_t)(half * 128 + slow) << 8);
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(base + ((size_t)(8 * k) << 8));
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, base + ((size_t)(8 * k) << 8));
}
// (7) 64-column tiles (512-byte segments): thread (col = tid&63, slow = tid>>6), 4 blocks per 64-col group
__global__ __launch_bounds__(256) void k_col64(u64* p) {
  const unsigned limb = blockIdx.x >> 4, cg = (blockIdx.x >> 2) & 3, quarter = blockIdx.x & 3, col = threadIdx.x & 63, slow = threadIdx.x >> 6;
  u64* base = p + ((size_t)limb << 16) + cg * 64 + col + ((size_t)(quarter * 64 + slow) << 8);
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(base + ((size_t)(4 * k) << 8));
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, base + ((size_t)(4 * k) << 8));
}
// (3) row-tile pattern: block = 16 rows of 16 consecutive limbs (same row), lane (pl = lane&15, line = ...) 128-B segments
__global__ __launch_bounds__(256) void k_row(u64* p) {
  const unsigned bundle = blockIdx.x >> 8, row = blockIdx.x & 255, pl = threadIdx.x & 15, line = threadIdx.x >> 4;
  u64* base = p + ((size_t)(bundle * 16 + line) << 16) + (row << 8) + pl;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(base + 16 * k);
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, base + 16 * k);
}
// (4) row-tile pattern with 16 B per lane: lane (pl8 = lane&7 -> 2 points, line), 8 loads
__global__ __launch_bounds__(256) void k_row16(u64x2* p) {
  const unsigned bundle = blockIdx.x >> 8, row = blockIdx.x & 255, pl = threadIdx.x & 7, line = (threadIdx.x >> 3) & 15, half = threadIdx.x >> 7;
  u64x2* base = p + (((size_t)(bundle * 16 + line) << 16) + (row << 8) + half * 128) / 2 + pl;
  u64x2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(base + 8 * k);
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k].x += 1; __builtin_nontemporal_store(v[k], base + 8 * k); }
}

int main() {
  const size_t limbs = 2048, n = limbs << 16;   // 1 GiB
  u64* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 8; which++) {
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipEventRecord(e0));
      switch (which) {
        case 0: hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, 0, (u64x2*)d, n / 2); break;
        case 1: hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(256), 0, 0, d, n); break;
        case 2: hipLaunchKernelGGL(k_col, dim3(limbs * 16), dim3(256), 0, 0, d); break;
        case 3: hipLaunchKernelGGL(k_row, dim3(limbs * 16), dim3(256), 0, 0, d); break;
        case 4: hipLaunchKernelGGL(k_row16, dim3(limbs * 16), dim3(256), 0, 0, (u64x2*)d); break;
        case 5: hipLaunchKernelGGL(k_col_xcd, dim3(limbs * 16), dim3(256), 0, 0, d); break;
        case 6: hipLaunchKernelGGL(k_col32, dim3(limbs * 16), dim3(256), 0, 0, d); break;
        case 7: hipLaunchKernelGGL(k_col64, dim3(limbs * 16), dim3(256), 0, 0, d); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const char* names[] = {"copy 16B/lane grid-stride", "copy 8B/lane grid-stride", "column-tile pattern 8B", "row-tile pattern 8B", "row-tile pattern 16B", "column-tile, XCD-affine order", "column tile 32 cols (256 B)", "column tile 64 cols (512 B)"};
    printf("%-28s %8.1f us  %7.0f GB/s (read+write)\n", names[which], best * 1e3, 2.0 * n * 8 / (best * 1e-3) / 1e9);
  }
  return 0;
}
