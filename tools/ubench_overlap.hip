// tools/ubench_overlap.hip -- what two HIP streams of ONE process buy on MI355X: a bandwidth-bound kernel beside a chain of
// small latency-bound kernels (the shape of a key-switch: the inner products with the key beside the inverse transforms of the
// special-prime accumulators), and what the event fork / join around them costs.  Round 4, before the engine's second lane.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_overlap.hip -o tools/ubench_overlap && tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// bandwidth-bound: in-place read + write of n2 16-byte words, one pass
__global__ __launch_bounds__(256) void k_stream(u64x2 *p, size_t n2) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    u64x2 v = __builtin_nontemporal_load(&p[i]); v.x += 1; __builtin_nontemporal_store(v, &p[i]);
  }
}
// latency-bound: few workgroups, each a serial chain of `rounds` x (dependent load, 64 multiply-adds, store)
__global__ __launch_bounds__(64) void k_chain(u64 *p, int rounds) {
  u64 *q = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
  u64 x = q[0];
  for (int r = 0; r < rounds; r++) {
    u64 y = q[(x & 7) * 64 + 64 * r % 2048];
#pragma unroll
    for (int k = 0; k < 64; k++) y = y * 0x9E3779B97F4A7C15ull + x;
    x = y;
  }
  q[0] = x;
}

static double med(std::vector<float> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t big = 1ull << 30;
  u64x2 *d_big; u64 *d_small;
  CK(hipMalloc(&d_big, big)); CK(hipMemset(d_big, 1, big));
  CK(hipMalloc(&d_small, 4096 * 8 * 4096)); CK(hipMemset(d_small, 0, 4096 * 8 * 4096));
  hipEvent_t e0, e1, ef, ej;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  const int reps = 41;
  // the big kernel: 256 MiB of traffic (~50 us), 3584 workgroups; the chain: 3 kernels of `wgs` workgroups
  for (int wgs : {128, 512, 1536}) {
    for (int rounds : {8, 24}) {
      auto A = [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(3584), dim3(256), 0, s, d_big, (size_t)(128u << 20) / 16); };
      auto B = [&](hipStream_t s) { for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, s, d_small, rounds); };
      std::vector<float> tA, tB, tAB, tPar, tDep;
      for (int r = 0; r < reps; r++) {
        float t;
        CK(hipEventRecord(e0, s1)); A(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tA.push_back(t * 1e3f);
        CK(hipEventRecord(e0, s1)); B(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tB.push_back(t * 1e3f);
        CK(hipEventRecord(e0, s1)); A(s1); B(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tAB.push_back(t * 1e3f);
        // fork: s2 waits for s1's position, runs the chain beside A; join: s1 waits for s2
        CK(hipEventRecord(e0, s1));
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
        A(s1); B(s2);
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tPar.push_back(t * 1e3f);
        // pure dependency through an event: A on s1, then B on s2 behind it, then back (what a hop between lanes costs)
        CK(hipEventRecord(e0, s1)); A(s1);
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); B(s2);
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tDep.push_back(t * 1e3f);
      }
      printf("chain 3 x %4d wgs x %2d rounds:  A %.1f us  B %.1f us  A;B one stream %.1f us  A || B two streams %.1f us  A -> B via events %.1f us\n",
             wgs, rounds, med(tA), med(tB), med(tAB), med(tPar), med(tDep));
    }
  }
  // host cost of the calls themselves
  {
    hipEvent_t ev[64];
    for (auto &evk : ev) CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int r = 0; r < 1000; r++) { CK(hipEventRecord(ev[r & 63], s1)); CK(hipStreamWaitEvent(s2, ev[r & 63], 0)); }
    clock_gettime(CLOCK_MONOTONIC, &b);
    CK(hipDeviceSynchronize());
    printf("host: hipEventRecord + hipStreamWaitEvent %.2f us per pair\n", ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / 1000.0 / 1e3);
  }
  return 0;
}
