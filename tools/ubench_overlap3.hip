// tools/ubench_overlap3.hip -- third probe (round 4): two streams with NO events between them, host-timed -- is it the event
// fork that serialises the streams of tools/ubench_overlap2.hip, or the queues themselves?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(64) void k_spin(u64 *p, int iters) {
  u64 x = p[blockIdx.x * 64 + threadIdx.x];
  for (int r = 0; r < iters; r++) {
#pragma unroll
    for (int k = 0; k < 64; k++) x = x * 0x9E3779B97F4A7C15ull + r;
  }
  p[blockIdx.x * 64 + threadIdx.x] = x;
}
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
int main() {
  hipStream_t s[4];
  for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  u64 *d; CK(hipMalloc(&d, 64 << 20)); CK(hipMemset(d, 0, 64 << 20));
  for (int iters : {400, 4000, 40000}) {
    for (int wgs : {64, 256, 2048}) {
      auto run = [&](int nstreams, int nk) {
        std::vector<double> ts;
        for (int r = 0; r < 15; r++) {
          CK(hipDeviceSynchronize());
          const double t0 = now();
          for (int k = 0; k < nk; k++) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s[k % nstreams], d + (size_t)k * (1 << 20), iters);
          CK(hipDeviceSynchronize());
          ts.push_back(now() - t0);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
      };
      printf("spin %5d iters x %4d wgs: 1 kernel %.1f us | 4 kernels: one stream %.1f, two streams %.1f, four streams %.1f us (host-timed, incl. launch + sync)\n",
             iters, wgs, run(1, 1), run(1, 4), run(2, 4), run(4, 4));
    }
  }
  return 0;
}
