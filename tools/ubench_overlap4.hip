// tools/ubench_overlap4.hip -- fourth probe (round 4): hipExtAnyOrderLaunch -- kernels of ONE stream launched without the AQL
// barrier bit, so that a kernel may start while its predecessors in the queue still run.  hip_ext.h says "not supported on
// GFX9xx"; measured here on gfx950.  Also: what a 1-wave "gate" kernel + a 1-thread "signal" kernel cost as a cross-stream
// dependency against hipEventRecord + hipStreamWaitEvent (10 us per hop, tools/ubench_overlap2.hip).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_spin(u64 *p, int iters) {
  u64 x = p[blockIdx.x * 64 + threadIdx.x];
  for (int r = 0; r < iters; r++) {
#pragma unroll
    for (int k = 0; k < 64; k++) x = x * 0x9E3779B97F4A7C15ull + r;
  }
  p[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_tiles(u64x2 *p) {
  u64x2 *q = p + (size_t)blockIdx.x * 2048 + threadIdx.x;
  u64x2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(&q[256 * k]);
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k].x += 1; __builtin_nontemporal_store(v[k], &q[256 * k]); }
}
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
// cross-stream dependency by hand: the producer's stream runs k_signal behind the producer, the consumer's stream runs k_gate
// in front of the consumer
__global__ void k_signal(unsigned *flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_gate(unsigned *flag, unsigned v, unsigned *status) {
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 200000000ll) { atomicAdd(status, 1u); break; }     // 2 s at 100 MHz: reported, not hung on
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
static double med(std::vector<float> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  u64 *d; CK(hipMalloc(&d, 256 << 20)); CK(hipMemset(d, 0, 256 << 20));
  u64x2 *d_big; CK(hipMalloc(&d_big, 1ull << 30)); CK(hipMemset(d_big, 1, 1ull << 30));
  unsigned *flags; CK(hipMalloc(&flags, 4096)); CK(hipMemset(flags, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto &&body) {
    std::vector<float> ts;
    for (int r = 0; r < 31; r++) {
      float t;
      CK(hipEventRecord(e0, s1)); body(); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1));
      CK(hipDeviceSynchronize());
      ts.push_back(t * 1e3f);
    }
    return med(ts);
  };
  auto spin = [&](hipStream_t s, int k, int wgs, int iters, unsigned fl) {
    hipExtLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s, nullptr, nullptr, fl, d + (size_t)k * (1 << 20), iters);
    CK(hipGetLastError());
  };
  for (int wgs : {256, 2048})
    for (int iters : {400, 4000}) {
      const double one = timed([&] { spin(s1, 0, wgs, iters, 0); });
      const double four = timed([&] { for (int k = 0; k < 4; k++) spin(s1, k, wgs, iters, 0); });
      const double any = timed([&] { spin(s1, 0, wgs, iters, 0); for (int k = 1; k < 4; k++) spin(s1, k, wgs, iters, hipExtAnyOrderLaunch); });
      printf("A1 spin %4d wgs x %4d iters: one %.1f us, four in order %.1f us, one + three any-order %.1f us\n", wgs, iters, one, four, any);
    }
  // a chip-filling streaming kernel with an any-order latency kernel behind it, then an ordered kernel (waits for both)
  {
    auto Big = [&](unsigned fl) { hipExtLaunchKernelGGL(k_tiles, dim3(4096), dim3(256), 0, s1, nullptr, nullptr, fl, d_big); };
    auto K = [&](unsigned fl, int wgs) { hipExtLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, s1, nullptr, nullptr, fl, d, 8); };
    for (int wgs : {128, 1536}) {
      const double a = timed([&] { Big(0); }), k = timed([&] { K(0, wgs); });
      const double ser = timed([&] { K(0, wgs); Big(0); K(0, wgs); });
      const double any1 = timed([&] { K(0, wgs); Big(hipExtAnyOrderLaunch); K(0, wgs); });       // Big beside the first K
      const double any2 = timed([&] { Big(0); K(hipExtAnyOrderLaunch, wgs); K(0, wgs); });        // K beside Big
      printf("A2 Big %.1f us, K(%d wgs) %.1f us: K;Big;K in order %.1f us | K;Big(any);K %.1f us | Big;K(any);K %.1f us\n", a, wgs, k, ser, any1, any2);
    }
  }
  // gate / signal kernels against events
  {
    unsigned seq = 0;
    auto K = [&](hipStream_t s) { hipLaunchKernelGGL(k_chain, dim3(128), dim3(64), 0, s, d, 8); };
    auto Big = [&](hipStream_t s) { hipLaunchKernelGGL(k_tiles, dim3(8192), dim3(256), 0, s, d_big); };
    const double ser = timed([&] { K(s1); K(s1); K(s1); });
    const double hop = timed([&] {
      K(s1); ++seq; hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s1, flags, seq);
      hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s2, flags, seq, flags + 64); K(s2);
      hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s2, flags + 32, seq);
      hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s1, flags + 32, seq, flags + 64); K(s1);
    });
    const double j0 = timed([&] { Big(s1); K(s1); });
    const double j1 = timed([&] {
      ++seq; K(s2); hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s2, flags + 32, seq);
      Big(s1); hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s1, flags + 32, seq, flags + 64); K(s1);
    });
    unsigned st = 0;
    CK(hipMemcpy(&st, flags + 64, 4, hipMemcpyDeviceToHost));
    printf("G1 gate/signal kernels: K;K;K one stream %.1f us, with two hops %.1f us (%.1f per hop); Big;K %.1f us, K aside + Big + satisfied gate + K %.1f us; gate timeouts %u\n",
           ser, hop, (hop - ser) / 2, j0, j1, st);
  }
  return 0;
}
