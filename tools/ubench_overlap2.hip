// tools/ubench_overlap2.hip -- second probe of two-stream execution on MI355X (round 4): do kernels of two streams run at the same
// time at all, what a hop between streams costs by event flavour, what a join costs when it is satisfied long before, and
// whether a high-priority stream lets a latency chain run THROUGH a grid that fills the chip.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_overlap2.hip -o tools/ubench_overlap2 && tools/ubench_overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_stream(u64x2 *p, size_t n2) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    u64x2 v = __builtin_nontemporal_load(&p[i]); v.x += 1; __builtin_nontemporal_store(v, &p[i]);
  }
}
// one tile per workgroup (no grid-stride loop): workgroups retire all the time, as in the transform kernels
__global__ __launch_bounds__(256) void k_tiles(u64x2 *p) {
  u64x2 *q = p + (size_t)blockIdx.x * 2048 + threadIdx.x;
  u64x2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(&q[256 * k]);
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k].x += 1; __builtin_nontemporal_store(v[k], &q[256 * k]); }
}
__global__ __launch_bounds__(64) void k_spin(u64 *p, int iters) {
  u64 x = p[blockIdx.x * 64 + threadIdx.x];
  for (int r = 0; r < iters; r++) {
#pragma unroll
    for (int k = 0; k < 64; k++) x = x * 0x9E3779B97F4A7C15ull + r;
  }
  p[blockIdx.x * 64 + threadIdx.x] = x;
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

static double med(std::vector<float> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: least %d greatest %d\n", lo, hi);
  hipStream_t s1, s2, sh;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, hi));
  u64x2 *d_big; u64 *d_small;
  CK(hipMalloc(&d_big, 1ull << 30)); CK(hipMemset(d_big, 1, 1ull << 30));
  CK(hipMalloc(&d_small, 4096 * 8 * 4096)); CK(hipMemset(d_small, 0, 4096 * 8 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 31;
  auto timed = [&](auto &&body) {
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) {
      float t;
      CK(hipEventRecord(e0, s1)); body(); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1));
      CK(hipDeviceSynchronize());
      ts.push_back(t * 1e3f);
    }
    return med(ts);
  };
  struct Flavour { const char *name; unsigned flags; } flavours[] = {
    {"default", 0}, {"disable-timing", hipEventDisableTiming}, {"disable-timing + no system fence", hipEventDisableTiming | hipEventDisableSystemFence}};
  // T1: basic concurrency -- two spin kernels of 256 workgroups x 64 threads
  {
    auto S = [&](hipStream_t s, int off) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s, d_small + off, 400); };
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const double one = timed([&] { S(s1, 0); });
    const double ser = timed([&] { S(s1, 0); S(s1, 1 << 20); });
    const double par = timed([&] { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); S(s1, 0); S(s2, 1 << 20); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); });
    printf("T1 spin kernels: one %.1f us, two in one stream %.1f us, two in two streams (fork + join) %.1f us\n", one, ser, par);
  }
  // T2: hop cost by event flavour: K on s1 -> K on s2 -> back, against K; K; on s1
  for (const Flavour &f : flavours) {
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, f.flags)); CK(hipEventCreateWithFlags(&ej, f.flags));
    auto K = [&](hipStream_t s) { hipLaunchKernelGGL(k_chain, dim3(128), dim3(64), 0, s, d_small, 8); };
    auto Big = [&](hipStream_t s) { hipLaunchKernelGGL(k_tiles, dim3(8192), dim3(256), 0, s, d_big); };   // 256 MiB written
    const double ser = timed([&] { K(s1); K(s1); K(s1); });
    const double hop = timed([&] { K(s1); CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); K(s2); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); K(s1); });
    const double serb = timed([&] { Big(s1); K(s1); K(s1); });
    const double hopb = timed([&] { Big(s1); CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); K(s2); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); K(s1); });
    printf("T2 [%s]: K;K;K one stream %.1f us, with two hops %.1f us (%.1f per hop); behind a 256 MiB writer: %.1f vs %.1f us (%.1f per hop)\n", f.name, ser, hop,
           (hop - ser) / 2, serb, hopb, (hopb - serb) / 2);
    // T3: a join that was satisfied long before: s2 runs K at the start, s1 runs Big, then waits for s2's event, then K
    const double j0 = timed([&] { Big(s1); K(s1); });
    const double j1 = timed([&] { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); K(s2); CK(hipEventRecord(ej, s2)); Big(s1); CK(hipStreamWaitEvent(s1, ej, 0)); K(s1); });
    printf("T3 [%s]: Big;K %.1f us; fork K aside at the start, Big, join (satisfied long before), K: %.1f us\n", f.name, j0, j1);
  }
  // T4: a chip-filling grid beside a latency chain of 3 kernels: plain second stream / high-priority stream; grid-stride or tiles
  {
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming | hipEventDisableSystemFence)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int form = 0; form < 2; form++) {
      auto A = [&](hipStream_t s) {
        if (form == 0) hipLaunchKernelGGL(k_stream, dim3(3584), dim3(256), 0, s, d_big, (size_t)(128u << 20) / 16);
        else hipLaunchKernelGGL(k_tiles, dim3(4096), dim3(256), 0, s, d_big);                           // 128 MiB in place
      };
      for (int wgs : {128, 1536}) {
        auto B = [&](hipStream_t s) { for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, s, d_small, 8); };
        const double a = timed([&] { A(s1); }), b = timed([&] { B(s1); }), ab = timed([&] { A(s1); B(s1); });
        const double p2 = timed([&] { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); A(s1); B(s2); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); });
        const double ph = timed([&] { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(sh, ef, 0)); A(s1); B(sh); CK(hipEventRecord(ej, sh)); CK(hipStreamWaitEvent(s1, ej, 0)); });
        const double pb = timed([&] { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(sh, ef, 0)); B(sh); A(s1); CK(hipEventRecord(ej, sh)); CK(hipStreamWaitEvent(s1, ej, 0)); });
        printf("T4 %s + chain 3 x %4d wgs: A %.1f  B %.1f  A;B %.1f | A || B plain stream %.1f  high-priority stream %.1f  high-priority, chain submitted first %.1f us\n",
               form ? "tiles      " : "grid-stride", wgs, a, b, ab, p2, ph, pb);
      }
    }
  }
  return 0;
}
