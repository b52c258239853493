// ubench_place.hip -- where do the waves of a SMALL launch land?  The N = 2^16 row pass runs one WAVE per workgroup (64 threads); on 24 limbs that
// is 1536 workgroups for 1024 SIMDs.  If the dispatcher spread them evenly no SIMD would hold more than two; this records HW_ID / XCC_ID of every
// wave of (a) 1536 x 64-thread and (b) 384 x 256-thread workgroups (each wave spins ~5 us so that all are resident together) and prints the
// histogram of waves per SIMD and per CU.          hipcc -O2 --offload-arch=gfx950 tools/ubench_place.hip -o tools/ubench_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void place_kernel(u32 *out, int lds_words, long long spin_ticks) {
  extern __shared__ unsigned long long lds[];
  const u32 hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
  const u32 xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15;  // HW_REG_XCC_ID
  if (lds_words) lds[threadIdx.x % lds_words] = hw;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const u32 w = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    out[2 * w] = hw; out[2 * w + 1] = xcc;
  }
}

static void report(const char *name, const std::vector<u32> &h, int nwaves) {
  std::map<u32, int> per_simd, per_cu;
  for (int w = 0; w < nwaves; w++) {
    const u32 hw = h[2 * w], xcc = h[2 * w + 1];
    const u32 simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const u32 cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cu_key]++; per_simd[(cu_key << 2) | simd]++;
  }
  std::map<int, int> hs, hc;
  for (auto &kv : per_simd) hs[kv.second]++;
  for (auto &kv : per_cu) hc[kv.second]++;
  printf("%s: %d waves on %zu SIMDs of %zu CUs\n   waves per SIMD:", name, nwaves, per_simd.size(), per_cu.size());
  for (auto &kv : hs) printf("  %d x%d", kv.first, kv.second);
  printf("\n   waves per CU:  ");
  for (auto &kv : hc) printf("  %d x%d", kv.first, kv.second);
  printf("\n");
}

int main(int argc, char **argv) {
  const int limbs = argc > 1 ? atoi(argv[1]) : 24;
  const int nwaves = limbs * 64;
  u32 *d;
  CK(hipMalloc(&d, nwaves * 8));
  std::vector<u32> h(2 * nwaves);
  const long long spin = 500;   // 5 us at 100 MHz
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(d, 0, nwaves * 8));
    hipLaunchKernelGGL(place_kernel<64>, dim3(nwaves), dim3(64), 8960, 0, d, 1120, spin);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, nwaves * 8, hipMemcpyDeviceToHost));
    if (rep) report("64-thread workgroups, 8960 B LDS each", h, nwaves);
    CK(hipMemset(d, 0, nwaves * 8));
    hipLaunchKernelGGL(place_kernel<256>, dim3(nwaves / 4), dim3(256), 35840, 0, d, 4480, spin);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, nwaves * 8, hipMemcpyDeviceToHost));
    if (rep) report("256-thread workgroups, 35840 B LDS each", h, nwaves);
  }
  return 0;
}
