// tools/ubench_icfuse.hip -- can the 512 KiB intermediate of a two-pass N = 2^16 transform be handed from
// the column pass to the row pass through the 256 MiB Infinity Cache inside ONE persistent launch?
// (Round-1's tools/ubench_l2fuse.hip tried the same through one XCD's 4 MiB L2 and lost: hand-off buffer and
// bytes in flight compete for the L2.  Here there is no XCD affinity and no capacity limit of that size.)
//
// Memory skeleton only (value + 1 per pass, every word checked in the kernels and after them).
// Persistent 256-thread workgroups draw tickets; the next ticket is drawn before the current tile is
// processed (its latency hides under the tile).  A single ticket counter serialises (measured: 54 ns per
// tile with one ticket counter and one arrival counter per 16 limbs = 7.2 ms per 4096 limbs), so the counter is
// sharded NS ways (workgroup b draws from shard b mod NS; ticket = shard + NS * count) and arrivals are
// counted per limb, every counter on a line of its own.
// Tickets are laid out in rounds of 2*T (T = GL*16 tiles, GL limbs per group): blocks of NS tickets alternate
// between column tiles of group r and row tiles of group r - D (ticket parity would pin a shard to one kind).  A row tile waits until its limb's 16 column tiles have signalled.
// Dependencies point to smaller tickets, every workgroup handles its tickets in increasing order and a
// ticket is owned only by a running workgroup => no deadlock whatever the residency; spins are bounded and
// a give-up flag ends the launch.
// Flavours (template):
//   IST  intermediate store : 0 = 8-B sc1, 1 = 16-B sc1 (lane pairs hold adjacent columns), 2 = 8-B plain (expected STALE)
//   ILD  intermediate load  : 0 = 8-B sc1, 1 = 16-B sc1
//   ring : the intermediate goes to a ring of (D+6)*GL limb slots instead of in place
//   defer: the column tile's completion is signalled inside the NEXT tile of that workgroup, behind the wait
//          for that tile's own loads (no store-drain stall)
// Baseline: the two patterns as two plain launches (what the library does today).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define AS1 __attribute__((address_space(1)))
typedef AS1 u64 *gptr;
typedef AS1 u32 *gu32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr u32 NS = 32;       // ticket shards
constexpr u32 CSTRIDE = 16;  // u32 words between arrival counters (64 B)

struct Ctl {                 // zeroed before every launch
  u32 ticket[NS][32];        // one 128-byte line per shard
  u32 giveup[32];            // [0]
  u32 stats[32];             // [0] spins on arrival counters, [1] wrong words (col loads), [2] wrong words (row loads), [3] tiles
  // followed by: done[nlimbs * CSTRIDE]
};

__device__ __forceinline__ u32 ld_sc1(gu32 p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld64_sc1(gptr p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st64_sc1(gptr p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte sc1 accesses through raw buffer instructions (compiler-tracked: waitcnt and store-data hazards;
// the first version used inline-asm global_store_dwordx4 and lost half its words to the >64-bit store-data hazard)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ u64 expect0(u64 idx) { return idx * 0x9E3779B97F4A7C15ull; }

template <int IST, int ILD, bool RING, bool DEFER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fused(u64 *data, u64 *ring, Ctl *ctl, u32 *done_, u32 nlimbs, u32 GL, u32 D, u32 ringslots, u32 pass) {
  __shared__ u32 sh[4];
  const u32 tid = threadIdx.x;
  const u32 shard = blockIdx.x % NS;
  gu32 ticket = (gu32)&ctl->ticket[shard][0];
  gu32 done = (gu32)done_;
  const u32 T = GL * 16, ngroups = nlimbs / GL, total = (ngroups + D) * 2 * T;
  u32 wrong_c = 0, wrong_r = 0, spins_total = 0, tiles = 0;
  u32 next = 0;
  if (tid == 0) next = shard + NS * __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32 pending = 0xffffffffu;           // DEFER: limb whose column tile of this workgroup has not been signalled yet
  auto signal = [&](u32 limb) { __hip_atomic_fetch_add(done + limb * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  for (;;) {
    if (tid == 0) {
      const u32 t = next;
      u32 kind = 2, limb = 0, tile = 0, ready = 1;      // kind: 0 col, 1 row, 2 skip, 3 exit
      if (t >= total) kind = 3;
      else {
        next = shard + NS * __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // blocks of NS tickets alternate column / row tiles, so every shard (= every workgroup) alternates too
        const u32 r = t / (2 * T), sub = t % (2 * T), idx = (sub / (2 * NS)) * NS + sub % NS;
        tile = idx & 15;
        if (((sub / NS) & 1) == 0) { limb = r * GL + (idx >> 4); kind = r < ngroups ? 0 : 2; }
        else if (r >= D) {
          limb = (r - D) * GL + (idx >> 4); kind = 1;
          ready = ld_sc1(done + limb * CSTRIDE) >= 16;
        }
      }
      sh[0] = kind; sh[1] = limb; sh[2] = tile; sh[3] = ready;
    }
    __syncthreads();
    u32 kind = sh[0];
    const u32 limb = __builtin_amdgcn_readfirstlane(sh[1]), tile = __builtin_amdgcn_readfirstlane(sh[2]), ready = sh[3];
    __syncthreads();
    auto flush_pending = [&](bool need_barrier) {
      if (DEFER && pending != 0xffffffffu) {
        if (need_barrier) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (tid == 0) signal(pending);
        pending = 0xffffffffu;
      }
    };
    if (kind >= 2 || !ready) flush_pending(true);     // never wait while holding back a signal of one's own
    if (!ready) {
      if (tid == 0) {
        u32 spins = 0;
        while (ld_sc1(done + limb * CSTRIDE) < 16) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 15) || ((spins & 63) == 0 && ld_sc1((gu32)&ctl->giveup[0]))) { atomicAdd(&ctl->giveup[0], 1u); sh[0] = 3; break; }
        }
        spins_total += spins;
      }
      __syncthreads();
      kind = sh[0];
      __syncthreads();
    }
    if (kind == 3) break;
    if (kind == 2) continue;
    tiles++;
    gptr base = (gptr)data + ((size_t)limb << 16);
    gptr mid = RING ? (gptr)ring + ((size_t)(limb % ringslots) << 16) : base;
    const u32 w = tid >> 6, l = tid & 63, rho = l >> 4, pos = l & 15;
    u64 v[16];
    if (kind == 0) {
      // column tile: thread (col = tid & 15, slow = tid >> 4) loads rows slow + 16k (8 B, nt)
      const u32 col = tid & 15, slow = tid >> 4;
      gptr p = base + tile * 16 + col;
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(p + ((size_t)(slow + 16 * k) << 8));
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const u64 i = ((u64)limb << 16) + ((slow + 16 * k) << 8) + tile * 16 + col;
        wrong_c += v[k] != expect0(i) + 2 * pass;
      }
      if (DEFER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();       // the real kernel exchanges through LDS here
      flush_pending(false);
      if (IST == 1) {
        // 16-B stores: lane (c2 = pos & 7, kh = 4w + 2(rho>>1) + (pos>>3)) writes columns 2c2, 2c2+1 of rows 16kh + 2jj + (rho&1)
        const u32 c2 = pos & 7, kh = 4 * w + 2 * (rho >> 1) + (pos >> 3);
        const __amdgpu_buffer_rsrc_t rs = rsrc_of((const u64 *)mid);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
          const u32 row = 16 * kh + 2 * jj + (rho & 1);
          const u64 i = ((u64)limb << 16) + (row << 8) + tile * 16 + 2 * c2;
          const u64 a = expect0(i) + 2 * pass + 1, b = expect0(i + 1) + 2 * pass + 1;
          u32x4 x = {(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)};
          __builtin_amdgcn_raw_buffer_store_b128(x, rs, ((row << 8) + tile * 16 + 2 * c2) * 8, 0, AUX_SC1);
        }
      } else {
        gptr pm = mid + tile * 16 + col;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const u64 i = ((u64)limb << 16) + ((16 * slow + j) << 8) + tile * 16 + col;
          if (IST == 0) st64_sc1(pm + ((size_t)(16 * slow + j) << 8), expect0(i) + 2 * pass + 1);
          else pm[(size_t)(16 * slow + j) << 8] = expect0(i) + 2 * pass + 1;
        }
      }
      if (DEFER) pending = limb;
      else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal(limb);
      }
    } else {
      // row tile: 16 rows of the limb, wave w takes 4 of them; lane (pl, lw) holds points pl + 16k of line lw
      // 16-B loads: lane (mu = pos & 7, parity = rho & 1, line 2(rho>>1) + (pos>>3)) loads points 2mu + 16(2k'+parity), +1
      // (the real kernel regroups them with v_permlane16_swap: one instruction per dword)
      const u32 mu = pos & 7, par = rho & 1;
      const u32 lw = ILD == 1 ? 2 * (rho >> 1) + (pos >> 3) : l >> 4, pl = ILD == 1 ? 2 * mu + par : l & 15;
      const u32 row = tile * 16 + 4 * w + lw;
      if (ILD == 1) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of((const u64 *)mid);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row << 8) + 2 * mu + 16 * (2 * k + par)) * 8, 0, AUX_SC1);
          v[2 * k] = ((u64)x.y << 32) | x.x;
          v[2 * k + 1] = ((u64)x.w << 32) | x.z;
        }
      } else {
        gptr pm = mid + (row << 8) + pl;
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = ld64_sc1(pm + 16 * k);
      }
      flush_pending(true);
      gptr q = base + (row << 8);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        // point held in v[k]
        const u32 pt = ILD == 1 ? 2 * mu + 16 * (2 * (k >> 1) + par) + (k & 1) : pl + 16 * k;
        const u64 i = ((u64)limb << 16) + (row << 8) + pt;
        wrong_r += v[k] != expect0(i) + 2 * pass + 1;
      }
      // final stores 8 B nt, 128-byte segments per 16 lanes (ILD == 1: after the swap lane (pl, lw) would hold
      // points pl + 16k; the skeleton writes the values it would hold)
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const u32 pt = pl + 16 * k;
        const u64 i = ((u64)limb << 16) + (row << 8) + pt;
        __builtin_nontemporal_store(ILD == 1 ? expect0(i) + 2 * pass + 2 : v[k] + 1, q + pt);
      }
    }
  }
  if (wrong_c) atomicAdd(&ctl->stats[1], wrong_c);
  if (wrong_r) atomicAdd(&ctl->stats[2], wrong_r);
  if (tid == 0) { if (spins_total) atomicAdd(&ctl->stats[0], spins_total); atomicAdd(&ctl->stats[3], tiles); }
}

// baseline: the two patterns as two launches (XCD-affine column order as in the library)
extern __shared__ u64 dyn_lds[];
__global__ __launch_bounds__(256) void k_col(u64 *data, u32 pass, u32 *wrong_) {
  if (pass == 0xffffffffu) dyn_lds[threadIdx.x] = 1;     // (keeps the dynamic LDS allocation alive)
  const u32 b = blockIdx.x, g = (b >> 7) * 8 + (b & 7), tile = (b >> 3) & 15, tid = threadIdx.x, col = tid & 15, slow = tid >> 4;
  gptr p = (gptr)data + ((size_t)g << 16) + tile * 16 + col;
  u64 v[16];
  u32 wrong = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(p + ((size_t)(slow + 16 * k) << 8));
#pragma unroll
  for (int k = 0; k < 16; k++) wrong += v[k] != expect0(((u64)g << 16) + ((slow + 16 * k) << 8) + tile * 16 + col) + 2 * pass;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; j++)
    __builtin_nontemporal_store(expect0(((u64)g << 16) + ((16 * slow + j) << 8) + tile * 16 + col) + 2 * pass + 1, p + ((size_t)(16 * slow + j) << 8));
  if (wrong) atomicAdd(wrong_, wrong);
}
__global__ __launch_bounds__(256) void k_row(u64 *data, u32 pass, u32 *wrong_) {
  if (pass == 0xffffffffu) dyn_lds[threadIdx.x] = 1;
  const u32 b = blockIdx.x, g = b >> 4, tile = b & 15, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, pl = lane & 15, lw = lane >> 4;
  const u32 row = tile * 16 + 4 * w + lw;
  gptr p = (gptr)data + ((size_t)g << 16) + (row << 8) + pl;
  u64 v[16];
  u32 wrong = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(p + 16 * k);
#pragma unroll
  for (int k = 0; k < 16; k++) wrong += v[k] != expect0(((u64)g << 16) + (row << 8) + pl + 16 * k) + 2 * pass + 1;
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k] + 1, p + 16 * k);
  if (wrong) atomicAdd(wrong_, wrong);
}
__global__ void k_init(u64 *data, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) data[i] = expect0(i);
}
__global__ void k_check(const u64 *data, size_t n, u64 add, u32 *wrong) {
  u32 w = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) w += data[i] != expect0(i) + add;
  if (w) atomicAdd(wrong, w);
}

typedef void (*fused_fn)(u64 *, u64 *, Ctl *, u32 *, u32, u32, u32, u32, u32);
struct Variant { const char *name; fused_fn fn; bool full; };

int main(int argc, char **argv) {
  const u32 nlimbs = argc > 1 ? atoi(argv[1]) : 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int only = argc > 3 ? atoi(argv[3]) : -1;      // run a single configuration index (for PMC runs)
  const u32 GL = 16;
  const size_t n = (size_t)nlimbs << 16;
  u64 *d; CK(hipMalloc(&d, n * 8));
  const u32 maxD = 16, ringslots_max = (maxD + 6) * GL;
  u64 *ring; CK(hipMalloc(&ring, (size_t)ringslots_max << 19));
  const size_t ctl_bytes = sizeof(Ctl) + (size_t)nlimbs * CSTRIDE * 4;
  char *ctl; CK(hipMalloc(&ctl, ctl_bytes));
  u32 *done = (u32 *)(ctl + sizeof(Ctl));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<char> h(sizeof(Ctl));
  u32 *wd; CK(hipMalloc(&wd, 4));
  auto final_wrong = [&](u32 passes) {
    CK(hipMemset(wd, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, n, (u64)2 * passes, wd);
    u32 hw; CK(hipMemcpy(&hw, wd, 4, hipMemcpyDeviceToHost));
    return hw;
  };
  int cfg = 0;
  for (u32 lds_bytes : {0u, 36000u}) {       // 36000 B of LDS per workgroup = the library kernels' occupancy (4 workgroups per CU)
  if (only < 0 || only == cfg) {
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, d, n);
    u32 *w; CK(hipMalloc(&w, 4)); CK(hipMemset(w, 0, 4));
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < reps; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_col, dim3(nlimbs * 16), dim3(256), lds_bytes, 0, d, (u32)rep, w);
      hipLaunchKernelGGL(k_row, dim3(nlimbs * 16), dim3(256), lds_bytes, 0, d, (u32)rep, w);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) { sum += ms; if (ms < best) best = ms; }
    }
    u32 hw; CK(hipMemcpy(&hw, w, 4, hipMemcpyDeviceToHost));
    printf("cfg %2d two launches (col, row) lds %5u                best %8.1f us  mean %8.1f us  %6.0f GB/s algorithmic  wrong(in-kernel) %u  wrong(final) %u\n", cfg, lds_bytes,
           best * 1e3, sum / (reps - 1) * 1e3, 2.0 * n * 8 / (best * 1e-3) / 1e9, hw, final_wrong(reps));
    fflush(stdout);
  }
  cfg++;
  }
  const Variant variants[] = {
      {"st8sc1 ld8sc1 inplace", k_fused<0, 0, false, false>, false},
      {"st16sc1 ld16sc1 inplace", k_fused<1, 1, false, false>, false},
      {"st16sc1 ld8sc1 inplace", k_fused<1, 0, false, false>, false},
      {"st8sc1 ld16sc1 inplace", k_fused<0, 1, false, false>, false},
      {"st16sc1 ld16sc1 inplace defer", k_fused<1, 1, false, true>, true},
      {"st16sc1 ld16sc1 ring defer", k_fused<1, 1, true, true>, false},
      {"st8sc1 ld8sc1 inplace defer", k_fused<0, 0, false, true>, true},
      {"st8PLAIN ld8sc1 inplace (stale?)", k_fused<2, 0, false, false>, false},
  };
  for (const Variant &va : variants)
    for (u32 wgs_per_cu : {3u, 4u})
      for (u32 D : {2u, 4u, 8u}) {
        if (!va.full && !(wgs_per_cu == 4 && D == 4)) continue;
        if (only >= 0 && only != cfg) { cfg++; continue; }
        hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, d, n);
        float best = 1e9, sum = 0;
        u32 wc = 0, wr = 0, gu = 0;
        const u32 ringslots = (D + 6) * GL;
        for (int rep = 0; rep < reps; rep++) {
          CK(hipMemsetAsync(ctl, 0, ctl_bytes, 0));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(va.fn, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, ring, (Ctl *)ctl, done, nlimbs, GL, D, ringslots, (u32)rep);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0) { sum += ms; if (ms < best) best = ms; }
          CK(hipMemcpy(h.data(), ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
          const Ctl *c = (const Ctl *)h.data();
          wc += c->stats[1]; wr += c->stats[2]; gu += c->giveup[0];
        }
        const Ctl *c = (const Ctl *)h.data();
        printf("cfg %2d fused %-34s wg/cu %u D %2u  best %8.1f us  mean %8.1f us  %6.0f GB/s algorithmic  wrong col/row loads %u/%u  wrong(final) %u  giveups %u  spins %u  tiles %u\n",
               cfg, va.name, wgs_per_cu, D, best * 1e3, sum / (reps - 1) * 1e3, 2.0 * n * 8 / (best * 1e-3) / 1e9, wc, wr, final_wrong(reps), gu, c->stats[0], c->stats[3]);
        fflush(stdout);
        cfg++;
      }
  return 0;
}
