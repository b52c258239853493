// ntt_product.hip -- c = iNTT(NTT(a) .* NTT(b)) per limb at N = 2^14 as ONE launch (gfx950).
//
// BASELINE config 3 (RNS polynomial product, N = 2^14, 8 limbs): as instructions it is
//   vle a; vntt; vle b; vntt; vfqmul.vv; vintt; vse           (mul_plain.mem:4-19's shape with the transforms added)
// and, run loop by loop through the batched kernels, five launches of 5-8 us for 3 MiB of data: launch latency.  A limb of
// 128 KiB would fit one CU's LDS, but three transforms on ONE CU are 3 x 114688 butterflies x ~19 instructions / 64 lanes
// per cycle = ~49 us: a product of 8 limbs has to be spread over the chip, i.e. its transforms need an exchange between
// CUs.  Here the exchange goes through memory INSIDE the launch.  Element e = r * 1024 + x (r < 16: "row", x < 1024):
//   P1  stages 0..3 of both operands   thread t holds e = r * 1024 + t, r = 0..15 (one radix-16 register step, wave-uniform
//                                      twiddles, coalesced 8-byte lanes); a and b in different workgroups
//   P2  stages 4..13 inside a row      one WAVE per (limb, row, operand): radix-16 over x bits 9..6, over bits 5..2, radix-4
//                                      over bits 1..0, two wave-private LDS exchanges; the operand-b wave hands its row to the
//                                      operand-a wave through LDS (one workgroup barrier), which multiplies and runs the
//                                      inverse stages 13..4 the same way backwards
//   P3  inverse stages 3..0, N^-1      the mirror of P1, canonical results to the destination
// Workgroups of a later phase follow those of an earlier one in the grid and wait for their data.  The hand-off is
// DATA-TAGGED: every intermediate word is < 4q < 2^62 and carries the launch's 2-bit tag in its top bits, written by ONE
// 8-byte sc1 (write-through) store; a consumer loads the words it needs with sc1 loads and repeats until every one shows
// the tag (MI355X guide, hand-off by tagged 8-byte granules: no flag, no ordering between stores, no drain at the
// producer).  Against arrival counters (stores drained, one atomic per workgroup, a poll of the counter, then the loads)
// this takes ~1.7 us off each of the two hops (tests/perf/product_trace.py, profiles/r03_*product_trace*).  The host never
// reuses the tag a slot's words still carry (vp_engine.cpp keeps the last tag per limb slot).  Progress needs only that
// every XCD dispatches its share of the grid in order (a waiting workgroup's producers have smaller indices, so they are
// resident or done); every wait is bounded and reports through a sticky status word.
// Arithmetic, twiddle tables and results are those of ntt_single.hip (the same steps, cut at the region boundary):
// bit-exact against the oracle's ISA machine in tests/test_product.py.
#include "kernels.hpp"
#include "modarith.hpp"
#include "ntt_device.hpp"

namespace aloha {
namespace k {

namespace {

constexpr int PLOGN = 14, PN = 1 << PLOGN, PR = PN / 16;     // row length 1024
constexpr uint32_t kP1PerOp = PR / 128;                      // P1 workgroups (128 threads) per limb and operand: 8
constexpr uint32_t kP2PerLimb = 16;                          // P2 workgroups per limb: one per row
constexpr uint32_t kP3PerLimb = PR / 128;                    // 8
constexpr uint32_t kSpinLimit = 1u << 15;                    // sweeps before a wait gives up (~30 ms)
constexpr u64 kTagMask = 3ull << 62;

typedef ALOHA_AS_GLOBAL uint32_t *gu32_t;
__device__ __forceinline__ u64 ld_handoff(cgptr_t p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }       // global_load_dwordx2 sc1
__device__ __forceinline__ void st_handoff(gptr_t p, u64 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }    // global_store_dwordx2 sc1

// LDS word of element x of a 1024-point row: ntt_single.hip's swizzle, under which the three layouts below touch 32
// distinct 8-byte banks per half-wave (checked exhaustively on the host, tests/test_product.py)
__device__ __forceinline__ uint32_t swz(uint32_t x) { return x ^ ((x >> 4) & 31) ^ ((x >> 5) & 31); }
// layouts of a row in a wave: register r of lane l holds x =
__device__ __forceinline__ uint32_t layA(uint32_t l, uint32_t r) { return (r << 6) | l; }                               // registers = bits 9..6
__device__ __forceinline__ uint32_t layB(uint32_t l, uint32_t r) { return ((l >> 2) << 6) | (r << 2) | (l & 3); }      // registers = bits 5..2
__device__ __forceinline__ uint32_t layC(uint32_t l, uint32_t r) { return ((r >> 2) << 8) | (l << 2) | (r & 3); }      // registers = bits 9..8, 1..0

struct TwRow {                     // entries 2^(S0+ss) + a*2^ss + g of the plain (w, w') table
  ctw_t tab;
  uint32_t base, a;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return load_tw(tab + (base << ss) + (a << ss) + g); }
};

// per-lane twiddles as ordinary global loads (they keep their program order): issued where the code says, not at their use
__device__ __forceinline__ tw_t ld_tw_now(ctw_t p) {
  const gtw_t g = (gtw_t)p;
  tw_t t;
  t.w = g->w;
  t.wp = g->wp;
  return t;
}
__device__ __forceinline__ void preload15(tw_t (&t)[15], ctw_t tab, uint32_t base, uint32_t a) {
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) t[(1 << ss) - 1 + g] = ld_tw_now(tab + (base << ss) + (a << ss) + g);
}

// the wave's 16 x 64 tagged words at p + off(r): swept with sc1 loads until every word of every lane carries `tag`
// (bits 63..62); returns them with the tag removed
template <class OFF>
__device__ __forceinline__ void load_tagged(u64 (&v)[16], cgptr_t p, const OFF &off, u64 tag, gu32_t status) {
  uint32_t sweeps = 0;
  for (;;) {
    u64 bad = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      v[r] = ld_handoff(p + off(r));
      bad |= v[r] ^ tag;
    }
    if (__builtin_amdgcn_ballot_w64((bad & kTagMask) != 0) == 0) break;
    if (++sweeps > kSpinLimit) {                                 // (cannot happen in a correct schedule: reported, not hung on)
      if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
        __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] &= ~kTagMask;
}

// values < B*q  ->  < 4q (borrow-subtracts of ceil(b/2)*q, as in the transforms' range plans); tagged_bound: what is left
constexpr int tagged_bound(int b) { while (b > 4) b = (b + 1) / 2; return b; }
template <int B>
__device__ __forceinline__ u64 below_4q(u64 x, u64 q) {
  if constexpr (B <= 4) return x;
  else return below_4q<(B + 1) / 2>(csub_u(x, q * (u64)((B + 1) / 2)), q);
}

// Variant builds with -DALOHA_PROD_TRACE: every workgroup leaves constant-rate clock stamps (100 MHz) of its steps in the
// scratch behind the intermediates: where a launch's time goes (tests/perf/product_trace.py).
#ifdef ALOHA_PROD_TRACE
#define PROD_STAMP(slot) do { if (lane == 0 && wv == 0) trace[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#define PROD_STAMP_VM(slot) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PROD_STAMP(slot); } while (0)   // ... once the loads have landed
#else
#define PROD_STAMP(slot) do { } while (0)
#define PROD_STAMP_VM(slot) do { } while (0)
#endif

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void poly_product_kernel(
    const ProdJob *__restrict__ jobs, const PrimeDev *__restrict__ primes, u64 *__restrict__ work, uint32_t *__restrict__ status_, uint32_t count, uint32_t tag_) {
  __shared__ u64 lds[2 * PR];
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = uni(tid >> 6);
  const uint32_t b = blockIdx.x;
  const uint32_t nP1 = count * 2 * kP1PerOp, nP2 = count * kP2PerLimb;
  gu32_t status = (gu32_t)status_;
  const u64 tag = (u64)(tag_ & 3) << 62;
#ifdef ALOHA_PROD_TRACE
  u64 *trace = work + (size_t)3 * kProdMaxLimbs * PN;
#endif
  PROD_STAMP(0);
  constexpr int BI = tagged_bound(bound_after4(FWD_IN));         // bound of a tagged forward intermediate (P1's output after below_4q)
  if (b < nP1) {
    // ---- P1: forward stages 0..3 of one operand, 128 columns ------------------------------------------------------------
    const uint32_t limb = b / (2 * kP1PerOp), idx = b % (2 * kP1PerOp), op = idx / kP1PerOp, t = (idx % kP1PerOp) * 128 + tid;
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    cgptr_t src = (cgptr_t)(op ? jb.b : jb.a) + t;
    gptr_t w = (gptr_t)work + (size_t)(3 * limb + op) * PN + t;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = ld_stream(src + r * PR);
    PROD_STAMP_VM(4);
    fwd_radix16<true, FWD_IN>(v, TwRow{(ctw_t)pr->tw.fwd, 1, 0}, q, nq);
    constexpr int B1 = bound_after4(FWD_IN);
    static_assert(B1 <= 16 && BI <= 4, "P1's output does not fit the tagged format");
    PROD_STAMP(1);
#pragma unroll
    for (int r = 0; r < 16; r++) st_handoff(w + r * PR, below_4q<B1>(v[r], q) | tag);
    PROD_STAMP(2);
    return;
  }
  if (b < nP1 + nP2) {
    // ---- P2: one row of both operands: forward stages 4..13, product, inverse stages 13..4 ------------------------------------
    const uint32_t i = b - nP1, limb = i / kP2PerLimb, rho = i % kP2PerLimb;
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    ctw_t ftab = (ctw_t)pr->tw.fwd;
    u64 *my = lds + wv * PR;
    cgptr_t row = (cgptr_t)work + (size_t)(3 * limb + wv) * PN + rho * PR;    // wave 0: operand a, wave 1: operand b
    const uint32_t hiB = lane >> 2;
    // A wave alone on its SIMD has nothing to hide a load's latency behind: every per-lane twiddle of the forward steps and
    // of the first inverse step is requested BEFORE the wait for the producers (156 VGPRs; the wave may use 256), the second
    // inverse step's once the forward step that held their registers is through.
    ctw_t itab = (ctw_t)pr->tw.inv;
    tw_t twB[15], tw12[4], tw13[8], itw13[8], itw12[4], itwB[15];
    preload15(twB, ftab, 256, rho * 16 + hiB);
#pragma unroll
    for (int h = 0; h < 4; h++) tw12[h] = ld_tw_now(ftab + 4096 + rho * 256 + h * 64 + lane);
#pragma unroll
    for (int j = 0; j < 8; j++) tw13[j] = ld_tw_now(ftab + 8192 + rho * 512 + (j >> 1) * 128 + 2 * lane + (j & 1));
    if (!wv) {
#pragma unroll
      for (int j = 0; j < 8; j++) itw13[j] = ld_tw_now(itab + 8192 + rho * 512 + (j >> 1) * 128 + 2 * lane + (j & 1));
#pragma unroll
      for (int h = 0; h < 4; h++) itw12[h] = ld_tw_now(itab + 4096 + rho * 256 + h * 64 + lane);
    }
    u64 v[16];
    load_tagged(v, row, [&](int r) { return layA(lane, r); }, tag, status);
    PROD_STAMP(2);
    constexpr int B1 = BI, B2 = bound_after4(B1), B3 = bound_after4(B2), B4 = next_bound(B3), B5 = next_bound(B4);
    static_assert(B1 <= 16 && B2 <= 16 && B3 <= 16 && B4 <= 16 && B5 <= 16, "forward lazy range overflows");
    fwd_radix16<true, B1>(v, TwRow{ftab, 16, rho}, q, nq);                                // stages 4..7
    PROD_STAMP(3);
#pragma unroll
    for (int r = 0; r < 16; r++) my[swz(layA(lane, r))] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = my[swz(layB(lane, r))];
    PROD_STAMP(4);
    fwd_radix16<false, B2>(v, TwStepReg{twB}, q, nq);                                     // stages 8..11
    PROD_STAMP(5);
    if (!wv) preload15(itwB, itab, 256, rho * 16 + hiB);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) my[swz(layB(lane, r))] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = my[swz(layC(lane, r))];
    PROD_STAMP(6);
    const u64 fourq = q << 2;
#pragma unroll
    for (int h = 0; h < 4; h++) {                                // stage 12: x bit 1; group e >> 2
      const tw_t w = tw12[h];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (need_corr(B3)) v[4 * h + j] = csub_u(v[4 * h + j], q * (u64)corr_t(B3));
        ct_bfly<false, false>(v[4 * h + j], v[4 * h + j + 2], w.w, w.wp, nq, fourq);
      }
    }
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
      for (int j = 0; j < 2; j++) {                              // stage 13: x bit 0; group e >> 1
        const tw_t w = tw13[2 * h + j];
        if (need_corr(B4)) v[4 * h + 2 * j] = csub_u(v[4 * h + 2 * j], q * (u64)corr_t(B4));
        ct_bfly<false, false>(v[4 * h + 2 * j], v[4 * h + 2 * j + 1], w.w, w.wp, nq, fourq);
      }
    PROD_STAMP(7);
    const u32 kc = (u32)pr->m.kc;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = canon_any(v[r], q, nq, kc);
    // operand b's row crosses to the other wave (its own region: the wave is done with its exchanges, and LDS
    // instructions of one wave execute in order)
    if (wv) {
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 16; r++) my[r * 64 + lane] = v[r];
    }
    PROD_STAMP(8);
    __syncthreads();
    if (wv) return;
    PROD_STAMP(9);
    const u64 mu = pr->m.mu;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = mulmod_lazy_2q(v[r], lds[PR + r * 64 + lane], nq, mu);   // < 2q
    PROD_STAMP(10);
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const tw_t w = itw13[2 * h + j];
        gs_bfly<false>(v[4 * h + 2 * j], v[4 * h + 2 * j + 1], w.w, w.wp, nq, fourq);
      }
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const tw_t w = itw12[h];
#pragma unroll
      for (int j = 0; j < 2; j++) gs_bfly<false>(v[4 * h + j], v[4 * h + j + 2], w.w, w.wp, nq, fourq);
    }
    PROD_STAMP(11);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) my[swz(layC(lane, r))] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = my[swz(layB(lane, r))];
    PROD_STAMP(12);
    inv_radix16<false, INV_MID, false>(v, TwStepReg{itwB}, q, nq);                        // stages 11..8
    PROD_STAMP(13);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) my[swz(layB(lane, r))] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = my[swz(layA(lane, r))];
    PROD_STAMP(14);
    inv_radix16<true, INV_MID, false>(v, TwRow{itab, 16, rho}, q, nq);                    // stages 7..4
    PROD_STAMP(15);
    gptr_t out = (gptr_t)work + (size_t)(3 * limb + 2) * PN + rho * PR;
#pragma unroll
    for (int r = 0; r < 16; r++) st_handoff(out + layA(lane, r), v[r] | tag);             // < 4q (INV_MID)
    return;
  }
  {
    // ---- P3: inverse stages 3..0 with N^-1, 128 columns -------------------------------------------------------------------------
    const uint32_t i = b - nP1 - nP2, limb = i / kP3PerLimb, t = (i % kP3PerLimb) * 128 + tid;
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    cgptr_t w = (cgptr_t)work + (size_t)(3 * limb + 2) * PN + t;
    gptr_t dst = (gptr_t)jb.d + t;
    const LastScale ls = last_scale<false>(nullptr, pr);
    u64 v[16];
    load_tagged(v, w, [&](int r) { return (uint32_t)r * PR; }, tag, status);
    PROD_STAMP(4);
    inv_radix16<true, INV_MID, true>(v, TwRow{(ctw_t)pr->tw.inv, 1, 0}, q, nq, ls.w, ls.ni, ls.nip);
    PROD_STAMP(2);
#pragma unroll
    for (int r = 0; r < 16; r++) st_stream(dst + r * PR, canon_4q(v[r], q));
    PROD_STAMP(3);
  }
}

}  // namespace

size_t poly_product_work_words() { return (size_t)3 * kProdMaxLimbs * PN; }

hipError_t poly_product_batch(hipStream_t st, const ProdJob *d_jobs, const PrimeDev *d_primes, uint32_t count, u64 *d_work, uint32_t *d_status,
                              uint32_t tag, uint64_t *launches) {
  if (!count) return hipSuccess;
  if (count > kProdMaxLimbs) return hipErrorInvalidValue;
  const uint32_t grid = count * (2 * kP1PerOp + kP2PerLimb + kP3PerLimb);
  hipLaunchKernelGGL(poly_product_kernel, dim3(grid), dim3(128), 0, st, d_jobs, d_primes, d_work, d_status, count, tag);
  if (launches) ++*launches;
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
