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
constexpr uint32_t kWG = 256;                                // threads per workgroup
constexpr uint32_t kP1PerOp = 2 * PR / kWG;                  // P1 workgroups per limb and operand: 8 (128 columns x two halves of the outputs)
constexpr int PH = PR / 2;                                    // half row: 512 points
constexpr uint32_t kP2PerLimb = 16;                          // P2 workgroups per limb: one per row
constexpr uint32_t kP3PerLimb = 2 * PR / kWG;                // 8
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void poly_product_kernel(
    const ProdJob *__restrict__ jobs, const PrimeDev *__restrict__ primes, u64 *__restrict__ work, uint32_t *__restrict__ status_, uint32_t count, uint32_t tag_) {
  __shared__ u64 lds[4 * PH];
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
    // Two waves per 64 columns, as in P2: stage 0 pairs row r with r + 8, both waves load all 16 rows and each keeps its half
    // of that stage's outputs (rows 8 hh + j), then runs stages 1..3 on its 8 values and stores 8.
    const uint32_t limb = b / (2 * kP1PerOp), idx = b % (2 * kP1PerOp), op = idx / kP1PerOp;
    const uint32_t hh = uni(tid >> 7), t = (idx % kP1PerOp) * 128 + (tid & 127);
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    ctw_t ftab = (ctw_t)pr->tw.fwd;
    cgptr_t src = (cgptr_t)(op ? jb.b : jb.a) + t;
    gptr_t w = (gptr_t)work + (size_t)(3 * limb + op) * PN + hh * (8 * PR) + t;
    u64 v[16], u[8];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = ld_stream(src + r * PR);
    PROD_STAMP_VM(4);
    constexpr int B0 = next_bound(FWD_IN), B1 = bound_after3(B0);
    static_assert(!need_corr(FWD_IN) && B1 <= 16 && BI <= 4 && B1 == bound_after4(FWD_IN), "P1's output does not fit the tagged format");
    {
      const tw_t w0 = load_tw(ftab + 1);                         // stage 0
      const u64 fourq = q << 2;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        ct_bfly<true, false>(v[j], v[j + 8], w0.w, w0.wp, nq, fourq);
        u[j] = hh ? v[j + 8] : v[j];
      }
    }
    fwd_radix8<true, B0>(u, TwRow{ftab, 2, hh}, q, nq);                                   // stages 1..3
    PROD_STAMP(1);
#pragma unroll
    for (int j = 0; j < 8; j++) st_handoff(w + j * PR, below_4q<B1>(u[j], q) | tag);
    PROD_STAMP(2);
    return;
  }
  if (b < nP1 + nP2) {
    // ---- P2: one row of both operands: forward stages 4..13, product, inverse stages 13..4 ------------------------------------
    // Four waves: (operand, half of the row).  Stages 5..13 stay inside a half; stage 4 pairs x with x + 512, so for the
    // forward transform both waves of an operand load the whole row and each keeps its half of that stage's outputs (8
    // redundant products per lane instead of an exchange); for the inverse the halves meet in LDS.  A half is 512 points =
    // 8 per lane: three radix-8 register steps, two wave-private exchanges.
    const uint32_t i = b - nP1, limb = i / kP2PerLimb, rho = i % kP2PerLimb;
    const uint32_t op = wv >> 1, h = wv & 1, R = rho * 2 + h;   // R: the half row's index among the limb's 32
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    ctw_t ftab = (ctw_t)pr->tw.fwd, itab = (ctw_t)pr->tw.inv;
    u64 *my = lds + wv * PH;
    cgptr_t row = (cgptr_t)work + (size_t)(3 * limb + op) * PN + rho * PR;
    // A wave alone on its SIMD has nothing to hide a load's latency behind: every per-lane twiddle is requested BEFORE the
    // wait for the producers (the wave may use 256 VGPRs)
    tw_t tw2[7], tw3[7], itw3[7], itw2[7];
    preload7(tw2, ftab, 256, R * 8 + (lane >> 3));
    preload7(tw3, ftab, 2048, R * 64 + lane);
    if (!op) {
      preload7(itw3, itab, 2048, R * 64 + lane);
      preload7(itw2, itab, 256, R * 8 + (lane >> 3));
    }
    u64 v[16];
    load_tagged(v, row, [&](int r) { return (uint32_t)(r << 6) | lane; }, tag, status);    // x = r * 64 + lane: registers = x bits 9..6
    PROD_STAMP(2);
    constexpr int B4 = next_bound(BI), B7 = bound_after3(B4), B10 = bound_after3(B7), B13 = bound_after3(B10);
    static_assert(!need_corr(BI) && B4 <= 16 && B7 <= 16 && B10 <= 16 && B13 <= 16, "forward lazy range overflows");
    const u64 fourq = q << 2;
    u64 u[8];
    {
      const tw_t w4 = load_tw(ftab + 16 + rho);                  // stage 4: pairs (x, x + 512)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        ct_bfly<true, false>(v[j], v[j + 8], w4.w, w4.wp, nq, fourq);
        u[j] = h ? v[j + 8] : v[j];
      }
    }
    fwd_radix8<true, B4>(u, TwRow{ftab, 32, R}, q, nq);                                   // stages 5..7: y bits 8..6
    PROD_STAMP(3);
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw1(lane, r)] = u[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) u[r] = my[sw2(lane, r)];
    PROD_STAMP(4);
    fwd_radix8<false, B7>(u, TwReg7{tw2}, q, nq);                                          // stages 8..10: y bits 5..3
    PROD_STAMP(5);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw2(lane, r)] = u[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) u[r] = my[sw3(lane, r)];
    PROD_STAMP(6);
    fwd_radix8<false, B10>(u, TwReg7{tw3}, q, nq);                                         // stages 11..13: y bits 2..0
    PROD_STAMP(7);
    const u32 kc = (u32)pr->m.kc;
#pragma unroll
    for (int r = 0; r < 8; r++) u[r] = canon_any(u[r], q, nq, kc);
    // operand b's half rows cross to operand a's waves (own region: the wave is done with its exchanges, and LDS
    // instructions of one wave execute in order)
    if (op) {
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) my[r * 64 + lane] = u[r];
    }
    PROD_STAMP(8);
    __syncthreads();
    if (!op) {
      PROD_STAMP(9);
      const u64 mu = pr->m.mu;
#pragma unroll
      for (int r = 0; r < 8; r++) u[r] = mulmod_lazy_2q(u[r], lds[(wv + 2) * PH + r * 64 + lane], nq, mu);   // < 2q
      PROD_STAMP(10);
      inv_radix8<false>(u, TwReg7{itw3}, q, nq);                                            // stages 13..11
      PROD_STAMP(11);
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) my[sw3(lane, r)] = u[r];
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) u[r] = my[sw2(lane, r)];
      PROD_STAMP(12);
      inv_radix8<false>(u, TwReg7{itw2}, q, nq);                                            // stages 10..8
      PROD_STAMP(13);
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) my[sw2(lane, r)] = u[r];
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) u[r] = my[sw1(lane, r)];
      inv_radix8<true>(u, TwRow{itab, 32, R}, q, nq);                                       // stages 7..5
      PROD_STAMP(14);
      // stage 4: the halves meet
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; r++) my[r * 64 + lane] = u[r];
    }
    __syncthreads();                                             // (operand b's waves wait here with the others, then leave)
    if (op) return;
    {
      const tw_t w4 = load_tw(itab + 16 + rho);
      const u64 *other = lds + (wv ^ 1) * PH;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const u64 o = other[r * 64 + lane];
        u[r] = h ? mul_shoup_4q<true>(o + fourq - u[r], w4.w, w4.wp, nq) : csub_u(u[r] + o, fourq);   // (X, Y) -> (X + Y, (X - Y) w): < 4q
      }
    }
    PROD_STAMP(15);
    gptr_t out = (gptr_t)work + (size_t)(3 * limb + 2) * PN + rho * PR + h * PH;
#pragma unroll
    for (int r = 0; r < 8; r++) st_handoff(out + (r << 6) + lane, u[r] | tag);            // < 4q (INV_MID)
    return;
  }
  {
    // ---- P3: inverse stages 3..0 with N^-1, 128 columns -------------------------------------------------------------------------
    // The mirror of P1: a wave runs stages 3..1 on its 8 rows (8 hh + j); for stage 0 the halves meet in LDS.
    const uint32_t i = b - nP1 - nP2, limb = i / kP3PerLimb, c = tid & 127, t = (i % kP3PerLimb) * 128 + c;
    const uint32_t hh = uni(tid >> 7);
    const ProdJob jb = jobs[limb];
    const PrimeDev *pr = &primes[jb.prime];
    const u64 q = pr->m.q, nq = pr->m.nq;
    ctw_t itab = (ctw_t)pr->tw.inv;
    cgptr_t w = (cgptr_t)work + (size_t)(3 * limb + 2) * PN + hh * (8 * PR) + t;
    gptr_t dst = (gptr_t)jb.d + hh * (8 * PR) + t;
    const LastScale ls = last_scale<false>(nullptr, pr);
    u64 u[8];
    load_tagged(u, w, [&](int r) { return (uint32_t)r * PR; }, tag, status);
    PROD_STAMP(4);
    inv_radix8<true>(u, TwRow{itab, 2, hh}, q, nq);                                        // stages 3..1
#pragma unroll
    for (int j = 0; j < 8; j++) lds[hh * (8 * 128) + j * 128 + c] = u[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u64 o = lds[(hh ^ 1) * (8 * 128) + j * 128 + c];
      // stage 0 with the scaling folded in: (X, Y) -> ((X + Y) N^-1, (X - Y) w N^-1), each < 4q
      u[j] = hh ? mul_shoup_4q<true>(o + (q << 2) - u[j], ls.w.w, ls.w.wp, nq) : mul_shoup_4q<true>(u[j] + o, ls.ni, ls.nip, nq);
    }
    PROD_STAMP(2);
#pragma unroll
    for (int j = 0; j < 8; j++) st_stream(dst + j * PR, canon_4q(u[j], q));
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
  hipLaunchKernelGGL(poly_product_kernel, dim3(grid), dim3(kWG), 0, st, d_jobs, d_primes, d_work, d_status, count, tag);
  if (launches) ++*launches;
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
