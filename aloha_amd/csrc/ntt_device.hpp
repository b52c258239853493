// ntt_device.hpp -- device-side building blocks shared by the transform kernels (ntt_tile.hip: two-pass
// tiles for N = 2^16 and the fallback sizes; ntt_single.hip: single-pass kernels for N = 2^12..2^14):
// twiddle records, streaming loads/stores, the lazy-range plan of the forward butterflies and the
// radix-16 register steps.  See ntt_tile.hip's header for the arithmetic conventions.
#pragma once
#include "kernels.hpp"
#include "modarith.hpp"

namespace aloha {
namespace k {
namespace {

struct alignas(16) tw_t { u64 w, wp; };

// Pointers that arrive through LimbJob / PrimeDev are generic; naming their address space lets
// hipcc emit global_load/global_store (SGPR base + 32-bit lane offset + immediate) instead of
// flat accesses with 64-bit per-lane address arithmetic, and scalar loads (s_load) for
// wave-uniform twiddle reads.
#define ALOHA_AS_GLOBAL __attribute__((address_space(1)))
#define ALOHA_AS_CONST __attribute__((address_space(4)))
typedef const ALOHA_AS_GLOBAL u64 *cgptr_t;
typedef ALOHA_AS_GLOBAL u64 *gptr_t;
typedef const ALOHA_AS_CONST tw_t *ctw_t;    // invariant loads: scalar when uniform, free to sink next to their use
typedef const ALOHA_AS_GLOBAL tw_t *gtw_t;   // ordinary global loads: keep their program order

// Limb data is touched exactly once per pass: ALOHA_NT=1 marks those accesses non-temporal.
#ifndef ALOHA_NT
#define ALOHA_NT 1
#endif
__device__ __forceinline__ u64 ld_stream(cgptr_t p) {
#if ALOHA_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream(gptr_t p, u64 x) {
#if ALOHA_NT
  __builtin_nontemporal_store(x, p);
#else
  *p = x;
#endif
}

// two adjacent words (16-byte aligned) as ONE 16-byte store
typedef u64 u64x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream2(gptr_t p, u64 x, u64 y) {
  u64x2v_t v = {x, y};
  typedef ALOHA_AS_GLOBAL u64x2v_t *gv2_t;
#if ALOHA_NT
  __builtin_nontemporal_store(v, (gv2_t)p);
#else
  *(gv2_t)p = v;
#endif
}

// uniform (SGPR) value from a value the compiler cannot prove wave-uniform
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// ---- lazy ranges of the forward butterflies -----------------------------------------------------
// Bounds are strict, in units of q: a register with bound b holds a value < b*q.  One relaxed
// butterfly (approximate quotient, C = 4q) maps X < b*q, Y arbitrary to outputs < (b+4)*q,
// which must stay <= 16q (< 2^64).  When b + 4 > 16 the X operands (only) of that sub-stage are
// first reduced by one borrow-subtract of ceil(b/2)*q, leaving them < ceil(b/2)*q.
constexpr int corr_t(int b) { return (b + 1) / 2; }
constexpr bool need_corr(int b) { return b + 4 > 16; }
constexpr int next_bound(int b) { return (need_corr(b) ? corr_t(b) : b) + 4; }
constexpr int bound_after4(int b) { return next_bound(next_bound(next_bound(next_bound(b)))); }

// radix-16 step on 16 registers: sub-stage ss pairs v[i] with v[i + (8 >> ss)]; group g of
// sub-stage ss takes twiddle tw(ss, g).  BIN = input bound; output bound = bound_after4(BIN).
template <bool UNI, int BIN, typename TW>
__device__ __forceinline__ void fwd_radix16(u64 (&v)[16], const TW &tw, u64 q, u64 nq) {
#if defined(ALOHA_EXPERIMENTS) && defined(ALOHA_ABLATE) && ALOHA_ABLATE == 1   // variant builds only: memory + LDS skeleton (results invalid)
  return;
#endif
  constexpr int B[5] = {BIN, next_bound(BIN), next_bound(next_bound(BIN)),
                        next_bound(next_bound(next_bound(BIN))), bound_after4(BIN)};
  const u64 fourq = q << 2;
#pragma unroll
  for (int ss = 0; ss < 4; ss++) {
    const int half = 8 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int i0 = g * 2 * half + jj;
        if (need_corr(B[ss])) v[i0] = csub_u(v[i0], q * (u64)corr_t(B[ss]));
        ct_bfly<UNI, false>(v[i0], v[i0 + half], t.w, t.wp, nq, fourq);
      }
    }
  }
}

// Inverse (Gentleman-Sande) butterfly, every value < 4q:
//   X' = X + Y (- 4q) < 4q,   Y' = (X - Y + 4q) * w  in [0, 4q)   (approximate-quotient Shoup)
template <bool UNI>
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, u64 w, u64 wp, u64 nq, u64 fourq) {
  const u64 s = csub_u(X + Y, fourq);
  const u64 d = X + fourq - Y;
  Y = mul_shoup_4q<UNI>(d, w, wp, nq);
  X = s;
}

// ---- lazy ranges of the inverse butterflies -----------------------------------------------------
// In a GS butterfly only the difference is multiplied (and thereby reduced to < 4q whatever its
// size); the sum X + Y doubles its bound at every stage.  Instead of one conditional subtract per
// butterfly (32 per radix-16 step) the sums are left to grow up to 16q and halved -- one
// borrow-subtract of ceil(b/2)*q takes a bound b to ceil(b/2) -- only where a compile-time plan
// says the next addition (or the step's output format) needs it: 20 subtracts per step with
// every input and output of the step < 4q.  Bounds are strict and in units of q, as above.
//   per butterfly:  [halve X] [halve Y]   S = X + Y   D = X + cq*q - Y   Y' = D*w (< 4q)   [halve S]   X' = S
// SCALE (the last step of the last pass): the final sub-stage multiplies the sum by N^-1 and the
// difference by w*N^-1 (both Shoup products: any 64-bit input), which is the reference's
// halving at every stage (halfred.sv:23-26) collected into one factor; outputs < 4q.
struct InvPlan {
  unsigned char hx[4][8][2];   // halving moduli (units of q) applied to X before the sum; 0 = none
  unsigned char hy[4][8][2];   // ... to Y
  unsigned char cq[4][8];      // the butterfly's C in units of q (>= bound of Y; bound of X + cq <= 16)
  unsigned char hs[4][8][3];   // halvings of the sum
  unsigned char fin[16][3];    // normalisation of the step's outputs to the output bound
  int subtracts;               // total borrow-subtracts of the step
  bool ok;
};
constexpr int inv_half(int b) { return (b + 1) / 2; }
constexpr InvPlan inv_plan(int bin, int bout, bool scale) {
  InvPlan p{};
  p.ok = true;
  int b[16] = {};
  for (int i = 0; i < 16; i++) b[i] = bin;
  for (int idx = 0; idx < 4; idx++) {
    const int ss = 3 - idx, half = 8 >> ss;
    for (int g = 0; g < (1 << ss); g++)
      for (int jj = 0; jj < half; jj++) {
        const int k = g * half + jj, i0 = g * 2 * half + jj, i1 = i0 + half;
        int nx = 0, ny = 0;
        while (b[i0] + b[i1] > 16) {
          if (b[i0] >= b[i1]) {
            if (nx == 2) { p.ok = false; break; }
            p.hx[idx][k][nx++] = (unsigned char)inv_half(b[i0]);
            b[i0] = inv_half(b[i0]);
          } else {
            if (ny == 2) { p.ok = false; break; }
            p.hy[idx][k][ny++] = (unsigned char)inv_half(b[i1]);
            b[i1] = inv_half(b[i1]);
          }
          p.subtracts++;
        }
        p.cq[idx][k] = (unsigned char)b[i1];
        int sum = b[i0] + b[i1], ns = 0;
        if (scale && ss == 0) {
          sum = 4;
        } else {
          while (sum > 8) {
            if (ns == 3) { p.ok = false; break; }
            p.hs[idx][k][ns++] = (unsigned char)inv_half(sum);
            sum = inv_half(sum);
            p.subtracts++;
          }
        }
        b[i0] = sum;
        b[i1] = 4;
      }
  }
  for (int i = 0; i < 16; i++) {
    int nf = 0;
    while (b[i] > bout) {
      if (nf == 3) { p.ok = false; break; }
      p.fin[i][nf++] = (unsigned char)inv_half(b[i]);
      b[i] = inv_half(b[i]);
      p.subtracts++;
    }
  }
  return p;
}

constexpr int INV_IN = 2;    // memory operands: < 2^60 < 2q
constexpr int INV_MID = 4;   // format between the steps and passes of an inverse transform

// radix-16 inverse step on 16 registers, sub-stages 3..0 (pairs 1, 2, 4, 8 apart).  Inputs < BIN*q,
// outputs < INV_MID*q (SCALE: < 4q and multiplied by N^-1; `last` = (w, w') of tw(0,0)*N^-1).
template <bool UNI, int BIN, bool SCALE, typename TW>
__device__ __forceinline__ void inv_radix16(u64 (&v)[16], const TW &tw, u64 q, u64 nq, tw_t last = tw_t{0, 0},
                                            u64 ni = 0, u64 nip = 0) {
  constexpr InvPlan P = inv_plan(BIN, INV_MID, SCALE);
  static_assert(P.ok, "inverse lazy range plan failed");
#pragma unroll
  for (int idx = 0; idx < 4; idx++) {
    const int ss = 3 - idx, half = 8 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = (SCALE && ss == 0) ? last : tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int k = g * half + jj, i0 = g * 2 * half + jj, i1 = i0 + half;
        u64 X = v[i0], Y = v[i1];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          if (P.hx[idx][k][h]) X = csub_u(X, q * (u64)P.hx[idx][k][h]);
          if (P.hy[idx][k][h]) Y = csub_u(Y, q * (u64)P.hy[idx][k][h]);
        }
        u64 s = X + Y;
        const u64 d = X + q * (u64)P.cq[idx][k] - Y;
        v[i1] = mul_shoup_4q<UNI>(d, t.w, t.wp, nq);
        if (SCALE && ss == 0) {
          s = mul_shoup_4q<true>(s, ni, nip, nq);
        } else {
#pragma unroll
          for (int h = 0; h < 3; h++)
            if (P.hs[idx][k][h]) s = csub_u(s, q * (u64)P.hs[idx][k][h]);
        }
        v[i0] = s;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int h = 0; h < 3; h++)
      if (P.fin[i][h]) v[i] = csub_u(v[i], q * (u64)P.fin[i][h]);
}

// [0, 4q) -> [0, q)
__device__ __forceinline__ u64 canon_4q(u64 x, u64 q) { return csub_u(csub_u(x, q << 1), q); }

// Constants of the last inverse stage: the prime's (w*N^-1, N^-1), or the job's own with its extra scalar folded in
// (LimbJob::sc) plus a constant to add to the canonical result.  Wave-uniform (scalar loads).
struct LastScale { tw_t w; u64 ni, nip, addc; };
template <bool LAST>
__device__ __forceinline__ LastScale last_scale(const LimbJob *jb, const PrimeDev *pr) {
  LastScale s{tw_t{pr->tw.w1n_inv, pr->tw.w1n_inv_shoup}, pr->tw.n_inv, pr->tw.n_inv_shoup, 0};
  if (LAST && jb->scaled) {
    s.w = tw_t{jb->sc[0], jb->sc[1]};
    s.ni = jb->sc[2]; s.nip = jb->sc[3]; s.addc = jb->addc;
  }
  return s;
}
template <bool SC>
__device__ __forceinline__ u64 canon_last(u64 x, u64 q, u64 addc) {
  const u64 r = canon_4q(x, q);
  return (SC && addc) ? csub_u(r + addc, q) : r;             // (wave-uniform branch)
}

// twiddle accessors over one 256-entry block
__device__ __forceinline__ tw_t load_tw(ctw_t p) {
  tw_t t;
  t.w = p->w;
  t.wp = p->wp;
  return t;
}
struct TwStepA {                   // stages s' = 0..3: L = 2^ss + g   (same for every thread)
  ctw_t blk;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return load_tw(blk + (1 << ss) + g); }
};
struct TwStepB {                   // stages s' = 4..7: L = 2^(4+ss) + kh*2^ss + g
  ctw_t blk;
  int kh;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    return load_tw(blk + (16 << ss) + (kh << ss) + g);
  }
};

template <int NPRE>
struct TwStepMixA {                // first step of a line with a block of its own: the first NPRE entries preloaded, the rest read at use
  const tw_t (&r)[15];
  ctw_t blk;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    const int idx = (1 << ss) - 1 + g;
    return idx < NPRE ? r[idx] : load_tw(blk + (1 << ss) + g);
  }
};

struct TwStepReg {                 // stages 4..7 from 15 preloaded registers: index 2^ss - 1 + g
  const tw_t (&r)[15];
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return r[(1 << ss) - 1 + g]; }
};

template <int NPRE>
struct TwStepMixT {                // the first NPRE entries preloaded, the rest read at use
  const tw_t (&r)[15];
  ctw_t blk;
  int kh;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    const int idx = (1 << ss) - 1 + g;
    return idx < NPRE ? r[idx] : load_tw(blk + (16 << ss) + (kh << ss) + g);
  }
};

template <int NPRE>
struct TwStepMixHi {               // the LAST NPRE entries preloaded (the inverse step uses sub-stage 3 first), the rest read at use
  const tw_t (&r)[15];
  ctw_t blk;
  int kh;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    const int idx = (1 << ss) - 1 + g;
    return idx >= 15 - NPRE ? r[idx] : load_tw(blk + (16 << ss) + (kh << ss) + g);
  }
};

// in-wave LDS hand-off: LDS instructions of one wave execute in issue order, so lanes of the
// same wave need no s_barrier between a ds_write and a dependent ds_read -- only the compiler
// must keep the order.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inverse last pass: multiply by N^-1 and canonicalise.  The reference halves at every stage
// (halfred.sv:23-26); the product of the halvings is N^-1 and the canonical result is the same
// (SURVEY A.2).
__device__ __forceinline__ u64 scale_canon(u64 x, u64 ni, u64 nip, u64 q, u64 nq) {
  const u64 r = mul_shoup_4q<true>(x, ni, nip, nq);          // < 4q
  return csub_u(csub_u(r, q << 1), q);
}

// ---- hand-off between workgroups of ONE launch by tagged words (ntt_product.hip): every handed-over 8-byte word is < 2^62 and
// carries the launch's 2-bit tag in its top bits, written by one sc1 (write-through) store; the consumer sweeps with sc1 loads
// until every word shows the tag
constexpr long long kWaitTicks = 100000000ll;               // a wait gives up after 1 s of wall_clock64 (100 MHz constant-rate counter): a bound in
                                                             // TIME -- a count of sweeps is a different time on a busy chip than on an idle one
constexpr u64 kTagMask = 3ull << 62;

typedef ALOHA_AS_GLOBAL uint32_t *gu32_t;
__device__ __forceinline__ u64 ld_handoff(cgptr_t p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }       // global_load_dwordx2 sc1
__device__ __forceinline__ void st_handoff(gptr_t p, u64 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }    // global_store_dwordx2 sc1

// ---- radix-8 register steps on 512-point regions (ntt_product.hip; the radix-8 single-pass kernel of ntt_single.hip) --------------
// A half row (512 points, 8 per lane) in LDS: word y ^ ((y >> 3) & 31) -- a bijection under which the three layouts below
// touch 32 distinct 8-byte banks per half-wave (checked exhaustively on the host, tests/test_product.py); register r of
// lane l holds y =
__device__ __forceinline__ uint32_t swh(uint32_t y) { return y ^ ((y >> 3) & 31); }
__device__ __forceinline__ uint32_t lay1(uint32_t l, uint32_t r) { return (r << 6) | l; }                               // registers = bits 8..6
__device__ __forceinline__ uint32_t lay2(uint32_t l, uint32_t r) { return ((l >> 3) << 6) | (r << 3) | (l & 7); }      // registers = bits 5..3
__device__ __forceinline__ uint32_t lay3(uint32_t l, uint32_t r) { return (l << 3) | r; }                              // registers = bits 2..0
// swh(layK(l, r)) in CLOSED FORM (round 4): the swizzle touches bits 0..4 only, so the LDS word is a per-lane base XOR a small
// constant of the register index, plus -- for lay1 -- the register's own bits as an instruction immediate: 0 / 1 / 1 vector
// instructions per access where the compiler derived ~2.4 from the composition (tests/test_lds_layout.py checks the identities)
__device__ __forceinline__ uint32_t sw1(uint32_t l, uint32_t r) { return ((l ^ (l >> 3)) ^ ((r & 3) << 3)) + (r << 6); }
__device__ __forceinline__ uint32_t sw2(uint32_t l, uint32_t r) { return ((((l >> 3) << 6) | (l & 7)) ^ (((l >> 3) & 3) << 3)) ^ (r * 9); }
__device__ __forceinline__ uint32_t sw3(uint32_t l, uint32_t r) { return ((l << 3) ^ (l & 31)) ^ r; }

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

__device__ __forceinline__ void preload7(tw_t (&t)[7], ctw_t tab, uint32_t base, uint32_t a) {
#pragma unroll
  for (int ss = 0; ss < 3; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) t[(1 << ss) - 1 + g] = ld_tw_now(tab + (base << ss) + (a << ss) + g);
}
struct TwReg7 {                    // the 7 twiddles of a radix-8 step from registers: index 2^ss - 1 + g
  const tw_t (&r)[7];
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return r[(1 << ss) - 1 + g]; }
};

// radix-8 register steps: sub-stage ss pairs v[i] with v[i + (4 >> ss)], group g of sub-stage ss takes tw(ss, g) -- the
// radix-16 steps of ntt_device.hpp with three sub-stages.  Forward: relaxed butterflies on the lazy-range plan (BIN -> bound_after3(BIN));
// inverse: Gentleman-Sande butterflies, every value < 4q before and after.
constexpr int bound_after3(int b) { return next_bound(next_bound(next_bound(b))); }
template <bool UNI, int BIN, typename TW>
__device__ __forceinline__ void fwd_radix8(u64 (&v)[8], const TW &tw, u64 q, u64 nq) {
  constexpr int B[3] = {BIN, next_bound(BIN), next_bound(next_bound(BIN))};
  const u64 fourq = q << 2;
#pragma unroll
  for (int ss = 0; ss < 3; ss++) {
    const int half = 4 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int i0 = g * 2 * half + jj;
        if (need_corr(B[ss])) v[i0] = csub_u(v[i0], q * (u64)corr_t(B[ss]));
        ct_bfly<UNI, false>(v[i0], v[i0 + half], t.w, t.wp, nq, fourq);
      }
    }
  }
}
template <bool UNI, typename TW>
__device__ __forceinline__ void inv_radix8(u64 (&v)[8], const TW &tw, u64 q, u64 nq) {
  const u64 fourq = q << 2;
#pragma unroll
  for (int idx = 0; idx < 3; idx++) {
    const int ss = 2 - idx, half = 4 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int i0 = g * 2 * half + jj;
        gs_bfly<UNI>(v[i0], v[i0 + half], t.w, t.wp, nq, fourq);
      }
    }
  }
}

// the wave's NV x 64 tagged words at p + off(r): swept with sc1 loads until every word of every lane carries `tag`
// (bits 63..62); returns them with the tag removed
template <int NV, class OFF>
__device__ __forceinline__ void load_tagged(u64 (&v)[NV], cgptr_t p, const OFF &off, u64 tag, gu32_t status) {
  long long t0 = 0;
  for (uint32_t sweeps = 0;; sweeps++) {
    u64 bad = 0;
#pragma unroll
    for (int r = 0; r < NV; r++) {
      v[r] = ld_handoff(p + off(r));
      bad |= v[r] ^ tag;
    }
    if (__builtin_amdgcn_ballot_w64((bad & kTagMask) != 0) == 0) break;
    if ((sweeps & 63) == 0) {                                    // (the clock is read on the first sweep that fails and every 64th after it)
      const long long now = wall_clock64();
      if (!sweeps) t0 = now;
      else if (now - t0 > kWaitTicks) {                          // (cannot happen while every XCD dispatches in order: reported, not hung on)
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
          __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int r = 0; r < NV; r++) v[r] &= ~kTagMask;
}

// Intermediate formats between the passes of one transform:
//   forward: column pass stores values < FWD_MID*q;   inverse: row pass stores values < 4q.
// A FIRST pass reads memory operands: < 2^60 < 2q (q > 2^59), i.e. bound 2 -- the RTL's one
// conditional subtract (modalu.sv:44-46) is subsumed by the lazy range.
constexpr int FWD_IN = 2;
constexpr int FWD_MID = bound_after4(bound_after4(FWD_IN));
static_assert(bound_after4(bound_after4(FWD_MID)) <= 16 && FWD_MID <= 16, "forward lazy range overflows");

}  // namespace
}  // namespace k
}  // namespace aloha
