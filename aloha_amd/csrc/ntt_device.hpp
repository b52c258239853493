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
// ALOHA_ABLATE (timing experiments only, results invalid): 1 = no butterflies (memory + LDS
// skeleton), 2 = no global loads/stores (arithmetic + LDS skeleton).
#ifndef ALOHA_ABLATE
#define ALOHA_ABLATE 0
#endif

template <bool UNI, int BIN, typename TW>
__device__ __forceinline__ void fwd_radix16(u64 (&v)[16], const TW &tw, u64 q, u64 nq) {
#if ALOHA_ABLATE == 1
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
        if (need_corr(B[ss])) v[i0] = csub_borrow(v[i0], q * (u64)corr_t(B[ss]));
        ct_bfly<UNI, false>(v[i0], v[i0 + half], t.w, t.wp, nq, fourq);
      }
    }
  }
}

// Inverse (Gentleman-Sande) butterfly, every value < 4q:
//   X' = X + Y (- 4q) < 4q,   Y' = (X - Y + 4q) * w  in [0, 4q)   (approximate-quotient Shoup)
template <bool UNI>
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, u64 w, u64 wp, u64 nq, u64 fourq) {
  const u64 s = csub_borrow(X + Y, fourq);
  const u64 d = X + fourq - Y;
  Y = mul_shoup_4q<UNI>(d, w, wp, nq);
  X = s;
}

template <bool UNI, typename TW>
__device__ __forceinline__ void inv_radix16(u64 (&v)[16], const TW &tw, u64 q, u64 nq) {
  const u64 fourq = q << 2;
#pragma unroll
  for (int ss = 3; ss >= 0; ss--) {
    const int half = 8 >> ss;
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
  return csub_borrow(csub_borrow(r, q << 1), q);
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
