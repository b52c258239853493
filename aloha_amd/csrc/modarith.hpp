// modarith.hpp -- 64-bit modular arithmetic for gfx950 (CDNA4), device side.
//
// The reference's modalu (src/vp/vxu/modalu.sv) is a pipelined Barrett unit on 23-bit DSP
// slices; on CDNA4 the natural unit is the 32x32->64 multiply-add (v_mad_u64_u32) and the
// 32-bit low/high multiplies.  All moduli are primes q with 2^59 < q < 2^60, so values up to
// 16q fit a 64-bit register: butterflies are evaluated lazily (Harvey) in [0, 4q) / [0, 2q)
// and canonicalised once at the end -- the canonical residues are identical to the
// reference's fully reduced ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aloha {

typedef unsigned long long u64;
typedef unsigned int u32;

#define ALOHA_DEV __device__ __forceinline__

ALOHA_DEV u32 lo32(u64 x) { return (u32)x; }
ALOHA_DEV u32 hi32(u64 x) { return (u32)(x >> 32); }
ALOHA_DEV u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// The three multiplier instructions, pinned by (non-volatile) inline asm: written as C,
// LLVM's InstCombine re-merges the 32-bit limb algebra below into generic 64-bit multiplies
// and the hand-counted issue budget is lost.  The asm statements have only VGPR/SGPR
// operands and no memory side effects, so the scheduler is free to move, CSE or delete them.
// v_mad_u64_u32 also writes a carry-out SGPR pair (VOP3B): always the fixed pair s[94:95], which
// nothing reads -- a compiler-allocated dummy can land on an SGPR that a later VALU instruction
// reads, and the hazard recogniser then pads every such pair with s_nop.
// "_s" variants take the second factor from an SGPR (wave-uniform twiddle / modulus limbs).
// ALOHA_PLAIN_MUL = 1 (round 4 experiment): the same helpers as plain C.  On gfx940+ LLVM's hazard recogniser assumes a "dst
// forwarding" hazard between ANY inline asm and a following instruction that touches one of its registers (the clobbered s[94:95]
// included) and pads with `s_nop 0`: ~25 % of the instruction stream of every transform kernel (452 per 1316 VALU instructions in the
// headline row kernel).  The compiler's own v_mad_u64_u32 gets no padding (tools/ubench_nop.hip); the price is its instruction
// selection: the cross-term chain narrows to v_mul_lo_u32 + v_add3_u32 (~+11 % VALU instructions per butterfly).
#ifndef ALOHA_PLAIN_MUL
#define ALOHA_PLAIN_MUL 0
#endif
#if ALOHA_PLAIN_MUL
ALOHA_DEV u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
ALOHA_DEV u64 mad32_s(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
ALOHA_DEV u64 mad32_z(u32 a, u32 b) { return (u64)a * b; }
ALOHA_DEV u64 mad32_zs(u32 a, u32 b) { return (u64)a * b; }
ALOHA_DEV u64 mad32_one(u32 a, u64 c) { return c + a; }
ALOHA_DEV u32 mullo32(u32 a, u32 b) { return a * b; }
ALOHA_DEV u32 mullo32_s(u32 a, u32 b) { return a * b; }
ALOHA_DEV u32 mulhi32(u32 a, u32 b) { return __umulhi(a, b); }
ALOHA_DEV u32 mulhi32_s(u32 a, u32 b) { return __umulhi(a, b); }
#else
ALOHA_DEV u64 mad32(u32 a, u32 b, u64 c) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "s94", "s95");
  return r;
}
ALOHA_DEV u64 mad32_s(u32 a, u32 b, u64 c) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c) : "s94", "s95");
  return r;
}
// a*b with a zero addend (inline constant): the low dword is mullo, but the result can head a
// 64-bit accumulator chain without a zero-extension move
ALOHA_DEV u64 mad32_z(u32 a, u32 b) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "s94", "s95");
  return r;
}
ALOHA_DEV u64 mad32_zs(u32 a, u32 b) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, %2, 0" : "=v"(r) : "v"(a), "s"(b) : "s94", "s95");
  return r;
}
// c + a as ONE instruction (a 32-bit value added into a 64-bit accumulator: a * 1 + c)
ALOHA_DEV u64 mad32_one(u32 a, u64 c) {
  u64 r;
  asm("v_mad_u64_u32 %0, s[94:95], %1, 1, %2" : "=v"(r) : "v"(a), "v"(c) : "s94", "s95");
  return r;
}
ALOHA_DEV u32 mullo32(u32 a, u32 b) {
  u32 r;
  asm("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
ALOHA_DEV u32 mullo32_s(u32 a, u32 b) {
  u32 r;
  asm("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
ALOHA_DEV u32 mulhi32(u32 a, u32 b) {
  u32 r;
  asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
ALOHA_DEV u32 mulhi32_s(u32 a, u32 b) {
  u32 r;
  asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}

#endif

// ---- canonicalisation ------------------------------------------------------------------
ALOHA_DEV u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }          // [0,2q) -> [0,q)
ALOHA_DEV u64 addmod(u64 a, u64 b, u64 q) { return csub(a + b, q); }        // a,b < q
ALOHA_DEV u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
// halfred.sv:23-26
ALOHA_DEV u64 halfmod(u64 x, u64 q) { return (x >> 1) + ((x & 1) ? ((q + 1) >> 1) : 0); }

// ---- Shoup multiplication by a precomputed constant --------------------------------------
// w < q, wp = floor(w * 2^64 / q).  For ANY 64-bit a:  a*w - floor(a*wp / 2^64) * q  lies in
// [0, 2q) and is congruent to a*w.

// Exact-quotient form (compiler-scheduled 64x64 high multiply): result in [0, 2q).
ALOHA_DEV u64 mul_shoup_2q(u64 a, u64 w, u64 wp, u64 q) {
  u64 qhat = __umul64hi(a, wp);
  return a * w - qhat * q;
}

// Approximate-quotient form: the low x low partial product and the carries out of the low
// half are dropped, so qhat is short by at most 2 and the result lies in [0, 4q) (< 2^62).
// 9 multiplier issues in all (3 for qhat, 6 for the two low products), no carry chains.
// UNI = the constant (w, wp) is wave-uniform and lives in SGPRs.
// ALOHA_QHAT_ADD = 1: the sums of a 32-bit term into a 64-bit one inside the quotients as C additions (a zero-extension move plus
// a 64-bit add each); 0 (shipped, round 4): as ONE multiply-add by 1 (mad32_one) -- -5 % VALU instructions in every transform kernel.
#ifndef ALOHA_QHAT_ADD
#define ALOHA_QHAT_ADD 0
#endif
template <bool UNI>
ALOHA_DEV u64 shoup_qhat_approx(u64 a, u64 wp) {
  u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(wp), w1 = hi32(wp);
  u32 c = UNI ? mulhi32_s(a1, w0) : mulhi32(a1, w0);
  u32 d = UNI ? mulhi32_s(a0, w1) : mulhi32(a0, w1);
#if ALOHA_QHAT_ADD
  u64 cd = (u64)c + d;
#else
  u64 cd = mad32_one(c, (u64)d);                                 // c + d as one multiply-add (c * 1 + {d, 0}): no zero-extension move for c
#endif
  return UNI ? mad32_s(a1, w1, cd) : mad32(a1, w1, cd);
}

// low 64 bits of a*w + b*nq  (nq = 2^64 - q, always wave-uniform): the two 64-bit low
// products share one accumulator chain, every addition folded into a multiply-add; the four
// cross terms -- of which only the low dword matters -- ride a multiply-add chain of their own
// (see shoup_fold_add below: an instruction costs one issue slot whatever it computes).
template <bool UNI>
ALOHA_DEV u64 mullo2_acc(u64 a, u64 w, u64 b, u64 nq) {
  u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(w), w1 = hi32(w);
  u32 b0 = lo32(b), b1 = hi32(b), n0 = lo32(nq), n1 = hi32(nq);
  u64 t = UNI ? mad32_zs(a1, w0) : mad32_z(a1, w0);
  t = UNI ? mad32_s(a0, w1, t) : mad32(a0, w1, t);
  t = mad32_s(b1, n0, t);
  t = mad32_s(b0, n1, t);                                       // lo32(t) = sum of the four cross terms (mod 2^32)
  u64 acc = UNI ? mad32_zs(a0, w0) : mad32_z(a0, w0);
  acc = mad32_s(b0, n0, acc);
  u32 h = hi32(acc);
#if ALOHA_PLAIN_MUL
  h += lo32(t);
#else
  asm("v_add_u32 %0, %0, %1" : "+v"(h) : "v"(lo32(t)));
#endif
  return pack64(lo32(acc), h);
}

template <bool UNI>
ALOHA_DEV u64 mul_shoup_4q(u64 a, u64 w, u64 wp, u64 nq) {
  u64 qhat = shoup_qhat_approx<UNI>(a, wp);
  return mullo2_acc<UNI>(a, w, qhat, nq);
}

// ---- relaxed-range Cooley-Tukey butterflies ------------------------------------------------
// The Shoup product reduces ANY 64-bit input, so only the un-multiplied operand X grows:
// (X, Y) -> (X + T, X - T + C) with T = Y*w mod q in [0, C), C a multiple of q.  Both outputs
// are < bound(X) + C; with q < 2^60 sixteen q's fit a register, so no per-butterfly
// conditional is needed (corrections are applied once per several stages by the caller).
// The sum X + T is folded into the multiply-add chain (T is never materialised) and the
// second output comes from out0 + out1 = 2X + C.

// exact quotient (4 multiplier issues): T in [0, 2q), C = 2q.  twoq2 = 2q.
template <bool UNI>
ALOHA_DEV u64 shoup_qhat_exact(u64 a, u64 wp) {
  u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(wp), w1 = hi32(wp);
  u32 t = UNI ? mulhi32_s(a0, w0) : mulhi32(a0, w0);
  u64 m1 = UNI ? mad32_s(a1, w0, (u64)t) : mad32(a1, w0, (u64)t);                 // < 2^64
  u64 m2 = UNI ? mad32_s(a0, w1, (u64)lo32(m1)) : mad32(a0, w1, (u64)lo32(m1));   // < 2^64
  u64 m3 = UNI ? mad32_s(a1, w1, (u64)hi32(m1)) : mad32(a1, w1, (u64)hi32(m1));   // < 2^64
#if ALOHA_QHAT_ADD
  return m3 + hi32(m2);
#else
  return mad32_one(hi32(m2), m3);
#endif
}

// out0 = X + (a*w + qhat*nq)  (mod 2^64): the X + T sum rides on the accumulator chain.
// Every VALU instruction of this family costs one ~4-cycle issue slot on gfx950 whatever it
// computes (tools/ubench_valu.hip), so the cross terms -- of which only the low dword matters --
// ride a multiply-add chain of their own whose high dword is garbage: 4 instructions instead of
// 4 multiplies + 2 three-input adds + a register move.
#ifndef ALOHA_FOLD_CHAIN
#define ALOHA_FOLD_CHAIN 1
#endif
template <bool UNI>
ALOHA_DEV u64 shoup_fold_add(u64 X, u64 a, u64 w, u64 qhat, u64 nq) {
  u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(w), w1 = hi32(w);
  u32 b0 = lo32(qhat), b1 = hi32(qhat), n0 = lo32(nq), n1 = hi32(nq);
#if ALOHA_FOLD_CHAIN
  u64 t = UNI ? mad32_zs(a1, w0) : mad32_z(a1, w0);
  t = UNI ? mad32_s(a0, w1, t) : mad32(a0, w1, t);
  t = mad32_s(b1, n0, t);
  t = mad32_s(b0, n1, t);                                       // lo32(t) = sum of the four cross terms (mod 2^32)
  u64 acc = UNI ? mad32_s(a0, w0, X) : mad32(a0, w0, X);
  acc = mad32_s(b0, n0, acc);
  u32 h = hi32(acc);
#if ALOHA_PLAIN_MUL
  h += lo32(t);
#else
  asm("v_add_u32 %0, %0, %1" : "+v"(h) : "v"(lo32(t)));          // pinned: as C, LLVM widens this to a 64-bit add plus moves
#endif
  return pack64(lo32(acc), h);
#else
  u32 x0 = UNI ? mullo32_s(a1, w0) : mullo32(a1, w0);
  u32 x1 = UNI ? mullo32_s(a0, w1) : mullo32(a0, w1);
  u32 x2 = mullo32_s(b1, n0);
  u32 x3 = mullo32_s(b0, n1);
  u32 hi = hi32(X) + x0 + x1 + x2 + x3;
  u64 acc = UNI ? mad32_s(a0, w0, pack64(lo32(X), hi)) : mad32(a0, w0, pack64(lo32(X), hi));
  return mad32_s(b0, n0, acc);
#endif
}

template <bool UNI, bool EXACT>
ALOHA_DEV void ct_bfly(u64 &X, u64 &Y, u64 w, u64 wp, u64 nq, u64 C) {
  u64 qhat = EXACT ? shoup_qhat_exact<UNI>(Y, wp) : shoup_qhat_approx<UNI>(Y, wp);
  u64 o0 = shoup_fold_add<UNI>(X, Y, w, qhat, nq);
  u64 o1 = (X << 1) + C - o0;
  X = o0;
  Y = o1;
}

// q == 1 (mod 2^32) (the shape of the reference's q0, q1: src/top/h2_top_no_axilite.sv:31-32):
// nq = 2^64 - q has low word 0xFFFFFFFF, so qhat*nq needs ONE multiply:
//   qhat*nq = (qhat*n1' << 32) - qhat  with n1' = hi32(nq) + 1  (mod 2^64).
template <bool UNI, bool EXACT>
ALOHA_DEV void ct_bfly_p32(u64 &X, u64 &Y, u64 w, u64 wp, u64 nq, u64 C) {
  u64 a = Y;
  u64 qhat = EXACT ? shoup_qhat_exact<UNI>(a, wp) : shoup_qhat_approx<UNI>(a, wp);
  u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(w), w1 = hi32(w);
  u32 x0 = UNI ? mullo32_s(a1, w0) : mullo32(a1, w0);
  u32 x1 = UNI ? mullo32_s(a0, w1) : mullo32(a0, w1);
  u32 x3 = mullo32_s(lo32(qhat), hi32(nq) + 1);
  u32 hi = hi32(X) + x0 + x1 + x3;
  u64 acc = UNI ? mad32_s(a0, w0, pack64(lo32(X), hi)) : mad32(a0, w0, pack64(lo32(X), hi));
  u64 o0 = acc - qhat;
  u64 o1 = (X << 1) + C - o0;
  X = o0;
  Y = o1;
}

// conditional subtraction by the borrow of the subtraction itself (exact 64-bit compare
// for free): x >= m ? x - m : x
ALOHA_DEV u64 csub_borrow(u64 x, u64 m) {
  u64 d;
  const bool borrow = __builtin_usubll_overflow(x, m, &d);
  return borrow ? x : d;
}
// The same, written on the 32-bit halves so that the borrow of the high subtraction itself selects
// (v_sub_co, v_subb_co, 2 x v_cndmask): from the 64-bit form LLVM re-derives the condition with a
// separate 64-bit compare.  (Not inline asm: a VALU write of VCC needs two wait states before a VALU
// read of it on this target, which the compiler only tracks for its own instructions.)
ALOHA_DEV u64 csub_u(u64 x, u64 m) {
  u32 dl, dh, t;
  const bool b1 = __builtin_usub_overflow(lo32(x), lo32(m), &dl);
  const bool b2 = __builtin_usub_overflow(hi32(x), hi32(m), &t);
  const bool b3 = __builtin_usub_overflow(t, (u32)b1, &dh);
  return (b2 | b3) ? x : pack64(dl, dh);
}

// Canonicalise ANY 64-bit x (used once per element at the end of a pass): estimate
// k ~ floor(x / q) from the high words, subtract k*q, one conditional subtraction.
//   qh = hi32(q) in [2^27, 2^28);  kc = floor(2^59 / (qh + 1)) < 2^32 (host-computed)
//   k  = floor(hi32(x) * kc / 2^59) <= hi32(x) / (qh + 1) <= x / q      (never too large)
//   x/q - k < (hi32(x)+1)/qh - hi32(x)/(qh+1) + 2^-27 + 1  < 2          (k >= floor(x/q) - 1)
// so x - k*q lies in [0, 2q).  3 multiplier issues.
ALOHA_DEV u64 canon_any(u64 x, u64 q, u64 nq, u32 kc) {
  const u32 k = mulhi32_s(hi32(x), kc) >> 27;
  u64 r = mad32_s(k, lo32(nq), x);                       // x + k*nq0 (carry into the high word)
  r = pack64(lo32(r), hi32(r) + mullo32_s(k, hi32(nq)));  // + (k*nq1 << 32)   == x - k*q (mod 2^64)
  return csub_u(r, q);
}

// ---- general product of two residues (Barrett, single-word ratio) -----------------------
// a, b < 2^60 (so z = a*b < 2^120); mu = floor(2^121 / q), nq = 2^64 - q.
// x1 = z >> 57 (63 bit), qhat = floor(x1 * mu / 2^64).  With x1 <= z/2^57 < x1+1 and
// mu <= 2^121/q < mu+1:  0 <= z/q - x1*mu/2^64 < (x1 + mu + 1)/2^64 < 0.76, so qhat is
// floor(z/q) or one less, the remainder estimate lies in [0, 2q) and ONE conditional
// subtraction canonicalises it.  Same scheme as the reference's modmul.sv:150-251 (Barrett,
// alpha = n+1) re-cut for 64-bit words.
ALOHA_DEV u64 mulmod_barrett(u64 a, u64 b, u64 q, u64 mu) {
  u64 zl = a * b;
  u64 zh = __umul64hi(a, b);
  u64 x1 = (zh << 7) | (zl >> 57);
  u64 qhat = __umul64hi(x1, mu);
  return csub(zl - qhat * q, q);
}

// The same product on the pinned multiplier forms, without the final conditional subtraction: a, b < 2^60, result in
// [0, 2q).  The 128-bit product is four multiply-adds (each partial product rides the carry of the one before), the
// quotient the exact high word of x1 * mu (wave-uniform mu: SGPR operands), the remainder one multiply-add chain on the
// low word.  ~19 instructions; used where a product's second factor is a vector (the key of the inner product).
ALOHA_DEV u64 mulmod_lazy_2q(u64 a, u64 b, u64 nq, u64 mu) {
  const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  const u64 p00 = mad32_z(a0, b0);
  const u64 m1 = mad32(a0, b1, (u64)hi32(p00));                 // < 2^64: (2^32-1)^2 + 2^32 - 1
  const u64 m2 = mad32(a1, b0, (u64)lo32(m1));
  const u64 m3 = mad32(a1, b1, (u64)hi32(m1));
#if ALOHA_QHAT_ADD
  const u64 zh = m3 + hi32(m2);
#else
  const u64 zh = mad32_one(hi32(m2), m3);
#endif
  const u64 zl = pack64(lo32(p00), lo32(m2));
  const u64 x1 = (zh << 7) | (zl >> 57);
  const u64 qhat = shoup_qhat_exact<true>(x1, mu);              // floor(x1 * mu / 2^64)
  const u64 acc = mad32_s(lo32(qhat), lo32(nq), zl);            // zl + qhat * nq  (mod 2^64)  ==  zl - qhat * q
  return pack64(lo32(acc), hi32(acc) + mullo32_s(hi32(qhat), lo32(nq)) + mullo32_s(lo32(qhat), hi32(nq)));
}

}  // namespace aloha
