// ntt_tiles.hpp -- the tile bodies of the two-pass transforms (N = 2^16 column tile, 256-point row tile), shared by
// ntt_tile.hip (plain transforms) and ntt_ks.hip (the key-switch forms: a row pass that multiplies by the key and sums
// over the digits instead of storing).  Conventions, twiddle layout and arithmetic: header of ntt_tile.hip.
#pragma once
#include "kernels.hpp"
#include "modarith.hpp"
#include "ntt_device.hpp"

namespace aloha {
namespace k {
namespace {

constexpr int LINES = 16;          // lines per column tile
constexpr int PITCH = 257;         // column-tile LDS pitch in words: odd => conflict-free 64-bit accesses; 32896 B, 4 workgroups per CU
constexpr int TW_BLK = 256;        // twiddle entries per block
// Row pass: LDS slot of point p of wave-line lw is lw*RP + p + (p >> 4).  gfx950 serves a ds_write_b64 16 lanes at a time on
// (a/4) mod 32 and a ds_read_b64 32 lanes at a time on (a/4) mod 64 (MI355X_MICROARCH.md, LDS): an access pattern is free of
// conflicts when the 8-byte slots of a 16-lane group differ mod 16 (stores) / of a 32-lane group mod 32 (loads).
//   ALOHA_ROW_BMAP = 0 (shipped): B mapping lane = (lw = lane & 3, kh = lane >> 2), RP = 280 (== 24 mod 32): the first exchange
//                       (A stores, B loads) is conflict-free, the trip back to the A mapping (B stores, A loads) is 2-way
//                       conflicted on both sides: a third of the row kernels' LDS cycles are conflict cycles (VERDICT r3;
//                       tests/test_lds_layout.py reproduces it from the two rules).
//   ALOHA_ROW_BMAP = 1: lane = (kh = lane & 15, lw = lane >> 4), RP = 272 (== 16 mod 32): all four patterns conflict-free
//                       (same test; PMC: SQ_LDS_BANK_CONFLICT 33.5 M -> 0 per headline launch, LDS cycles -33 %) -- and
//                       SLOWER, measured in round 1 and again in round 4 at equal occupancy on one box
//                       (profiles/r04_row_lds_layout_ab.txt: forward -0.4 %, inverse -1.3 %, HMult +1 %): the LDS is a quarter
//                       busy in these kernels, its conflict cycles hide behind the butterflies, and the other lane mapping costs
//                       registers (98 instead of 96 in the headline kernel: 4 waves per SIMD instead of 5 unless a twiddle
//                       preload is given up).  Kept as a build knob with its test; not shipped.
#ifndef ALOHA_ROW_BMAP
#define ALOHA_ROW_BMAP 0
#endif
constexpr int RP = ALOHA_ROW_BMAP ? 272 : 280;
// Per-lane twiddles issued ahead of the data they meet (measured in rounds 1-2, profiles/r01_w_*):
constexpr int kInvColPreload = 15;   // inverse column pass: all 15 of the first step before the data loads (0 / 8 / 15: 2.38 / 2.38 / 2.44 M inverse NTT/s)
#ifndef ALOHA_ROW_PRELOAD
#define ALOHA_ROW_PRELOAD 7
#endif
constexpr int kRowPreload = ALOHA_ROW_PRELOAD;   // forward row pass: sub-stages 0..2 of the second step before the first step
constexpr int kInvRowPreload = 15;   // inverse row pass (0 / 12 / 15 = 2.44 / 2.45 / 2.48 M)

// Variant builds with -DALOHA_TILE_TRACE: the first thread of every workgroup leaves constant-rate clock stamps (100 MHz) of its
// steps behind the prime table (kernels.hpp: kTileTraceWords): where the time of a SMALL launch goes -- one tile's latency, which
// is what the key-switch's 8- and 24-limb transforms cost (tests/perf/tile_trace.py).  _VM: once the loads have landed.
#ifdef ALOHA_TILE_TRACE
// (column kernels stamp workgroups [0, 2048) of the buffer, row kernels [2048, 8192): both passes of one transform stay readable)
#define TILE_STAMP_AT(prime_ptr, prime_idx, first, count, slot)                                                                 \
  do {                                                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < (count))                                                                                \
      ((u64 *)((prime_ptr) - (prime_idx) + kPrimeTableEntries))[(size_t)((first) + blockIdx.x) * 16 + (slot)] = wall_clock64();   \
  } while (0)
#define TILE_STAMP(prime_ptr, prime_idx, slot) TILE_STAMP_AT(prime_ptr, prime_idx, 0, 2048, slot)
#define ROW_STAMP(prime_ptr, prime_idx, slot) TILE_STAMP_AT(prime_ptr, prime_idx, 2048, 6144, slot)
#define TILE_STAMP_VM(prime_ptr, prime_idx, slot) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TILE_STAMP(prime_ptr, prime_idx, slot); } while (0)
#define ROW_STAMP_VM(prime_ptr, prime_idx, slot) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ROW_STAMP(prime_ptr, prime_idx, slot); } while (0)
#else
#define TILE_STAMP(prime_ptr, prime_idx, slot) do { } while (0)
#define TILE_STAMP_VM(prime_ptr, prime_idx, slot) do { } while (0)
#define ROW_STAMP(prime_ptr, prime_idx, slot) do { } while (0)
#define ROW_STAMP_VM(prime_ptr, prime_idx, slot) do { } while (0)
#endif

// ---- scalar-sum prologue of a forward column tile ------------------------------------------------------------------
// The limb to transform is not in memory: it is  sum_t a_t[i] * s_t + cst  (mod q) over <= kSumTerms memory operands (a
// basis conversion: keyswitch.mem:9-14,:29-34,:86-91 are the reference's one-operand cases).  <= 4 terms: approximate
// quotients (< 4q each), else exact ones (< 2q each): < 16q < 2^64 either way; one canon_any, the constant on top.
__device__ __forceinline__ u64 shoup_2q_uni(u64 a, u64 s, u64 sp, u64 nq) {
  return mullo2_acc<true>(a, s, shoup_qhat_exact<true>(a, sp), nq);
}
// Two operands at a time: their 2 x NV loads are in flight together (a loop that waited for every operand in turn was
// latency-bound: 66 us for the 144 limbs of a key-switch against 72 us for the two separate kernels), more would not fit
// the register budget of 4 waves per SIMD.
template <bool APPROX, int NV, class ADDR>
__device__ __forceinline__ void sum_pair(u64 (&v)[NV], const SumSrc *__restrict__ ss, int t, bool two, bool first, const ADDR &addr, u64 nq) {
  cgptr_t a0 = (cgptr_t)ss->a[t], a1 = (cgptr_t)ss->a[two ? t + 1 : t];
  const u64 s0 = ss->s[t], sp0 = ss->sp[t], s1 = ss->s[two ? t + 1 : t], sp1 = ss->sp[two ? t + 1 : t];
  u64 x0[NV], x1[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) x0[i] = a0[addr(i)];
  if (two) {
#pragma unroll
    for (int i = 0; i < NV; i++) x1[i] = a1[addr(i)];
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const u64 p = APPROX ? mul_shoup_4q<true>(x0[i], s0, sp0, nq) : shoup_2q_uni(x0[i], s0, sp0, nq);
    v[i] = first ? p : v[i] + p;
  }
  if (two) {
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] += APPROX ? mul_shoup_4q<true>(x1[i], s1, sp1, nq) : shoup_2q_uni(x1[i], s1, sp1, nq);
  }
}
// ALOHA_PRO_CHAIN = 1 (round 4, shipped): the <= 4 products of an element on SHARED multiply-add chains -- a*s accumulates over
// the terms on one 64-bit chain (+ one for the cross terms), the approximate quotients on a third, and quotient * (2^64 - q) is added
// ONCE: 8 instructions per term + 4 per element instead of 14 per term.  The same words: sum_t (a_t s_t - qhat_t q) is linear in the
// low 64 bits, every term lies in [0, 4q), four of them below 2^64.  Six live registers per element instead of two, so the tile's
// 16 elements go in quarters with all terms' loads of a quarter in flight (112 VGPRs, no scratch).  5..8 terms: two such sums, each
// canonicalised.  Measured on one box, interleaved (profiles/r04_modup_prologue_chain_ab.txt): HMult L = 24 alpha = 4 0.2738 -> 0.2677
// ms, alpha = 8 0.2500 -> 0.2419; rotate L = 32 alpha = 4 0.3879 -> 0.3834, alpha = 8 0.3427 -> 0.3264.  = 0: rounds 2-3 (two operands
// at a time over all 16 elements, one reduced product per term).
#ifndef ALOHA_PRO_CHAIN
#define ALOHA_PRO_CHAIN 1
#endif
template <int NV, class ADDR>
__device__ __forceinline__ void sum_prologue_chain(u64 (&v)[NV], const SumSrc *__restrict__ ss, const ADDR &addr, u64 q, u64 nq, u32 kc) {
  const int T = ss->nterms;                                      // (wave-uniform)
  const u64 c = ss->cst;
  const u32 n0 = lo32(nq), n1 = hi32(nq);
#pragma unroll
  for (int qb = 0; qb < NV / 4; qb++) {
    u64 res[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
      if (4 * g >= T) { res[g] = 0; continue; }                  // (wave-uniform)
      u64 x[4][4];
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (4 * g + t < T) {
          cgptr_t a = (cgptr_t)ss->a[4 * g + t];
#pragma unroll
          for (int e = 0; e < 4; e++) x[t][e] = a[addr(4 * qb + e)];
        }
      u64 acc[4], tt[4], Q[4];
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (4 * g + t < T) {
          const u64 s = ss->s[4 * g + t], sp = ss->sp[4 * g + t];
          const u32 s0 = lo32(s), s1 = hi32(s), p0 = lo32(sp), p1 = hi32(sp);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const u32 a0 = lo32(x[t][e]), a1 = hi32(x[t][e]);
            const u32 hc = mulhi32_s(a1, p0), hd = mulhi32_s(a0, p1);
            if (t == 0) {
              Q[e] = mad32_s(a1, p1, (u64)hc + hd);
              tt[e] = mad32_s(a0, s1, mad32_zs(a1, s0));
              acc[e] = mad32_zs(a0, s0);
            } else {
              Q[e] = mad32_s(a1, p1, mad32_one(hd, mad32_one(hc, Q[e])));
              tt[e] = mad32_s(a0, s1, mad32_s(a1, s0, tt[e]));
              acc[e] = mad32_s(a0, s0, acc[e]);
            }
          }
        }
      u64 r[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        u64 t2 = mad32_s(hi32(Q[e]), n0, tt[e]);
        t2 = mad32_s(lo32(Q[e]), n1, t2);
        const u64 a2 = mad32_s(lo32(Q[e]), n0, acc[e]);
        r[e] = pack64(lo32(a2), hi32(a2) + lo32(t2));          // sum of <= 4 products, each in [0, 4q): < 16q
      }
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * qb + e] = r[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * qb + e] = canon_any(v[4 * qb + e], q, nq, kc) + canon_any(r[e], q, nq, kc);   // < 2q
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u64 r = canon_any(v[4 * qb + e], q, nq, kc);
      v[4 * qb + e] = c ? csub_u(r + c, q) : r;                  // (wave-uniform: a basis extension has no constant, the mod-down's has)
    }
  }
}

template <int NV, class ADDR>
__device__ __forceinline__ void sum_prologue(u64 (&v)[NV], const SumSrc *__restrict__ ss, const ADDR &addr, u64 q, u64 nq, u32 kc) {
#if ALOHA_PRO_CHAIN
  sum_prologue_chain<NV>(v, ss, addr, q, nq, kc);
  return;
#endif
  const int T = ss->nterms;                                      // (wave-uniform)
  if (T <= 4) {
    sum_pair<true>(v, ss, 0, T > 1, true, addr, nq);
    if (T > 2) sum_pair<true>(v, ss, 2, T > 3, false, addr, nq);
  } else {
    for (int t = 0; t < T; t += 2) sum_pair<false>(v, ss, t, t + 1 < T, t == 0, addr, nq);
  }
  const u64 c = ss->cst;
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = csub_u(canon_any(v[i], q, nq, kc) + c, q);
}

// ---- column pass (N = 2^16): 16 adjacent columns x 256 rows per workgroup ---------------------
// A mapping: thread (col = tid & 15, pl = tid >> 4) holds rows pl + 16k   (stages 0..3)
// B mapping: thread (col = tid & 15, kh = tid >> 4) holds rows 16*kh + j  (stages 4..7)
// Both mappings touch memory in 128-byte row segments: forward loads in A and stores from B,
// inverse loads in B and stores from A.  One LDS exchange and one barrier per tile.  The
// stage 0..3 twiddles are the same for every thread (scalar loads -> SGPR operands).
template <bool FIRST, bool LAST, bool PRO>
__device__ __forceinline__ void col_tile_fwd(u64 *sdata, const LimbJob &jb, const PrimeDev *pr, uint32_t cg, int tid,
                                             const SumSrc *__restrict__ sums) {
  const uint32_t col = tid & 15, slow = tid >> 4;
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t blk = (ctw_t)pr->tw.fwd_blk;                                      // block 0
  cgptr_t src = (cgptr_t)jb.src + cg * 16;                                // wave-uniform bases
  gptr_t dst = (gptr_t)jb.dst + cg * 16;
  const uint32_t offA = (slow << 8) + col, offB = (slow << 12) + col;     // 32-bit lane offsets
  u64 v[16];
  constexpr int BIN = FIRST ? FWD_IN : FWD_MID;
  TILE_STAMP(pr, jb.prime, 0);
  if (PRO && jb.pro) {                                                    // (workgroup-uniform)
    const uint32_t base = cg * 16 + offA;
    sum_prologue<16>(v, sums + (jb.pro - 1), [&](int kk) { return base + ((uint32_t)(16 * kk) << 8); }, q, nq, (u32)pr->m.kc);
  } else {
#pragma unroll
    for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&(src + ((size_t)(16 * kk) << 8))[offA]);
  }
  // second-step twiddles: issued now (ordinary global loads keep their program order), so
  // their L2 latency hides under the first radix-16 step instead of stalling the second
  tw_t twb[15];
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const gtw_t p = (gtw_t)blk + (16 << ss) + (slow << ss) + g;
      twb[(1 << ss) - 1 + g].w = p->w;
      twb[(1 << ss) - 1 + g].wp = p->wp;
    }
  TILE_STAMP_VM(pr, jb.prime, 1);
  fwd_radix16<true, BIN>(v, TwStepA{blk}, q, nq);
  TILE_STAMP(pr, jb.prime, 2);
#pragma unroll
  for (int kk = 0; kk < 16; kk++) sdata[col * PITCH + slow + 16 * kk] = v[kk];
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) v[jj] = sdata[col * PITCH + 16 * slow + jj];
  TILE_STAMP(pr, jb.prime, 3);
  fwd_radix16<false, bound_after4(BIN)>(v, TwStepReg{twb}, q, nq);
  TILE_STAMP(pr, jb.prime, 4);
#pragma unroll
  for (int jj = 0; jj < 16; jj++)
    st_stream(&(dst + ((size_t)jj << 8))[offB], LAST ? canon_any(v[jj], q, nq, (u32)pr->m.kc) : v[jj]);
  TILE_STAMP(pr, jb.prime, 5);
  TILE_STAMP_VM(pr, jb.prime, 6);
}

template <bool FIRST, bool LAST, bool SC>
__device__ __forceinline__ void col_tile_inv(u64 *sdata, const LimbJob &jb, const LimbJob *jbp, const PrimeDev *pr, uint32_t cg, int tid) {
  const uint32_t col = tid & 15, slow = tid >> 4;
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t blk = (ctw_t)pr->tw.inv_blk;                                      // block 0
  cgptr_t src = (cgptr_t)jb.src + cg * 16;                                // wave-uniform bases
  gptr_t dst = (gptr_t)jb.dst + cg * 16;
  const uint32_t offA = (slow << 8) + col, offB = (slow << 12) + col;     // 32-bit lane offsets
  u64 v[16];
  tw_t twb[15];
  TILE_STAMP(pr, jb.prime, 0);
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++)
      if ((1 << ss) - 1 + g >= 15 - kInvColPreload) {
        const gtw_t p = (gtw_t)blk + (16 << ss) + (slow << ss) + g;
        twb[(1 << ss) - 1 + g].w = p->w;
        twb[(1 << ss) - 1 + g].wp = p->wp;
      }
#pragma unroll
  for (int jj = 0; jj < 16; jj++) v[jj] = ld_stream(&(src + ((size_t)jj << 8))[offB]);
  TILE_STAMP_VM(pr, jb.prime, 1);
  inv_radix16<false, FIRST ? INV_IN : INV_MID, false>(v, TwStepMixHi<kInvColPreload>{twb, blk, (int)slow}, q, nq);
  TILE_STAMP(pr, jb.prime, 2);
#pragma unroll
  for (int jj = 0; jj < 16; jj++) sdata[col * PITCH + 16 * slow + jj] = v[jj];
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = sdata[col * PITCH + slow + 16 * kk];
  TILE_STAMP(pr, jb.prime, 3);
  const LastScale ls = last_scale<LAST && SC>(jbp, pr);
  inv_radix16<true, INV_MID, LAST>(v, TwStepA{blk}, q, nq, ls.w, ls.ni, ls.nip);
  TILE_STAMP(pr, jb.prime, 4);
  const uint32_t zk = (LAST && SC) ? jbp->zk : 0, zkinv = (LAST && SC) ? jbp->zkinv : 0;   // (wave-uniform)
#pragma unroll
  for (int kk = 0; kk < 16; kk++) {
    u64 r = LAST ? canon_last<SC>(v[kk], q, ls.addc) : v[kk];
    if (LAST && SC && zk && r == 0) {                          // vautz folded in: a zero at a sign-flipped target of the automorphism leaves as q
      const uint32_t t = ((uint32_t)(16 * kk) << 8) + offA + cg * 16, i = (t * zkinv) & 0xffffu;
      if (((i * zk) & 0x1ffffu) >= 0x10000u) r = q;
    }
    st_stream(&(dst + ((size_t)(16 * kk) << 8))[offA], r);
  }
  TILE_STAMP(pr, jb.prime, 5);
  TILE_STAMP_VM(pr, jb.prime, 6);
}

// ---- row pass: 256 contiguous points per line, 4 lines per WAVE ------------------------------------------------
// A wave owns 4 lines and a private LDS region (4 x RP words), so the pass has no s_barrier at all (wave_lds_fence).
// A mapping: lane (pl = lane & 15, lw = lane >> 4) holds points pl + 16k  -- 128-byte segments
// B mapping: lane (kh, lw) holds points 16*kh + j; which lanes: ALOHA_ROW_BMAP above
struct RowLanes {
  uint32_t plA, lwA, lwB, khB, ldsA, ldsB;
  __device__ __forceinline__ explicit RowLanes(int lane)
      : plA(lane & 15), lwA(lane >> 4), lwB(ALOHA_ROW_BMAP ? lane >> 4 : lane & 3), khB(ALOHA_ROW_BMAP ? lane & 15 : lane >> 2),
        ldsA((lane >> 4) * RP + (lane & 15)), ldsB(lwB * RP + 17 * khB) {}
};

// entries 2^ss + g (ss = 0..3) of a twiddle block, as ordinary global loads in program order
template <int NPRE = 15>
__device__ __forceinline__ void load_tw15(tw_t (&t)[15], ctw_t blk) {
#pragma unroll
  for (int i = 0; i < NPRE; i++) {
    const gtw_t p = (gtw_t)blk + 1 + i;                            // index 2^ss - 1 + g  <->  entry 2^ss + g
    t[i].w = p->w;
    t[i].wp = p->wp;
  }
}

// The forward transform of the wave's 4 lines: v (A mapping, < BIN*q) -> transformed, canonical when LAST, back in the
// A mapping.  blkA / blkB: the line's twiddle block as seen from the lane's A / B line (UNIA: one block for the wave,
// its first-step twiddles through scalar loads).
// PRE: how many of the second step's per-lane twiddles are issued before the first step (registers against latency);
// PREA: how many of the first step's, for lines with blocks of their own.
template <bool UNIA, int BIN, bool LAST, int PRE = kRowPreload, int PREA = 15>
__device__ __forceinline__ void row_fwd_core(u64 (&v)[16], u64 *lds, const RowLanes &L, ctw_t blkA, ctw_t blkB, const PrimeDev *pr) {
  const u64 q = pr->m.q, nq = pr->m.nq;
  tw_t twb[15];
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++)
      if ((1 << ss) - 1 + g < PRE) {
        const gtw_t p = (gtw_t)blkB + (16 << ss) + (L.khB << ss) + g;
        twb[(1 << ss) - 1 + g].w = p->w;
        twb[(1 << ss) - 1 + g].wp = p->wp;
      }
  if (UNIA) {
    fwd_radix16<true, BIN>(v, TwStepA{blkA}, q, nq);
  } else {
    // lines with different blocks: the first step's 15 twiddles are vector loads (4 distinct addresses per instruction).
    // Read at their use they cost one exposed L2 / HBM latency per sub-stage group (the compiler trickles them in one or
    // two at a time: ~10 waits per tile, which is what made the tiles of 1-2 limbs per prime latency-bound); issued all at
    // once behind the data loads they cost one.
    tw_t twa[15];
    load_tw15<PREA>(twa, blkA);
    fwd_radix16<false, BIN>(v, TwStepMixA<PREA>{twa, blkA}, q, nq);
  }
#pragma unroll
  for (int kk = 0; kk < 16; kk++) lds[L.ldsA + 17 * kk] = v[kk];
  wave_lds_fence();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) v[jj] = lds[L.ldsB + jj];
  fwd_radix16<false, bound_after4(BIN)>(v, TwStepMixT<PRE>{twb, blkB, (int)L.khB}, q, nq);
  if (LAST) {
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = canon_any(v[jj], q, nq, (u32)pr->m.kc);
  }
  // back to the A mapping: global stores then leave in the same 128-byte-segment pattern as the loads (measured free)
  wave_lds_fence();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) lds[L.ldsB + jj] = v[jj];
  wave_lds_fence();
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = lds[L.ldsA + 17 * kk];
}

// The inverse transform of the wave's 4 lines (the FIRST pass of an inverse transform: strides 1..128 inside a 2 KiB row):
// v (A mapping, < BIN*q) -> < INV_MID*q, back in the A mapping.  blkA / blkB: the line's INVERSE twiddle block as seen from the
// lane's A / B line.  Shared by the plain row kernel (ntt_tile.hip) and the key-switch forms (ntt_ks.hip: the tensor product in
// front of it, the inner products' accumulators in front of it).
template <bool UNIA, int BIN>
__device__ __forceinline__ void row_inv_core(u64 (&v)[16], u64 *lds, const RowLanes &L, ctw_t blkA, ctw_t blkB, const PrimeDev *pr) {
  const u64 q = pr->m.q, nq = pr->m.nq;
  tw_t twb[15];
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++)
      if ((1 << ss) - 1 + g >= 15 - kInvRowPreload) {
        const gtw_t p = (gtw_t)blkB + (16 << ss) + (L.khB << ss) + g;
        twb[(1 << ss) - 1 + g].w = p->w;
        twb[(1 << ss) - 1 + g].wp = p->wp;
      }
#pragma unroll
  for (int kk = 0; kk < 16; kk++) lds[L.ldsA + 17 * kk] = v[kk];
  wave_lds_fence();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) v[jj] = lds[L.ldsB + jj];
  inv_radix16<false, BIN, false>(v, TwStepMixHi<kInvRowPreload>{twb, blkB, (int)L.khB}, q, nq);
  // lines with different blocks: the second step's 15 twiddles as ONE group of loads, in flight across the transpose
  // (read at their use they are ~10 separately exposed latencies, see row_fwd_core)
  tw_t twa[15];
  if (!UNIA) load_tw15(twa, blkA);
  // back to the A mapping: second inverse step / 128-byte-segment stores
  wave_lds_fence();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) lds[L.ldsB + jj] = v[jj];
  wave_lds_fence();
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = lds[L.ldsA + 17 * kk];
  if (UNIA) inv_radix16<true, INV_MID, false>(v, TwStepA{blkA}, q, nq);
  else inv_radix16<false, INV_MID, false>(v, TwStepReg{twa}, q, nq);
}

}  // namespace
}  // namespace k
}  // namespace aloha
