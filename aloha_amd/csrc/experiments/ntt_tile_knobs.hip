// experiments/ntt_tile_knobs.hip -- the two-pass tile kernels of round 2 WITH every measured-and-rejected knob, and the
// single-launch transforms (Infinity-Cache hand-off).  NOT part of the default build: `python -m aloha_amd.build
// --variant=name:MACRO=value,...` compiles this file in place of ntt_tile.hip (with -DALOHA_EXPERIMENTS), and the
// tests of the experiments select such a library with ALOHA_VP_LIB.  Kept as the measured reference points of
// DESIGN.md section 3.2.  The default kernels are csrc/ntt_tile.hip + csrc/ntt_tiles.hpp.
//
// ntt_tile.hip -- negacyclic NTT / iNTT for gfx950: radix-16 register butterflies, LDS exchange.
//
// Replaces the reference's vntt/vintt datapath: CT / GS butterflies in modalu
// (src/vp/vxu/modalu.sv:160-165,:223-249,:264-328), the constant-geometry schedule of
// ntt_fsm (src/vp/ntt/ntt_fsm.sv:49-81), the pair/shuffle wiring (vxu_lane.sv:568-577,
// iconn_shuffle.sv:33) and the per-lane twiddle ROM (tf_rom_generator.sv:28-63).  Results
// are the reference's: natural-order in -> bit-reversed out for vntt, the mirror for vintt,
// twiddle psi^bitrev(m+i), every output canonical.
//
// Decomposition (N = 2^n, n = n1 + 8).  Element j = r*256 + x  (row r < 2^n1, column x < 256).
//   forward : COLUMN pass = the first n1 stages (stride >= 256: fixed x, over r)
//             ROW    pass = the last 8 stages  (inside one 2 KiB row)
//   inverse : ROW pass first (strides 1..128), then the COLUMN pass.
// A tile is 16 "lines" x 256 points = 32 KiB:
//   column tile, N = 2^16 (ntt_col_kernel): 16 adjacent columns of one limb, one 256-thread
//             workgroup, one LDS exchange + barrier, XCD-affine tile order;
//   column tile, N = 2^12..2^15 (ntt_col_small_kernel<E>): 16*2^E rows x 256>>E columns;
//   row tile (ntt_row_kernel): 16 lines = (16 >> pb) rows x 2^pb limbs that share a prime
//             (pb = 4..0), so one 4 KiB twiddle block feeds the same row of 2^pb limbs; each
//             WAVE owns 4 lines and runs as its own 64-thread workgroup, no barrier.
// A 256-point line transform is two radix-16 steps (4 butterfly stages each, 16 values per
// thread in registers) with a transpose through LDS in between; the row pass transposes back
// so that global stores leave in the same 128-byte-segment pattern as the loads (measured
// free).  N < 2^12 falls back to radix-2 stage launches (ntt_radix2_batch).
//
// Twiddles: for both passes stage s' (0..7) of a 256-point line uses entry
//   L = 2^s' + (p >> (8 - s'))      p = point index inside the line
// of a 256-entry block of (w, floor(w*2^64/q)) pairs: block 0 is W[0..255] (column pass),
// block 1+r is the row-r slice W[2^(n1+s') + r*2^s' + i] re-indexed the same way (built by
// the host, vp_engine.cpp).  The first radix-16 step's 15 entries do not depend on the
// thread: scalar loads into SGPR operands.  The second step's entries depend on the thread's
// high nibble: vector loads through the cache (16 distinct 16-byte words per instruction).
//
// Arithmetic: lazy, see modarith.hpp.  Forward: relaxed CT butterflies with approximate Shoup
// quotients, values < 16q, borrow-subtracts placed by a compile-time range plan.  Inverse:
// GS butterflies whose sums grow lazily up to 16q on a compile-time plan of their own (inv_plan,
// ntt_device.hpp), < 4q between steps and passes, N^-1 folded into the two products of the last
// stage.  HBM traffic per pass:
// read + write of the limb (+ twiddle blocks); algorithmic bytes per transform: 16*N
// (SURVEY 8d).  Timing experiments (ALOHA_ABLATE builds, profiles/): memory skeleton ~195 us
// per pass per 1024 limbs, arithmetic ~150-180 us; in the row pass the transpose-back costs
// nothing, the final canonicalisation ~4 %, the per-lane second-step twiddles ~7 %.
#include <cstdlib>
#include <vector>

#ifndef ALOHA_ABLATE
#define ALOHA_ABLATE 0
#endif

#include "../kernels.hpp"
#include "../modarith.hpp"
#include "../ntt_device.hpp"

namespace aloha {
namespace k {

namespace {

constexpr int LINES = 16;          // lines per column tile
// Column-tile LDS image.  Default: pitch 257 words (odd => conflict-free 64-bit accesses), 32896 B, 4 workgroups per CU.
// ALOHA_COL5=1 (experiment): no padding -- 32768 B, FIVE workgroups fit the 160 KiB exactly -- and an XOR swizzle of the
// row index instead: g(col) = (col & 7) << 1 | (col >> 3) * 0b10001 keeps both mappings conflict-free (in the A
// mapping a half-wave varies row bit 0 and the column, in the B mapping row bit 4 and the column; bit 3 of the column
// is folded into both); the forward kernel then preloads 7 of its 15 second-step twiddles to stay within 96 VGPRs.
#ifndef ALOHA_COL5
#define ALOHA_COL5 0
#endif
#ifndef ALOHA_SETPRIO
#define ALOHA_SETPRIO 0
#endif
#if ALOHA_COL5
constexpr int PITCH = 256;
__device__ __forceinline__ uint32_t cswz(uint32_t col) { return ((col & 7) << 1) | ((col >> 3) * 17u); }
#else
constexpr int PITCH = 257;
__device__ __forceinline__ uint32_t cswz(uint32_t) { return 0; }
#endif
constexpr int TW_BLK = 256;        // twiddle entries per block

// Hooks of the fused single-launch transform (ntt_fused_fwd_kernel below): a tile body calls them where the
// workgroup's previous column tile may be signalled -- behind a barrier that every wave reaches with its stores
// drained.  The two-pass kernels pass NoHook.
struct NoHook {
  __device__ __forceinline__ void after_loads() const {}
  __device__ __forceinline__ void before_barrier() const {}
  __device__ __forceinline__ void after_barrier() const {}
  __device__ __forceinline__ void row_point() const {}
};

// 16-byte sc1 (write-through) stores: compiler-tracked raw buffer instructions with a wave-uniform base
__device__ __forceinline__ __amdgpu_buffer_rsrc_t limb_rsrc(const u64 *p) {
  const uint64_t a = (uint64_t)p;
  const uint64_t u = ((uint64_t)uni((uint32_t)(a >> 32)) << 32) | uni((uint32_t)a);
  return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, 0x7fffffff, 0x00020000);
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;          // cache-policy operand of the gfx940+ buffer builtins: bit 4 = sc1
// (ld_handoff: ntt_device.hpp)

// ---- column pass (N = 2^16): 16 adjacent columns x 256 rows per workgroup ---------------------
// A mapping: thread (col = tid & 15, pl = tid >> 4) holds rows pl + 16k   (stages 0..3)
// B mapping: thread (col = tid & 15, kh = tid >> 4) holds rows 16*kh + j  (stages 4..7)
// Both mappings touch memory in 128-byte row segments: forward loads in A and stores from B,
// inverse loads in B and stores from A.  One LDS exchange and one barrier per tile.  The
// stage 0..3 twiddles are the same for every thread (scalar loads -> SGPR operands).
// HANDOFF (fused transform): the tile's output is read by a row tile of the SAME launch on another CU, possibly
// another XCD, so it must leave the XCD's L2 as it is stored (sc1 = write-through) -- and 8-byte write-through
// stores are one fabric write each (2.7x the time per byte of 16-byte ones, MI355X guide; measured here:
// tools/ubench_icfuse.hip, 2120 vs 1243 us per 4096 limbs).  The B mapping therefore pairs lanes 16 apart on
// adjacent columns, col = 2*(pos & 7) + (rho & 1), kh = 4*wave + 2*(rho >> 1) + (pos >> 3) (rho = lane >> 4,
// pos = lane & 15), and after the second step v_permlane16_swap (gfx950: odd rows of X <-> even rows of Y, one
// instruction per dword) leaves each lane with BOTH columns of one row: 8 x 16-byte stores per lane.
template <bool FIRST, bool LAST, bool HANDOFF, class HOOK>
__device__ __forceinline__ void col_tile_fwd(u64 *sdata, const LimbJob &jb, const PrimeDev *pr, uint32_t cg, int tid,
                                             const HOOK &hook) {
  const uint32_t col = tid & 15, slow = tid >> 4;
  const uint32_t lane = tid & 63, rho = lane >> 4, pos = lane & 15;
  const uint32_t colB = HANDOFF ? 2 * (pos & 7) + (rho & 1) : col;
  const uint32_t khB = HANDOFF ? 4 * (tid >> 6) + 2 * (rho >> 1) + (pos >> 3) : slow;
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t blk = (ctw_t)pr->tw.fwd_blk;                                      // block 0
  cgptr_t src = (cgptr_t)jb.src + cg * 16;                                // wave-uniform bases
  gptr_t dst = (gptr_t)jb.dst + cg * 16;
  const uint32_t offA = (slow << 8) + col, offB = (khB << 12) + colB;     // 32-bit lane offsets
  u64 v[16];
  constexpr int BIN = FIRST ? FWD_IN : FWD_MID;
#if ALOHA_SETPRIO
  __builtin_amdgcn_s_setprio(3);        // (experiment) waves that are about to issue their loads go first
#endif
#pragma unroll
#if ALOHA_ABLATE == 2
  for (int kk = 0; kk < 16; kk++) v[kk] = (u64)offA * (kk + 3) + q;
#else
  for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&(src + ((size_t)(16 * kk) << 8))[offA]);
#endif
  // second-step twiddles: issued now (ordinary global loads keep their program order), so
  // their L2 latency hides under the first radix-16 step instead of stalling the second
  tw_t twb[15];
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      if (ALOHA_COL5 && (1 << ss) - 1 + g >= 7) continue;
      const gtw_t p = (gtw_t)blk + (16 << ss) + (khB << ss) + g;
      twb[(1 << ss) - 1 + g].w = p->w;
      twb[(1 << ss) - 1 + g].wp = p->wp;
    }
#if ALOHA_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  hook.after_loads();
  fwd_radix16<true, BIN>(v, TwStepA{blk}, q, nq);
#pragma unroll
  for (int kk = 0; kk < 16; kk++) sdata[col * PITCH + ((slow + 16 * kk) ^ cswz(col))] = v[kk];
  hook.before_barrier();
  __syncthreads();
  hook.after_barrier();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) v[jj] = sdata[colB * PITCH + ((16 * khB + jj) ^ cswz(colB))];
#if ALOHA_COL5
  fwd_radix16<false, bound_after4(BIN)>(v, TwStepMixT<7>{twb, blk, (int)khB}, q, nq);
#else
  fwd_radix16<false, bound_after4(BIN)>(v, TwStepReg{twb}, q, nq);
#endif
  if (HANDOFF) {
    const __amdgpu_buffer_rsrc_t rs = limb_rsrc(jb.dst);
    const uint32_t off0 = (((16 * khB + (rho & 1)) << 8) + cg * 16 + 2 * (pos & 7)) * 8;
#pragma unroll
    for (int jj = 0; jj < 16; jj += 2) {
      const u64 X = LAST ? canon_any(v[jj], q, nq, (u32)pr->m.kc) : v[jj];
      const u64 Y = LAST ? canon_any(v[jj + 1], q, nq, (u32)pr->m.kc) : v[jj + 1];
      const auto lo = __builtin_amdgcn_permlane16_swap(lo32(X), lo32(Y), false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(hi32(X), hi32(Y), false, false);
      // even rho: (own column, row jj), (column + 1, row jj);  odd rho: (column - 1, row jj + 1), (own column, row jj + 1)
      const u32x4_t x = {lo[0], hi[0], lo[1], hi[1]};
#if ALOHA_ABLATE == 2
      if (x[0] != 0x12345678u || x[3] != 0x9abcdef0u) continue;
#endif
      __builtin_amdgcn_raw_buffer_store_b128(x, rs, off0 + ((uint32_t)jj << 11), 0, AUX_SC1);
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < 16; jj++) {
#if ALOHA_ABLATE == 2
      if (v[jj] != 0x123456789abcull) continue;
#endif
      st_stream(&(dst + ((size_t)jj << 8))[offB], LAST ? canon_any(v[jj], q, nq, (u32)pr->m.kc) : v[jj]);
    }
  }
}

template <bool INV, bool FIRST, bool LAST, bool SC = false>     // SC: jobs may carry their own last-stage scalar (LimbJob::sc)
__global__ __launch_bounds__(256)
#if ALOHA_COL5
__attribute__((amdgpu_waves_per_eu(INV ? 4 : 5, 5)))
#endif
void ntt_col_kernel(const LimbJob *__restrict__ jobs,
                                                      const PrimeDev *__restrict__ primes, uint32_t count) {
  __shared__ u64 sdata[LINES * PITCH];
  const int tid = threadIdx.x;
  const uint32_t col = tid & 15, slow = tid >> 4;
  // XCD-affine tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share
  // one), so limb = 8*(b/128) + b%8, column group = (b/8)%16 puts the 16 column groups of a
  // limb -- which together cover each 2 KiB row -- on ONE XCD's L2.  Measured on the bare
  // access pattern: 6.05 vs 5.2 TB/s (tools/ubench_mem.hip).  Placement only affects speed;
  // the remainder limbs (count not a multiple of 8) use the plain order.
  const uint32_t b = blockIdx.x, full = (count >> 3) << 7;
  const uint32_t limb = b < full ? ((b >> 7) << 3) + (b & 7) : (b >> 4);
  const uint32_t cg = b < full ? (b >> 3) & 15 : (b & 15);
  const LimbJob jb = jobs[limb];
  const PrimeDev *pr = &primes[jb.prime];
  if (!INV) {
    col_tile_fwd<FIRST, LAST, false>(sdata, jb, pr, cg, tid, NoHook{});
  } else {
    const u64 q = pr->m.q, nq = pr->m.nq;
    ctw_t blk = (ctw_t)pr->tw.inv_blk;                                      // block 0
    cgptr_t src = (cgptr_t)jb.src + cg * 16;                                // wave-uniform bases
    gptr_t dst = (gptr_t)jb.dst + cg * 16;
    const uint32_t offA = (slow << 8) + col, offB = (slow << 12) + col;     // 32-bit lane offsets
    u64 v[16];
#ifndef ALOHA_INV_PRELOAD
#define ALOHA_INV_PRELOAD 15       // per-lane twiddles of the first inverse step issued before the data loads (0 / 8 / 15: 2.38 / 2.38 / 2.44 M inverse NTT/s)
#endif
    tw_t twb[15];
#pragma unroll
    for (int ss = 0; ss < 4; ss++)
#pragma unroll
      for (int g = 0; g < (1 << ss); g++)
        if ((1 << ss) - 1 + g >= 15 - ALOHA_INV_PRELOAD) {
          const gtw_t p = (gtw_t)blk + (16 << ss) + (slow << ss) + g;
          twb[(1 << ss) - 1 + g].w = p->w;
          twb[(1 << ss) - 1 + g].wp = p->wp;
        }
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = ld_stream(&(src + ((size_t)jj << 8))[offB]);
    inv_radix16<false, FIRST ? INV_IN : INV_MID, false>(v, TwStepMixHi<ALOHA_INV_PRELOAD>{twb, blk, (int)slow}, q, nq);
#pragma unroll
    for (int jj = 0; jj < 16; jj++) sdata[col * PITCH + ((16 * slow + jj) ^ cswz(col))] = v[jj];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) v[kk] = sdata[col * PITCH + ((slow + 16 * kk) ^ cswz(col))];
    const LastScale ls = last_scale<LAST && SC>(&jobs[limb], pr);
    inv_radix16<true, INV_MID, LAST>(v, TwStepA{blk}, q, nq, ls.w, ls.ni, ls.nip);
#pragma unroll
    for (int kk = 0; kk < 16; kk++)
      st_stream(&(dst + ((size_t)(16 * kk) << 8))[offA], LAST ? canon_last<SC>(v[kk], q, ls.addc) : v[kk]);
  }
}

// ---- column pass for N = 2^12 .. 2^15 (n1 = 4 + E column stages, E = 0..3) -------------------------
// A limb has R = 16 * 2^E rows of 256 columns.  A tile is R rows x C = 256 >> E columns (4096
// words): thread (col = tid mod C, pl = tid div C) first holds rows pl + 2^E * k (k = 0..15) for
// the radix-16 step over the stride-2^E .. stride-8*2^E stages (wave-uniform twiddles), then --
// for E > 0 -- after one LDS exchange it holds 16 >> E groups of 2^E consecutive rows,
//   row = ((kh * (16 >> E) + g) << E) + j,
// for the last E stages.  Loads and stores are whole 2^(11-E)-byte row segments.  The reference's
// own size, N = 8192, is E = 1.
template <int E, bool INV, bool FIRST, bool LAST, bool SC = false>
__global__ __launch_bounds__(256) void ntt_col_small_kernel(const LimbJob *__restrict__ jobs,
                                                            const PrimeDev *__restrict__ primes) {
  constexpr int R = 16 << E, C = 256 >> E, GR = 16 >> E, PIT = R + 1;
  __shared__ u64 sdata[E ? C * PIT : 1];
  const int tid = threadIdx.x;
  const uint32_t col = tid & (C - 1), hi = tid >> (8 - E);               // hi = pl (step A) = kh (step B)
  const uint32_t limb = blockIdx.x >> E, cg = blockIdx.x & ((1u << E) - 1);
  const LimbJob jb = jobs[limb];
  const PrimeDev *pr = &primes[jb.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t blk = (ctw_t)(INV ? pr->tw.inv_blk : pr->tw.fwd_blk);            // block 0 = W[0..255]
  cgptr_t src = (cgptr_t)jb.src + cg * C + col;
  gptr_t dst = (gptr_t)jb.dst + cg * C + col;
  auto rowA = [&](int k) -> uint32_t { return hi + ((uint32_t)k << E); };
  auto rowB = [&](int i) -> uint32_t { return (((hi * GR) + ((uint32_t)i >> E)) << E) + ((uint32_t)i & ((1u << E) - 1)); };
  // last E stages on the B-mapped registers: sub-stage t pairs i with i + (2^(E-1-t)) inside each
  // group of 2^E; twiddle L = 2^(4+t) + ((kh*GR + g) << t) + sub-group
  auto step_b = [&](u64 (&v)[16], auto bfly, bool fwd) {
#pragma unroll
    for (int tt = 0; tt < E; tt++) {
      const int t = fwd ? tt : E - 1 - tt;
      const int half = 1 << (E - 1 - t);
#pragma unroll
      for (int g = 0; g < GR; g++)
#pragma unroll
        for (int sg = 0; sg < (1 << t); sg++) {
          const tw_t w = load_tw(blk + (16 << t) + (((hi * GR) + g) << t) + sg);
#pragma unroll
          for (int jj = 0; jj < half; jj++) {
            const int i0 = (g << E) + sg * 2 * half + jj;
            bfly(v[i0], v[i0 + half], w, tt);
          }
        }
    }
  };
  u64 v[16];
  if (!INV) {
    constexpr int BIN = FIRST ? FWD_IN : FWD_MID;
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = ld_stream(&src[(size_t)rowA(k) << 8]);
    fwd_radix16<true, BIN>(v, TwStepA{blk}, q, nq);
    constexpr int B0 = bound_after4(BIN);
    constexpr int BS[4] = {B0, next_bound(B0), next_bound(next_bound(B0)), next_bound(next_bound(next_bound(B0)))};
    if (E) {
#pragma unroll
      for (int k = 0; k < 16; k++) sdata[col * PIT + rowA(k)] = v[k];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = sdata[col * PIT + rowB(i)];
      step_b(v, [&](u64 &X, u64 &Y, const tw_t &w, int tt) {
        if (need_corr(BS[tt])) X = csub_u(X, q * (u64)corr_t(BS[tt]));
        ct_bfly<false, false>(X, Y, w.w, w.wp, nq, q << 2);
      }, true);
    }
    constexpr int BOUT = BS[E];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      u64 x = v[i];
      if (LAST) x = canon_any(x, q, nq, (u32)pr->m.kc);
      else if (BOUT > FWD_MID) x = csub_u(x, q * (u64)corr_t(BOUT));   // the row pass expects < FWD_MID*q
      st_stream(&dst[(size_t)(E ? rowB(i) : rowA(i)) << 8], x);
    }
  } else {
    const u64 fourq = q << 2;
    if (E) {
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = ld_stream(&src[(size_t)rowB(i) << 8]);
      step_b(v, [&](u64 &X, u64 &Y, const tw_t &w, int) { gs_bfly<false>(X, Y, w.w, w.wp, nq, fourq); }, false);
#pragma unroll
      for (int i = 0; i < 16; i++) sdata[col * PIT + rowB(i)] = v[i];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = sdata[col * PIT + rowA(k)];
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = ld_stream(&src[(size_t)rowA(k) << 8]);
    }
    // (with E = 0 and FIRST the registers hold memory operands, < 2q < INV_MID*q)
    const LastScale ls = last_scale<LAST && SC>(&jobs[limb], pr);
    inv_radix16<true, INV_MID, LAST>(v, TwStepA{blk}, q, nq, ls.w, ls.ni, ls.nip);
#pragma unroll
    for (int k = 0; k < 16; k++)
      st_stream(&dst[(size_t)rowA(k) << 8], LAST ? canon_last<SC>(v[k], q, ls.addc) : v[k]);
  }
}

// ---- row pass: 16 lines of 256 contiguous points per workgroup, 4 lines per WAVE ------------------
// The four waves of a workgroup are independent: each owns 4 lines and a private LDS region,
// so the pass has no s_barrier at all (wave_lds_fence).  Lines of a tile are the same row of
// 2^pb limbs that share a prime (TileDesc), so with pb >= 2 a wave's four lines use ONE twiddle
// block: the first radix-16 step takes its 15 twiddles from SGPRs (scalar loads), the second
// reads its kh-dependent ones through the vector cache (16 distinct 16-byte words per load).
// A mapping: lane (pl = lane & 15, lw = lane >> 4) holds points pl + 16k  -- 128-byte segments
// B mapping: lane (lw = lane & 3, kh = lane >> 2) holds points 16*kh + j
// LDS slot of point p of wave-line lw: lw*RP + p + (p >> 4); RP = 280 (== 24 mod 32) makes the
// B-mapping 64-bit reads conflict-free.
// ALOHA_ROW_BMAP=1 (lw = lane >> 4, kh = lane & 15, RP = 272) makes BOTH mappings conflict-free
// (SQ_LDS_BANK_CONFLICT 33.5 M -> 0 cycles per launch, LDS active cycles -33 %) and was measured 0.8 %
// SLOWER end to end (2.50 vs 2.52 M NTT/s): the LDS is 24 % busy in this kernel, not a limiter.
#ifndef ALOHA_ROW_BMAP
#define ALOHA_ROW_BMAP 0
#endif
#if ALOHA_ROW_BMAP
constexpr int RP = 272;            // == 16 (mod 32)
#else
constexpr int RP = 280;
#endif

// Launch shape: one WAVE per workgroup (64 threads, blockIdx = tile*4 + wave): the waves share
// nothing, so giving each its own workgroup lets the dispatcher refill a finished wave's slot
// at once and lifts the occupancy limit from 4 to 5 waves per SIMD (LDS: 8960 B per wave).
#ifndef ALOHA_ROW_INV_WAVES
#define ALOHA_ROW_INV_WAVES 1   // 5 would hold the inverse instantiations to 96 VGPRs (they settle at 106-110: 4 waves per SIMD) at the price of 28 B of scratch: measured 2 % slower
#endif
template <bool INV, bool UNIA, bool FIRST, bool LAST, bool HANDOFF, class HOOK>
__device__ __forceinline__ void row_tile(u64 *lds, const LimbJob *__restrict__ jobs, const PrimeDev *__restrict__ primes,
                                         const TileDesc td, const uint32_t w, const int lane, const HOOK &hook) {
  const uint32_t first_limb = td.first_limb, pb_log = td.pb_log, chunk = td.chunk;
  const uint32_t pb_mask = (1u << pb_log) - 1, rpt = 16u >> pb_log;
  const LimbJob j0 = jobs[first_limb];
  const PrimeDev *pr = &primes[j0.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t gtw = (ctw_t)(INV ? pr->tw.inv_blk : pr->tw.fwd_blk) + TW_BLK;   // row blocks

  const uint32_t plA = lane & 15, lwA = lane >> 4;
#if ALOHA_ROW_BMAP
  const uint32_t lwB = lane >> 4, khB = lane & 15;
#else
  const uint32_t lwB = lane & 3, khB = lane >> 2;
#endif
  const uint32_t lineA = 4 * w + lwA, lineB = 4 * w + lwB;
  const uint32_t rowA = chunk * rpt + (lineA >> pb_log), rowB = chunk * rpt + (lineB >> pb_log);
  const uint32_t rowU = chunk * rpt + ((4 * w) >> pb_log);               // wave-uniform when pb_log >= 2
  ctw_t blkA = gtw + (size_t)(UNIA ? rowU : rowA) * TW_BLK;
  ctw_t blkB = gtw + (size_t)(UNIA ? rowU : rowB) * TW_BLK;
  const LimbJob jA = jobs[first_limb + (lineA & pb_mask)];
  cgptr_t src = (cgptr_t)jA.src + ((size_t)rowA << 8) + plA;
  gptr_t dst = (gptr_t)jA.dst + ((size_t)rowA << 8) + plA;
  const uint32_t ldsA = lwA * RP + plA, ldsB = lwB * RP + 17 * khB;

  u64 v[16];
#if ALOHA_SETPRIO
  __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
#if ALOHA_ABLATE == 2
  for (int kk = 0; kk < 16; kk++) v[kk] = (u64)ldsA * (kk + 3) + q;
#else
  for (int kk = 0; kk < 16; kk++) v[kk] = HANDOFF ? ld_handoff(&src[16 * kk]) : ld_stream(&src[16 * kk]);
#endif
#if ALOHA_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  if (!INV) {
    constexpr int BIN = FIRST ? FWD_IN : FWD_MID;
#ifndef ALOHA_ROW_PRELOAD
#define ALOHA_ROW_PRELOAD 7   // second-step twiddles issued before the first step: 0, 7 (sub-stages 0..2) or 15 (all)
#endif
    tw_t twb[15];
#pragma unroll
    for (int ss = 0; ss < 4; ss++)
#pragma unroll
      for (int g = 0; g < (1 << ss); g++)
        if ((1 << ss) - 1 + g < ALOHA_ROW_PRELOAD) {
          const gtw_t p = (gtw_t)blkB + (16 << ss) + (khB << ss) + g;
          twb[(1 << ss) - 1 + g].w = p->w;
          twb[(1 << ss) - 1 + g].wp = p->wp;
        }
    hook.after_loads();
    fwd_radix16<UNIA, BIN>(v, TwStepA{blkA}, q, nq);
    hook.row_point();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) lds[ldsA + 17 * kk] = v[kk];
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = lds[ldsB + jj];
#if ALOHA_ABLATE == 3   // timing experiment: second-step twiddles from the wave-uniform block entries
    fwd_radix16<false, bound_after4(BIN)>(v, TwStepA{blkA}, q, nq);
#else
    fwd_radix16<false, bound_after4(BIN)>(v, TwStepMixT<ALOHA_ROW_PRELOAD>{twb, blkB, (int)khB}, q, nq);
#endif
#if ALOHA_ABLATE != 5   // timing experiment 5: no final canonicalisation
    if (LAST) {
#pragma unroll
      for (int jj = 0; jj < 16; jj++) v[jj] = canon_any(v[jj], q, nq, (u32)pr->m.kc);
    }
#endif
  } else {
#ifndef ALOHA_INV_ROW_PRELOAD
#define ALOHA_INV_ROW_PRELOAD 15   // (with the column pass at 15: 0 / 12 / 15 = 2.44 / 2.45 / 2.48 M)
#endif
    tw_t twb[15];
#pragma unroll
    for (int ss = 0; ss < 4; ss++)
#pragma unroll
      for (int g = 0; g < (1 << ss); g++)
        if ((1 << ss) - 1 + g >= 15 - ALOHA_INV_ROW_PRELOAD) {
          const gtw_t p = (gtw_t)blkB + (16 << ss) + (khB << ss) + g;
          twb[(1 << ss) - 1 + g].w = p->w;
          twb[(1 << ss) - 1 + g].wp = p->wp;
        }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) lds[ldsA + 17 * kk] = v[kk];
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = lds[ldsB + jj];
    inv_radix16<false, FIRST ? INV_IN : INV_MID, false>(v, TwStepMixHi<ALOHA_INV_ROW_PRELOAD>{twb, blkB, (int)khB}, q, nq);
  }
  // back to the A mapping: second inverse step / 128-byte-segment stores
#if ALOHA_ABLATE != 4   // timing experiment 4: skip the transpose back (stores land in the wrong order)
  wave_lds_fence();
#pragma unroll
  for (int jj = 0; jj < 16; jj++) lds[ldsB + jj] = v[jj];
  wave_lds_fence();
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = lds[ldsA + 17 * kk];
#endif
  if (INV) {
    inv_radix16<UNIA, INV_MID, false>(v, TwStepA{blkA}, q, nq);
    if (LAST) {
      const u64 ni = pr->tw.n_inv, nip = pr->tw.n_inv_shoup;
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] = scale_canon(v[kk], ni, nip, q, nq);
    }
  }
#pragma unroll
  for (int kk = 0; kk < 16; kk++) {
#if ALOHA_ABLATE == 2
    if (v[kk] != 0x123456789abcull) continue;
#endif
    st_stream(&dst[16 * kk], v[kk]);
  }
}


template <bool INV, bool UNIA, bool FIRST, bool LAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INV ? ALOHA_ROW_INV_WAVES : 1))) void ntt_row_kernel(const LimbJob *__restrict__ jobs,
                                                     const PrimeDev *__restrict__ primes,
                                                     const TileDesc *__restrict__ tiles) {
  __shared__ u64 sdata[4 * RP];
  row_tile<INV, UNIA, FIRST, LAST, false>(sdata, jobs, primes, tiles[blockIdx.x >> 2], blockIdx.x & 3, threadIdx.x, NoHook{});
}

// ---- fused forward transform, N = 2^16: ONE persistent launch, intermediate handed over through the Infinity Cache ----
// The two passes above move the limb through HBM twice (measured 2.04x the algorithmic bytes).  Here both tile
// kinds run inside one launch: a column tile stores its 32 KiB write-through (sc1), and a row tile of the same
// limb, scheduled a few limbs later, loads it with sc1 loads -- by then every byte is at the memory side, where the
// 256 MiB Infinity Cache still holds it, so the intermediate costs fabric bandwidth but no HBM round trip
// (tools/ubench_icfuse.hip: 1200-1250 us per 4096 limbs against 1450 us for two launches of the same patterns).
// The whole transform is one pass over the machine as in the reference, where vntt runs to completion over the
// register file (src/vp/ntt/ntt_fsm.sv:68).
//
// Scheduling.  The host lays the tiles out as a ticket list (build_fused_schedule): column tiles of limb group r
// interleaved, in blocks of kFusedShards, with the row tiles of group r - D.  Persistent 256-thread workgroups draw
// positions from kFusedShards counters (workgroup b uses shard b mod kFusedShards; position = shard +
// kFusedShards * count: one counter would serialise at ~27 ns per draw, measured), three draws ahead of the
// tile in hand: the draw, the ticket load and the poll of the arrival counters each hide under a tile of their own.  A column tile ends by adding 1 to its limb's
// arrival counter (16 = complete); a row tile first checks the counters of its bundle's limbs.  Deadlock freedom:
// dependencies point to smaller positions, a workgroup handles its positions in increasing order, and a position
// is owned only by a running workgroup -- so the smallest unfinished position is always in the hands of a running
// workgroup whose dependencies are complete, whatever the residency.  Spins are bounded all the same; a give-up
// adds to *status (sticky, owned by the handle) and ends the launch (the host reports ALOHA_ERR_DEVICE at the next synchronisation).
// Visibility (MI355X guide, "valid forms"): every handed-over byte is stored sc1 and drained (s_waitcnt vmcnt(0) in
// every storing wave, then the workgroup barrier) before ONE lane adds to the counter; the consumer polls the
// counter with sc1 loads, passes a barrier, and loads every handed-over byte with global_load ... sc1 into registers.
// The signal of a column tile is deferred into the workgroup's next tile (behind the wait for that tile's own
// loads), so no wave ever stalls on its store drain; a workgroup that has to wait flushes its own signal first.
constexpr uint32_t kFusedStats = kFusedShards * 32;                       // [0] tiles that had to wait, [1] their poll iterations
constexpr uint32_t kFusedDone = (kFusedShards + 1) * 32;                  // first arrival counter
constexpr uint32_t kTicketExit = 0xfffffffeu, kTicketSkip = 0xffffffffu;

// (gu32_t: ntt_device.hpp)
__device__ __forceinline__ uint32_t ld_u32_sc1(gu32_t p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Signalling a column tile without a store-drain stall.  vmcnt decrements in issue order on gfx9-class targets
// (loads, stores and atomics share the counter; the compiler's own s_waitcnt placement relies on it), so once a wave
// has consumed data loads it issued AFTER a tile's stores, those stores have completed -- exactly what the
// `s_waitcnt vmcnt(0)` of the guide's hand-off recipe establishes, without draining the younger loads as well.
// The next tile of the workgroup therefore signals the previous column tile right after its first radix-16 step
// (which needs all 16 data loads): a column tile behind its exchange barrier (one lane, for all four waves); a
// row tile, which has no barrier, by the wave that arrives last at an LDS counter.  ALOHA_FUSED_DRAIN=1 restores the
// explicit drain (A/B; same results).
#ifndef ALOHA_FUSED_DRAIN
#define ALOHA_FUSED_DRAIN 0
#endif
template <class SCHED>
struct FusedHook {
  gu32_t done;
  uint32_t *pending;       // limb whose column tile of this workgroup is not signalled yet (0xffffffff: none); workgroup-uniform
  uint32_t *arrive;        // LDS word: waves that have passed a row tile's signalling point (monotonic, 4 per tile)
  int tid;
  const SCHED &sched;      // wave 0's scheduler stage: issued BEHIND the tile's own loads so that it never delays them
  __device__ __forceinline__ void signal() const {
    __hip_atomic_fetch_add(done + *pending * kFusedCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void after_loads() const { sched(); }
  __device__ __forceinline__ void before_barrier() const {
#if ALOHA_FUSED_DRAIN
    if (*pending != 0xffffffffu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __device__ __forceinline__ void after_barrier() const {
    if (*pending != 0xffffffffu) {
      if (tid == 0) signal();
      *pending = 0xffffffffu;
    }
  }
  __device__ __forceinline__ void row_point() const {
    if (*pending != 0xffffffffu) {
#if ALOHA_FUSED_DRAIN
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      if ((tid & 63) == 0 && (__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 3u) == 3u) signal();
      *pending = 0xffffffffu;
    }
  }
  __device__ __forceinline__ void flush() const {          // outside a tile: drain for real
    if (*pending != 0xffffffffu) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      after_barrier();
    }
  }
};

#ifndef ALOHA_FUSED_NOPOLL
#define ALOHA_FUSED_NOPOLL 0
#endif
#ifndef ALOHA_FUSED_EAGER
#define ALOHA_FUSED_EAGER 0
#endif
#ifndef ALOHA_FUSED_WAVES
#define ALOHA_FUSED_WAVES 3      // waves per SIMD the register budget is sized for (LDS admits 4 workgroups per CU; at 4 the kernel spills 17 dwords)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ALOHA_FUSED_WAVES, 4)))
void ntt_fused_fwd_kernel(const LimbJob *__restrict__ jobs, const LimbJob *__restrict__ inplace,
                          const PrimeDev *__restrict__ primes, const TileDesc *__restrict__ tiles,
                          const uint2 *__restrict__ tickets, uint32_t npos, uint32_t *ctl_, uint32_t *status_) {
  constexpr int LDS_WORDS = LINES * PITCH > 16 * RP ? LINES * PITCH : 16 * RP;
  __shared__ u64 sdata[LDS_WORDS];
  __shared__ uint32_t sh[2][4];
  __shared__ uint32_t arrive;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t shard = blockIdx.x % kFusedShards;
  gu32_t ctl = (gu32_t)ctl_, status = (gu32_t)status_;
  gu32_t counter = ctl + shard * 32, done = ctl + kFusedDone;
  uint32_t pending = 0xffffffffu;
  if (tid == 0) arrive = 0;
  // Wave 0 runs the scheduler as a pipeline, one stage per tile, so that nothing it waits for was issued less than
  // a whole tile ago: position draws three tickets ahead (lane 0), ticket loads two ahead (lane 0), arrival counters
  // of the next ticket's limbs one ahead (lanes < limbs of the bundle).  A ticket is {what, poll}: what = limb << 4 |
  // column group, or bit 31 + row-tile index; poll = first limb | log2(limbs) << 28 of a row tile's bundle.
  auto draw = [&]() { return shard + kFusedShards * __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto fetch = [&](uint32_t pos) { return pos < npos ? tickets[pos] : make_uint2(kTicketExit, 0u); };
  auto poll = [&](uint2 t) -> uint32_t {               // arrival count of this lane's limb of the bundle (16 = complete / nothing to wait for)
    if (t.x == kTicketExit || t.x == kTicketSkip || !(t.x >> 31)) return 16u;
    const uint32_t first = t.y & 0x0fffffffu, nl = 1u << (t.y >> 28);
    return (uint32_t)lane < nl ? ld_u32_sc1(done + (first + lane) * kFusedCtrStride) : 16u;
  };
  uint2 cur = make_uint2(kTicketExit, 0u), nxt = cur;
  uint32_t cnt = 16u, pos_a = 0;
  if (wave == 0) {
    uint2 t0 = cur;
    if (lane == 0) {
      const uint32_t p0 = draw(), p1 = draw();
      pos_a = draw();
      t0 = fetch(p0);
      nxt = fetch(p1);
    }
    cur = make_uint2(uni(t0.x), uni(t0.y));
    cnt = poll(cur);
  }
  // one scheduler stage: the ticket after the one in hand becomes current, its counters are polled, the next ticket
  // is fetched, a new position drawn.  Called once per loop iteration, behind the tile's own loads.
  const auto sched = [&]() {
    if (wave == 0 && cur.x != kTicketExit) {
      cur = make_uint2(uni(nxt.x), uni(nxt.y));
      cnt = poll(cur);
      if (lane == 0) {
        nxt = fetch(pos_a);
        pos_a = draw();
      }
    }
  };
  const FusedHook<decltype(sched)> hook{done, &pending, &arrive, tid, sched};
  for (uint32_t it = 0;; it++) {
    const uint32_t buf = it & 1;
    if (wave == 0 && lane == 0) {
      sh[buf][0] = cur.x; sh[buf][1] = cur.y;
    }
    if (wave == 0) {
      const uint32_t ready = __builtin_amdgcn_ballot_w64(cnt < 16u) == 0;
      if (lane == 0) sh[buf][2] = ready;
    }
    __syncthreads();
    const uint32_t tk = uni(sh[buf][0]), tpoll = uni(sh[buf][1]), ready = uni(sh[buf][2]);
    if (tk == kTicketExit || !ready) hook.flush();         // never wait -- or leave -- holding back a signal of one's own
    if (tk == kTicketExit) break;
    if (tk == kTicketSkip) { sched(); continue; }
    if (!ready) {
      if (wave == 0) {
        uint32_t spins = 0, ok = 0;
        for (;;) {
          const uint32_t c = poll(make_uint2(tk, tpoll));
          if (__builtin_amdgcn_ballot_w64(c < 16u) == 0) { ok = 1; break; }
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 16) || ((spins & 63) == 0 && ld_u32_sc1(status))) break;
        }
        if (lane == 0) {
          atomicAdd(ctl_ + kFusedStats, 1u);               // scheduling statistics (ALOHA_NTT_FUSED_STATS)
          atomicAdd(ctl_ + kFusedStats + 1, spins);
          if (!ok) __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sh[buf][3] = ok;
        }
      }
      __syncthreads();
      if (!uni(sh[buf][3])) break;                     // give up: the launch drains, the host sees the status word
    }
    if (!(tk >> 31)) {
      const uint32_t limb = tk >> 4, cg = tk & 15;
      const LimbJob jb = jobs[limb];
      col_tile_fwd<true, false, true>(sdata, jb, &primes[jb.prime], cg, tid, hook);
#if ALOHA_FUSED_EAGER            // A/B: signal at once (every wave stalls on its store drain) instead of inside the next tile
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(done + limb * kFusedCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      pending = limb;
#endif
    } else {
      row_tile<false, true, false, true, true>(sdata + wave * 4 * RP, inplace, primes, tiles[tk & 0x7fffffffu], (uint32_t)wave, lane, hook);
    }
    // (no barrier here: the next tile touches LDS only behind the barrier at the top of the loop, and sh[] is double-buffered)
  }
}

// The same hand-off with the HARDWARE dispatcher as scheduler: one workgroup per ticket, launched as an ordinary
// grid in ticket order.  No persistent loop, no draws, no scheduler pipeline; a finished workgroup's slot is refilled
// by the dispatcher while its neighbours compute, as in the two-launch kernels.  A row workgroup polls its bundle's
// arrival counters before its first load (the column tiles it needs lie `delay` groups back in the grid, so the first
// poll normally succeeds); a column workgroup drains its stores, passes a barrier and adds 1 to its limb's counter.
// Progress needs each XCD to start its workgroups in increasing blockIdx order (what the hardware does -- the guide:
// blocks are dealt round-robin over the XCDs -- though HIP does not promise it): then the smallest unfinished
// ticket is either running with all its dependencies complete, or next in line on an XCD whose resident workgroups
// all have smaller tickets and therefore finish.  Waits are bounded regardless; a give-up adds to *status.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ntt_fused_dispatch_kernel(const LimbJob *__restrict__ jobs, const LimbJob *__restrict__ inplace,
                               const PrimeDev *__restrict__ primes, const TileDesc *__restrict__ tiles,
                               const uint2 *__restrict__ tickets, uint32_t *ctl_, uint32_t *status_) {
  constexpr int LDS_WORDS = LINES * PITCH > 16 * RP ? LINES * PITCH : 16 * RP;
  __shared__ u64 sdata[LDS_WORDS];
  __shared__ uint32_t ok_flag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  gu32_t done = (gu32_t)ctl_ + kFusedDone, status = (gu32_t)status_;
  const uint2 tk = tickets[blockIdx.x];
  if (tk.x == kTicketSkip) return;
  if (!(tk.x >> 31)) {
    const uint32_t limb = tk.x >> 4, cg = tk.x & 15;
    const LimbJob jb = jobs[limb];
    col_tile_fwd<true, false, true>(sdata, jb, &primes[jb.prime], cg, tid, NoHook{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(done + limb * kFusedCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
#if ALOHA_FUSED_NOPOLL       // TIMING EXPERIMENT ONLY (results may be wrong): what would a free readiness check buy?
    if (false) {
#else
    if (wave == 0) {
#endif
      const uint32_t first = tk.y & 0x0fffffffu, nl = 1u << (tk.y >> 28);
      uint32_t spins = 0, ok = 0;
      for (;;) {
        const uint32_t c = (uint32_t)lane < nl ? ld_u32_sc1(done + (first + lane) * kFusedCtrStride) : 16u;
        if (__builtin_amdgcn_ballot_w64(c < 16u) == 0) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 16) || ((spins & 63) == 0 && ld_u32_sc1(status))) break;
      }
      if (lane == 0) {
        if (spins) { atomicAdd(ctl_ + kFusedStats, 1u); atomicAdd(ctl_ + kFusedStats + 1, spins); }
        if (!ok) __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_flag = ok;
      }
    }
#if !ALOHA_FUSED_NOPOLL
    __syncthreads();
    if (!ok_flag) return;
#endif
    row_tile<false, true, false, true, true>(sdata + wave * 4 * RP, inplace, primes, tiles[tk.x & 0x7fffffffu], (uint32_t)wave, lane, NoHook{});
  }
}

// ---- radix-2 stage kernels (any N; column passes of N < 2^16; bring-up cross-check) ---------
// Forward stage `s` (m = 2^s groups, t = N/2m): canonical in, canonical out.
__global__ __launch_bounds__(256) void ntt_radix2_fwd_stage(const LimbJob *__restrict__ jobs,
                                                            const PrimeDev *__restrict__ primes,
                                                            unsigned logn, unsigned s, int first) {
  const LimbJob jb = jobs[blockIdx.y];
  const PrimeDev *pr = &primes[jb.prime];
  const u64 q = pr->m.q;
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;                      // butterfly index < N/2
  const uint32_t t = (1u << logn) >> (s + 1);
  const uint32_t i = b / t, jj = b % t, j = 2 * i * t + jj;
  const tw_t tw = reinterpret_cast<const tw_t *>(pr->tw.fwd)[(1u << s) + i];
  const u64 *src = first ? jb.src : jb.dst;
  u64 U = src[j], V = src[j + t];
  if (first) { U = csub(U, q); V = csub(V, q); }
  V = csub(mul_shoup_2q(V, tw.w, tw.wp, q), q);
  jb.dst[j] = addmod(U, V, q);
  jb.dst[j + t] = submod(U, V, q);
}

// Inverse stage with t = 2^u (h = N/2t groups): inputs < 4q, canonical out, no halving; the
// final stage (scale = 1) multiplies both outputs by N^-1.
__global__ __launch_bounds__(256) void ntt_radix2_inv_stage(const LimbJob *__restrict__ jobs,
                                                            const PrimeDev *__restrict__ primes,
                                                            unsigned logn, unsigned u, int first, int scale) {
  const LimbJob jb = jobs[blockIdx.y];
  const PrimeDev *pr = &primes[jb.prime];
  const u64 q = pr->m.q;
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  const uint32_t t = 1u << u, h = (1u << logn) >> (u + 1);
  const uint32_t i = b / t, jj = b % t, j = 2 * i * t + jj;
  const tw_t tw = reinterpret_cast<const tw_t *>(pr->tw.inv)[h + i];
  const u64 *src = first ? jb.src : jb.dst;
  // the first stage may read the row pass's intermediate format (< 4q); later stages read canonical data
  u64 U = csub(csub(src[j], q << 1), q), V = csub(csub(src[j + t], q << 1), q);
  u64 s0 = addmod(U, V, q);
  u64 s1 = csub(mul_shoup_2q(submod(U, V, q), tw.w, tw.wp, q), q);
  if (scale) {
    s0 = csub(mul_shoup_2q(s0, pr->tw.n_inv, pr->tw.n_inv_shoup, q), q);
    s1 = csub(mul_shoup_2q(s1, pr->tw.n_inv, pr->tw.n_inv_shoup, q), q);
  }
  jb.dst[j] = s0;
  jb.dst[j + t] = s1;
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------
void build_row_tiles(const LimbJob *h_jobs, uint32_t begin, uint32_t count, unsigned log_rows,
                     std::vector<TileDesc> &uni_tiles, std::vector<TileDesc> &glb_tiles) {
  const uint32_t rows = 1u << log_rows, end = begin + count;
  uint32_t i = begin;
  while (i < end) {
    uint32_t run = 1;
    while (i + run < end && h_jobs[i + run].prime == h_jobs[i].prime) run++;
    uint32_t done = 0;
    while (done < run) {
      uint32_t pb_log = 4;
      while ((1u << pb_log) > run - done) pb_log--;
      const uint32_t rpt = 16u >> pb_log;
      if (rpt > rows) {                     // fewer rows than a tile holds: no tile form
        uni_tiles.clear(); glb_tiles.clear();
        return;
      }
      // bundles of >= 4 limbs give every wave (4 lines) a single twiddle block
      std::vector<TileDesc> &dst = (pb_log >= 2) ? uni_tiles : glb_tiles;
      for (uint32_t c = 0; c < rows / rpt; c++) dst.push_back(TileDesc{i + done, (uint16_t)pb_log, (uint16_t)c});
      done += 1u << pb_log;
    }
    i += run;
  }
}

template <bool INV, bool FIRST, bool LAST, bool SC = false>
static void launch_cols(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                        uint64_t *launches) {
  if (!count) return;
  hipLaunchKernelGGL((ntt_col_kernel<INV, FIRST, LAST, SC>), dim3(count * 16), dim3(256), 0, st, d_jobs, d_primes, count);
  if (launches) ++*launches;
}

template <bool INV, bool FIRST, bool LAST, bool SC = false>
static void launch_cols_small(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                              unsigned e, uint64_t *launches) {
  if (!count) return;
  switch (e) {
#define ALOHA_COL_SMALL(EE)                                                                                  \
  case EE:                                                                                                   \
    hipLaunchKernelGGL((ntt_col_small_kernel<EE, INV, FIRST, LAST, SC>), dim3(count << EE), dim3(256), 0, st, d_jobs, \
                       d_primes);                                                                            \
    break;
    ALOHA_COL_SMALL(0) ALOHA_COL_SMALL(1) ALOHA_COL_SMALL(2) ALOHA_COL_SMALL(3)
#undef ALOHA_COL_SMALL
  }
  if (launches) ++*launches;
}

template <bool INV, bool FIRST, bool LAST>
static void launch_rows(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes,
                        const TileDesc *d_tiles, uint32_t ntiles, bool uniform, uint64_t *launches) {
  if (!ntiles) return;
  if (uniform)
    hipLaunchKernelGGL((ntt_row_kernel<INV, true, FIRST, LAST>), dim3(ntiles * 4), dim3(64), 0, st, d_jobs,
                       d_primes, d_tiles);
  else
    hipLaunchKernelGGL((ntt_row_kernel<INV, false, FIRST, LAST>), dim3(ntiles * 4), dim3(64), 0, st, d_jobs,
                       d_primes, d_tiles);
  if (launches) ++*launches;
}

// Ticket list of the fused transform (see ntt_fused_fwd_kernel).  `uni_tiles` = the row tiles of the whole batch as
// build_row_tiles emits them (bundle by bundle; every bundle has >= 4 limbs).  Limbs are grouped bundle-wise into
// groups of >= 16; round r interleaves, in blocks of kFusedShards positions, the column tickets of group r with the
// row tickets of group r - delay (short blocks are padded with skip tickets so that a shard keeps alternating).
void build_fused_schedule(const std::vector<TileDesc> &uni_tiles, unsigned log_rows, unsigned delay,
                          std::vector<uint64_t> &tickets) {
  tickets.clear();
  struct Group { std::vector<uint64_t> cols, rows; };
  std::vector<Group> groups;
  size_t t = 0;
  uint32_t in_group = 0;
  while (t < uni_tiles.size()) {
    const TileDesc &td = uni_tiles[t];
    const uint32_t nl = 1u << td.pb_log, ntiles = (1u << log_rows) / (16u >> td.pb_log);
    if (groups.empty() || in_group >= 16) { groups.emplace_back(); in_group = 0; }
    Group &g = groups.back();
    // column tickets of a bundle in XCD-affine order where the bundle allows it: positions are dealt round-robin over the 8
    // XCDs, so limb = 8*half + (k mod 8) puts the 16 column tiles of a limb -- which together cover each 2 KiB row -- on
    // ONE XCD (as in ntt_col_kernel: 6.05 vs 5.2 TB/s on the bare pattern); smaller bundles keep the plain order
    if (nl % 8 == 0) {
      for (uint32_t h = 0; h < nl / 8; h++)
        for (uint32_t cg = 0; cg < 16; cg++)
          for (uint32_t l = 0; l < 8; l++) g.cols.push_back(((td.first_limb + 8 * h + l) << 4) | cg);
    } else {
      for (uint32_t l = 0; l < nl; l++)
        for (uint32_t cg = 0; cg < 16; cg++) g.cols.push_back(((td.first_limb + l) << 4) | cg);
    }
    const uint64_t pollw = (uint64_t)(td.first_limb | ((uint32_t)td.pb_log << 28)) << 32;     // high word: the bundle to poll
    for (uint32_t i = 0; i < ntiles; i++) g.rows.push_back(pollw | 0x80000000u | (uint32_t)(t + i));
    in_group += nl;
    t += ntiles;
  }
  const size_t G = groups.size();
  for (size_t r = 0; r < G + delay; r++) {
    const std::vector<uint64_t> empty;
    const std::vector<uint64_t> &C = r < G ? groups[r].cols : empty;
    const std::vector<uint64_t> &R = r >= delay ? groups[r - delay].rows : empty;
    size_t ci = 0, ri = 0;
    while (ci < C.size() || ri < R.size()) {
      for (const std::vector<uint64_t> *src : {&C, &R}) {
        size_t &i = src == &C ? ci : ri;
        if (i >= src->size()) continue;
        const size_t n = std::min<size_t>(kFusedShards, src->size() - i);
        tickets.insert(tickets.end(), src->begin() + i, src->begin() + i + n);
        tickets.insert(tickets.end(), kFusedShards - n, (uint64_t)0xffffffffu);
        i += n;
      }
    }
  }
}

size_t fused_ctl_words(uint32_t count) { return (size_t)(kFusedShards + 1) * 32 + (size_t)count * kFusedCtrStride; }

uint32_t fused_workgroups_per_cu() {
  static int per_cu = -1;
  if (per_cu < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ntt_fused_fwd_kernel, 256, 0) != hipSuccess || n < 1) n = 1;
    per_cu = n;
  }
  return (uint32_t)per_cu;
}

hipError_t ntt_fused_fwd(hipStream_t st, const LimbJob *d_jobs, const LimbJob *d_inplace, const PrimeDev *d_primes,
                         const TileDesc *d_tiles, const uint64_t *d_tickets, uint32_t npos, uint32_t count,
                         uint32_t *d_ctl, uint32_t *d_status, uint32_t workgroups, uint64_t *launches) {
  hipError_t e = hipMemsetAsync(d_ctl, 0, fused_ctl_words(count) * 4, st);
  if (e != hipSuccess) return e;
  if (workgroups == 0)        // one workgroup per ticket, scheduled by the hardware dispatcher
    hipLaunchKernelGGL(ntt_fused_dispatch_kernel, dim3(npos), dim3(256), 0, st, d_jobs, d_inplace, d_primes, d_tiles,
                       (const uint2 *)d_tickets, d_ctl, d_status);
  else
    hipLaunchKernelGGL(ntt_fused_fwd_kernel, dim3(workgroups), dim3(256), 0, st, d_jobs, d_inplace, d_primes, d_tiles,
                       (const uint2 *)d_tickets, npos, d_ctl, d_status);
  if (launches) ++*launches;
  return hipGetLastError();
}

hipError_t ntt_radix2_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                            unsigned logn, unsigned stage_begin, unsigned stage_end, bool inverse,
                            bool first_reads_src, bool last_scales, uint64_t *launches) {
  // forward: stages s = stage_begin .. stage_end-1 (s = 0 is the stride-N/2 stage)
  // inverse: stages u = stage_begin .. stage_end-1 (u = 0 is the stride-1 stage)
  // the job index rides in gridDim.y (<= 65535): larger batches go stage by stage in chunks of that many limbs
  constexpr uint32_t kMaxY = 65535;
  for (unsigned s = stage_begin; s < stage_end; s++) {
    const int first = (first_reads_src && s == stage_begin) ? 1 : 0;
    for (uint32_t off = 0; off < count; off += kMaxY) {
      const dim3 grid((1u << logn) / 2 / 256, std::min(kMaxY, count - off)), blk(256);
      if (!inverse)
        hipLaunchKernelGGL(ntt_radix2_fwd_stage, grid, blk, 0, st, d_jobs + off, d_primes, logn, s, first);
      else
        hipLaunchKernelGGL(ntt_radix2_inv_stage, grid, blk, 0, st, d_jobs + off, d_primes, logn, s, first,
                           (last_scales && s + 1 == stage_end) ? 1 : 0);
      if (launches) ++*launches;
    }
  }
  return hipGetLastError();
}

hipError_t ntt_tile_batch(hipStream_t st, const LimbJob *d_jobs, const LimbJob *inplace, const PrimeDev *d_primes,
                          uint32_t begin, uint32_t count, unsigned logn, bool inverse, const TileDesc *d_lds_tiles,
                          uint32_t n_lds, const TileDesc *d_glb_tiles, uint32_t n_glb, uint64_t *launches,
                          unsigned single_max, bool scaled) {
  if (logn < 10 || logn > 16) return hipErrorInvalidValue;
  // N = 2^12, 2^13: the limb fits one CU's LDS several times over -> one pass over memory
  // (ntt_single.hip: 3.0 / 2.6 TB/s algorithmic against 2.6 / 2.5 for the tile form below).  At
  // N = 2^14 the single-pass kernel exists and is exact but slower (one 1024-thread workgroup per CU:
  // 1.5 vs 2.4 TB/s), so it is off by default.  `single_max` (ALOHA_FLAG_NTT_TWO_PASS /
  // ALOHA_FLAG_NTT_SINGLE_14 of the handle) moves the limit: A/B timing and the on-device cross-check
  // of the two forms in the tests.
  if (logn >= 12 && logn <= 14 && logn <= single_max)
    return ntt_single_batch(st, d_jobs + begin, d_primes, count, logn, inverse, launches, scaled);
  const unsigned n1 = logn - 8;
  const bool col_tiles = (n1 == 8);
  const bool col_small = (n1 >= 4 && n1 <= 7);
  if (!inverse) {
    if (col_tiles || col_small) {
      if (col_tiles) launch_cols<false, true, false>(st, d_jobs + begin, d_primes, count, launches);
      else launch_cols_small<false, true, false>(st, d_jobs + begin, d_primes, count, n1 - 4, launches);
      launch_rows<false, false, true>(st, inplace, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_rows<false, false, true>(st, inplace, d_primes, d_glb_tiles, n_glb, false, launches);
    } else {
      hipError_t e = ntt_radix2_batch(st, d_jobs + begin, d_primes, count, logn, 0, n1, false, true, false, launches);
      if (e != hipSuccess) return e;
      // the radix-2 stages left canonical data in dst; the row tiles run in place on dst
      launch_rows<false, true, true>(st, inplace, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_rows<false, true, true>(st, inplace, d_primes, d_glb_tiles, n_glb, false, launches);
    }
  } else {
    launch_rows<true, true, false>(st, d_jobs, d_primes, d_lds_tiles, n_lds, true, launches);
    launch_rows<true, true, false>(st, d_jobs, d_primes, d_glb_tiles, n_glb, false, launches);
    if (col_tiles) {
      if (scaled) launch_cols<true, false, true, true>(st, inplace + begin, d_primes, count, launches);
      else launch_cols<true, false, true>(st, inplace + begin, d_primes, count, launches);
    } else if (col_small) {
      if (scaled) launch_cols_small<true, false, true, true>(st, inplace + begin, d_primes, count, n1 - 4, launches);
      else launch_cols_small<true, false, true>(st, inplace + begin, d_primes, count, n1 - 4, launches);
    } else {
      hipError_t e = ntt_radix2_batch(st, inplace + begin, d_primes, count, logn, 8, logn, true, true, true, launches);
      if (e != hipSuccess) return e;
    }
  }
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
