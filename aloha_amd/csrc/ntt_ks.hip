// ntt_ks.hip -- the forward transforms of a key-switch at N = 2^16 fused with what surrounds them (gfx950).
//
// The reference's key-switch (sim/vp/isram_file_generator/keyswitch.mem:1-122) is a chain of whole-limb instructions:
// basis extension of a digit (:9-14,:29-34), vntt of the extended limb, product with the key and accumulation over the
// digits (:46-79), and in the mod-down a basis change (:86-91), vntt (:94-99) and (acc - t) * P^-1 + base (:102-117).
// Run instruction by instruction every intermediate limb makes a round trip through HBM.  The sequencer
// (vp_engine.cpp) recognises these chains in a program and hands them to the kernels below, which are the two-pass
// transform of ntt_tile.hip with the neighbouring element-wise work folded into its passes:
//   ntt_col_pro_kernel   column pass whose input is  sum_t a_t * s_t + cst  formed in its loads        (SumSrc)
//   ntt_row_epi_kernel   row pass that ends in  t * s0 + sum_i a_i * s_i + cst  instead of storing t    (RowEpi)
//   ntt_rowdot_kernel    row pass of all digits of one modulus, times the key rows, summed over the digits:
//                        two limbs leave where dnum transformed limbs would have been written and read back (RowDotGroup)
// All results are the canonical residues the instruction-by-instruction execution produces (the transforms are exact
// and every sum is reduced to [0, q) at the end): bit-exact against the oracle's ISA machine in tests/.
#include <cstdlib>

#include "kernels.hpp"
#include "modarith.hpp"
#include "ntt_device.hpp"
#include "ntt_tiles.hpp"

namespace aloha {
namespace k {

namespace {

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_col_pro_kernel(const LimbJob *__restrict__ jobs, const PrimeDev *__restrict__ primes, uint32_t count,
                                                          const SumSrc *__restrict__ sums) {
  __shared__ u64 sdata[LINES * PITCH];
  // XCD-affine tile order, as in ntt_col_kernel
  const uint32_t b = blockIdx.x, full = (count >> 3) << 7;
  const uint32_t limb = b < full ? ((b >> 7) << 3) + (b & 7) : (b >> 4);
  const uint32_t cg = b < full ? (b >> 3) & 15 : (b & 15);
  const LimbJob jb = jobs[limb];
  col_tile_fwd<true, false, true>(sdata, jb, &primes[jb.prime], cg, threadIdx.x, sums);
}

// Row pass (second pass of a forward transform) whose lines may end in an epilogue.  A line with an epilogue is not
// canonicalised first: its Shoup product takes any 64-bit value.
// (amdgpu_waves_per_eu(4) brings the <false> form from 136 to 128 VGPRs and four waves per SIMD without scratch: measured within
// the noise, HMult 0.2703 against 0.2706 ms -- profiles/r04_ks_kernel_experiments.txt; again at settled clocks, ALOHA_EPI_WAVES=4:
// HMult 0.2406 against 0.2391 ms, rotate 0.3155 against 0.3178)
#ifndef ALOHA_EPI_WAVES
#define ALOHA_EPI_WAVES 1
#endif
template <bool UNIA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ALOHA_EPI_WAVES))) void ntt_row_epi_kernel(const LimbJob *__restrict__ jobs,
                                                                                                 const PrimeDev *__restrict__ primes,
                                                                                                 const TileDesc *__restrict__ tiles,
                                                                                                 const RowEpi *__restrict__ epis) {
  __shared__ u64 lds[4 * RP];
  const TileDesc td = tiles[blockIdx.x >> 2];
  const uint32_t w = blockIdx.x & 3;
  const int lane = threadIdx.x;
  const uint32_t first_limb = td.first_limb, pb_log = td.pb_log, chunk = td.chunk;
  const uint32_t pb_mask = (1u << pb_log) - 1, rpt = 16u >> pb_log;
  const LimbJob j0 = jobs[first_limb];
  const PrimeDev *pr = &primes[j0.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  const u32 kc = (u32)pr->m.kc;
  ctw_t gtw = (ctw_t)pr->tw.fwd_blk + TW_BLK;
  const RowLanes L(lane);
  const uint32_t lineA = 4 * w + L.lwA, lineB = 4 * w + L.lwB;
  const uint32_t rowA = chunk * rpt + (lineA >> pb_log), rowB = chunk * rpt + (lineB >> pb_log);
  const uint32_t rowU = chunk * rpt + ((4 * w) >> pb_log);
  ctw_t blkA = gtw + (size_t)(UNIA ? rowU : rowA) * TW_BLK;
  ctw_t blkB = gtw + (size_t)(UNIA ? rowU : rowB) * TW_BLK;
  const LimbJob jA = jobs[first_limb + (lineA & pb_mask)];
  const uint32_t off = (rowA << 8) + L.plA;
  cgptr_t src = (cgptr_t)jA.src + off;
  gptr_t dst = (gptr_t)jA.dst + off;
  // the epilogue's scalars and operand pointers travel with the line: loaded now, behind the data loads
  const RowEpi *e = epis + (jA.epi ? jA.epi - 1 : 0);
  const bool has = jA.epi != 0;                                  // (per line: the 16 lanes of a line agree)
  u64 es[kEpiOps + 1], esp[kEpiOps + 1], cst = 0;
  cgptr_t ea[kEpiOps];
  gptr_t out = dst;
  uint32_t nops = 0, store_t = 1;
  u64 v[16];
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&src[16 * kk]);
  if (has) {
#pragma unroll
    for (int i = 0; i <= kEpiOps; i++) { es[i] = e->s[i]; esp[i] = e->sp[i]; }
#pragma unroll
    for (int i = 0; i < kEpiOps; i++) ea[i] = (cgptr_t)e->a[i] + off;
    cst = e->cst; out = (gptr_t)e->out + off; nops = e->nops; store_t = e->store_t;
  }
  row_fwd_core<UNIA, FWD_MID, false>(v, lds, L, blkA, blkB, pr);
  if (has) {
    // the first two operands' loads go out together, ahead of the product with the transform.  (The three products on shared
    // multiply-add chains, as in the column pass's prologue: measured in round 4, 0.2672 against 0.2643 ms -- this kernel waits for
    // memory at three waves per SIMD, its instructions are not what it costs; profiles/r04_ks_kernel_experiments.txt)
    u64 x0[16], x1[16];
    if (nops > 0) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) x0[kk] = ea[0][16 * kk];
    }
    if (nops > 1) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) x1[kk] = ea[1][16 * kk];
    }
    if (store_t) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) st_stream(&dst[16 * kk], canon_any(v[kk], q, nq, kc));
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) v[kk] = mul_shoup_4q<false>(v[kk], es[0], esp[0], nq);       // < 4q
    if (nops > 0) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] += mul_shoup_4q<false>(x0[kk], es[1], esp[1], nq);
    }
    if (nops > 1) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] += mul_shoup_4q<false>(x1[kk], es[2], esp[2], nq);
    }
    if (nops > 2) {                                              // <= 4 products < 4q: < 16q
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] += mul_shoup_4q<false>(ea[2][16 * kk], es[3], esp[3], nq);
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) st_stream(&out[16 * kk], csub_u(canon_any(v[kk], q, nq, kc) + cst, q));
  } else {
#pragma unroll
    for (int kk = 0; kk < 16; kk++) st_stream(&dst[16 * kk], canon_any(v[kk], q, nq, kc));
  }
}

// ---- row pass + inner product with the key -----------------------------------------------------------------------
// One wave per (group, R consecutive rows); group = one modulus of one ciphertext (RowDotGroup).  The wave walks the
// group's digits and keeps the two accumulators of its R rows in registers:
//   * four digits at a time as a row tile of the shape the plain row pass uses for bundles of 4 limbs -- 4 lines = the
//     SAME row of 4 digits, one twiddle block through scalar loads -- once per row;
//   * the remaining 1-3 transformed digits, and the digits that arrive in NTT form already (the limb to switch under
//     its own digit's moduli), as tiles of 4 / R digits x R rows.
// After a tile a lane (pl, lw) holds line lw's 16 points pl + 16k.  They are multiplied by the key's words at the same
// positions (mulmod_lazy_2q, < 2q) and then leave the lane: v_permlane32_swap / v_permlane16_swap (gfx950) exchange
// register pairs between the lane groups 32 and 16 apart, so that two swaps turn "16 points of one line" into
// "4 points of four lines" -- ADDED where the lines exchanged are digits of one row (a reduce-scatter), kept apart where
// they are rows of one digit (a transpose).  Either way lane (pl, lw) ends with the 4 points
//   k = 4m + 2(lw & 1) + (lw >> 1),  m = 0..3
// of each of the wave's rows: R x 4 x 2 accumulators, lazily < 8q; the stores at the end are 128-byte segments again.
// R = 4: 168 VGPRs, 3 waves per SIMD, no idle lines.  R = 2: half the accumulators and no twiddle preload -> 4 waves per
// SIMD and twice the waves (a key-switch has few groups: 28 at L = 24), an odd leftover digit leaves two lines idle.
__device__ __forceinline__ void swap32(u64 &x, u64 &y) {        // lanes >= 32 of x  <->  lanes < 32 of y
  const auto lo = __builtin_amdgcn_permlane32_swap(lo32(x), lo32(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(hi32(x), hi32(y), false, false);
  x = pack64(lo[0], hi[0]);
  y = pack64(lo[1], hi[1]);
}
__device__ __forceinline__ void swap16(u64 &x, u64 &y) {        // odd 16-lane rows of x  <->  even rows of y
  const auto lo = __builtin_amdgcn_permlane16_swap(lo32(x), lo32(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(hi32(x), hi32(y), false, false);
  x = pack64(lo[0], hi[0]);
  y = pack64(lo[1], hi[1]);
}

// Line lw of a tile = (digit index lw >> DSH, row index lw & ((1 << DSH) - 1)):  DSH = 0: four digits of one row;
// DSH = 1: two digits x two rows;  DSH = 2: one digit x four rows.  The exchange over lane bit 1 of lw (32 apart) adds
// when that bit selects a digit (DSH <= 1), the exchange over bit 0 (16 apart) adds when it does (DSH == 0).
// products of the lane's points 4m .. 4m+3 with the two key rows (`live`: lines without a digit contribute nothing), then
// the exchange: r[i][c] = row i's contribution at the lane's point of chunk m.
template <int DSH>
__device__ __forceinline__ void products_chunk(const u64 (&v)[16], int m, cgptr_t key0, cgptr_t key1, bool live, u64 nq, u64 mu,
                                               u64 (&r)[1 << DSH][2]) {
  u64 p[4][2];
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      p[j][0] = key0[16 * (4 * m + j)];
      p[j][1] = key1[16 * (4 * m + j)];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int c = 0; c < 2; c++) p[j][c] = mulmod_lazy_2q(v[4 * m + j], p[j][c], nq, mu);
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) p[j][0] = p[j][1] = 0;
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    // lanes 32 apart, pairs (4m, 4m+1) and (4m+2, 4m+3): a lane keeps k = k0 + (lw >> 1); x <-> lw bit 1 clear, y <-> set
    swap32(p[0][c], p[1][c]);
    swap32(p[2][c], p[3][c]);
    if (DSH <= 1) {
      u64 s0 = p[0][c] + p[1][c], s1 = p[2][c] + p[3][c];       // < 4q
      swap16(s0, s1);                                           // lanes 16 apart: a lane keeps j = 2m + (lw & 1); x <-> lw bit 0 clear
      if (DSH == 0) r[0][c] = s0 + s1;                          // < 8q
      else { r[0][c] = s0; r[1][c] = s1; }
    } else {
      swap16(p[0][c], p[2][c]);                                 // rows 0 | 1
      swap16(p[1][c], p[3][c]);                                 // rows 2 | 3
      r[0][c] = p[0][c]; r[1][c] = p[2][c]; r[2][c] = p[1][c]; r[3][c] = p[3][c];
    }
  }
}

// (Round 4, measured and dropped: touching the lines of a tile's key rows -- one dword per 128-byte line and lane, between the two
// radix-16 steps of the tile's transform -- so that the products' loads find them in L2 / the Infinity Cache: HMult 0.283 against
// 0.272 ms, rotate 0.401 against 0.393: the extra requests cost more than the latency they hide, profiles/r04_row_lds_layout_ab.txt.)
// (Round 4, built, bit-exact, measured and removed -- profiles/r04_ks_kernel_experiments.txt: the FIRST pass of the special-prime
// accumulators' inverse transforms as a tail of this kernel (the wave holds complete rows of both accumulators; phases 2 and 3 as one
// program so that the sequencer sees the chain): one launch and 9 us of latency-bound row pass less, and HMult 0.2794 against 0.2766 ms,
// rotate 0.4008 against 0.3953 on one box -- the tail's waves are the last to finish and its code costs the whole kernel ~5 us.)
// S waves per workgroup share the digits of one (group, R rows): each takes a contiguous share of the blocks, wave 0 adds the
// accumulators of the others (through LDS) and stores.  A wave's chain of tiles is serial -- load, transform, load the
// key, multiply -- and a key-switch has few groups, so at S = 1 the kernel's time was one wave's latency (the same ~100 us for
// 18 groups as for 28); S = 2 halves the chain and doubles the waves.
template <int R, int S>
__global__ __launch_bounds__(64 * S) __attribute__((amdgpu_waves_per_eu(R == 4 ? 3 : 4, R == 4 ? 3 : 4))) void ntt_rowdot_kernel(
    const RowDotGroup *__restrict__ groups, const PrimeDev *__restrict__ primes) {
  static_assert(R == 1 || R == 2 || R == 4, "rows per wave");
  constexpr int DSH = R == 4 ? 2 : R == 2 ? 1 : 0, DPT = 4 / R;  // remainder tiles: DPT digits x R rows
  constexpr int PRE = R == 2 ? 0 : 3;                            // second-step twiddles issued early (registers against latency)
  constexpr uint32_t WPG = 256 / R;                              // workgroups per group
  __shared__ u64 lds_all[S * 4 * RP];
  static_assert(S == 1 || R * 8 * 64 <= 4 * RP, "the cross-wave reduction writes R*4*2*64 words into a wave's transpose region");
  const uint32_t wv = S > 1 ? uni(threadIdx.x >> 6) : 0;
  u64 *lds = lds_all + wv * 4 * RP;
  const RowDotGroup *g = &groups[blockIdx.x / WPG];
  const uint32_t r0 = (blockIdx.x % WPG) * R;
  const int lane = threadIdx.x & 63;
  const PrimeDev *pr = &primes[g->prime];
  const u64 q = pr->m.q, nq = pr->m.nq, mu = pr->m.mu;
  const u32 kc = (u32)pr->m.kc;
  ctw_t gtw = (ctw_t)pr->tw.fwd_blk + TW_BLK;
  const RowLanes L(lane);
  const uint32_t nt = g->nt, np = g->np;
  const u64 eightq = q << 3;
  u64 A[R][4][2];
#pragma unroll
  for (int i = 0; i < R; i++)
#pragma unroll
    for (int m = 0; m < 4; m++) A[i][m][0] = A[i][m][1] = 0;

  // The digits are walked in BLOCKS -- four transformed digits over the wave's R rows (R tiles), or one tile of DPT
  // left-over transformed / plain digits -- and every wave starts at a different block of the cycle: waves that share a
  // SIMD would otherwise all wait for their loads, then all compute, in step (the sum does not care about the order).
  // The digits left over by the groups of four and the plain ones are ONE list, cut into tiles of DPT digits x R rows: a line is a digit
  // of either kind (a plain digit riding in a tile with transformed lines is read again after the transform), so a group with
  // 4 nquad + tr transformed and np plain digits costs ceil(tr / DPT) tiles with a transform and the rest of ceil((tr + np) / DPT) without.
  const uint32_t nquad = nt >> 2, tr = nt - 4 * nquad;
  const uint32_t ntr = (tr + DPT - 1) / DPT, npl = (tr + np + DPT - 1) / DPT - ntr;
  const uint32_t nblocks = nquad + ntr + npl;
  const uint32_t rowA = r0 + (L.lwA & (R - 1)), rowB = r0 + (L.lwB & (R - 1));
  const uint32_t offP = (rowA << 8) + L.plA;
  ctw_t blkA = gtw + (size_t)rowA * TW_BLK, blkB = gtw + (size_t)rowB * TW_BLK;
  // this wave's share [b_lo, b_hi) of the blocks, by cost (a tile with a transform 2, a plain one 1): the block whose
  // midpoint falls into the wave's S-th of the total
  uint32_t b_lo = 0, b_hi = nblocks;
  if (S > 1) {
    const uint32_t total = 2 * (2 * R * nquad + 2 * ntr + npl);
    uint32_t pre = 0;
    b_lo = b_hi = nblocks;
    for (uint32_t bb = 0; bb < nblocks; bb++) {
      const uint32_t cost = 2 * (bb < nquad ? 2 * R : bb < nquad + ntr ? 2 : 1);
      const uint32_t owner = (pre + cost / 2) * S / total;
      if (owner == wv && b_lo == nblocks) b_lo = bb;
      if (owner > wv && b_hi == nblocks) b_hi = bb;
      pre += cost;
    }
    if (b_lo == nblocks) b_lo = b_hi = 0;                        // (no share: fewer blocks than waves)
  }
  const uint32_t nmine = b_hi - b_lo;
  uint32_t b = nmine ? b_lo + (blockIdx.x >> 3) % nmine : 0;    // (workgroups 8 apart share an XCD)
  for (uint32_t bi = 0; bi < nmine; bi++, b = b + 1 == b_hi ? b_lo : b + 1) {
    if (b < nquad) {
      // -- four digits, row by row: 4 lines = row r0 + i of digits 4 b + lw ------------------------------------------------
      const uint32_t dA = 4 * b + L.lwA;
      cgptr_t xs = (cgptr_t)g->x[dA], k0 = (cgptr_t)g->key[dA][0], k1 = (cgptr_t)g->key[dA][1];
      gptr_t xo = (gptr_t)g->xout[dA];
      for (uint32_t i = 0; i < R; i++) {
        const uint32_t row = r0 + i;                             // (wave-uniform)
        const uint32_t off = (row << 8) + L.plA;
        cgptr_t src = xs + off;
        ctw_t blk = gtw + (size_t)row * TW_BLK;
        u64 v[16];
#pragma unroll
        for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&src[16 * kk]);
        row_fwd_core<true, FWD_MID, true, PRE>(v, lds, L, blk, blk, pr);
        if (xo) {
#pragma unroll
          for (int kk = 0; kk < 16; kk++) st_stream(&(xo + off)[16 * kk], v[kk]);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
          u64 r[1][2];
          products_chunk<0>(v, m, k0 + off, k1 + off, true, nq, mu, r);
          A[0][m][0] = csub_u(A[0][m][0] + r[0][0], eightq);    // both < 8q
          A[0][m][1] = csub_u(A[0][m][1] + r[0][1], eightq);
        }
        // the next row accumulates into A[0] again: rotate the rows (register moves; identity after R)
        u64 t[4][2];
#pragma unroll
        for (int m = 0; m < 4; m++) { t[m][0] = A[0][m][0]; t[m][1] = A[0][m][1]; }
#pragma unroll
        for (int ii = 0; ii + 1 < R; ii++)
#pragma unroll
          for (int m = 0; m < 4; m++) { A[ii][m][0] = A[ii + 1][m][0]; A[ii][m][1] = A[ii + 1][m][1]; }
#pragma unroll
        for (int m = 0; m < 4; m++) { A[R - 1][m][0] = t[m][0]; A[R - 1][m][1] = t[m][1]; }
      }
    } else {
      // -- DPT digits over the wave's R rows: the transformed ones left over by the groups of four, then the plain ones ----
      const bool transformed = b < nquad + ntr;                 // the tile has transformed lines
      const uint32_t d0 = 4 * nquad + (b - nquad) * DPT;
      const uint32_t d = d0 + (L.lwA >> DSH);                   // the lane's digit (A mapping)
      const bool live = R == 4 ? true : d < nt + np;            // (R = 4: one digit per tile, every line has it)
      const bool plain_line = d >= nt;                           // (per line)
      const uint32_t dd = live ? d : d0;                         // (idle lines read the first digit's rows: defined data, zero weight)
      cgptr_t src = (cgptr_t)g->x[dd] + offP;
      u64 v[16];
      if (transformed) {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&src[16 * kk]);
        if constexpr (R == 1) row_fwd_core<true, FWD_MID, true, PRE>(v, lds, L, blkA, blkA, pr);   // (one row: one block for the wave)
        else row_fwd_core<false, FWD_MID, true, PRE, (R == 2 ? 7 : 3)>(v, lds, L, blkA, blkB, pr);
        gptr_t xo = (gptr_t)g->xout[dd];
        if (live && !plain_line && xo) {
#pragma unroll
          for (int kk = 0; kk < 16; kk++) st_stream(&(xo + offP)[16 * kk], v[kk]);
        }
        if (plain_line) {                                        // a plain digit riding in the tile: its words as they are in memory
#pragma unroll
          for (int kk = 0; kk < 16; kk++) v[kk] = src[16 * kk];
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) v[kk] = src[16 * kk];   // NTT form already; any word < 2^60 (modalu.sv:44-46)
      }
      cgptr_t key0 = (cgptr_t)g->key[dd][0] + offP, key1 = (cgptr_t)g->key[dd][1] + offP;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        u64 r[R][2];
        products_chunk<DSH>(v, m, key0, key1, live, nq, mu, r);
#pragma unroll
        for (int i = 0; i < R; i++)
#pragma unroll
          for (int c = 0; c < 2; c++) A[i][m][c] = csub_u(A[i][m][c] + r[i][c], eightq);   // < 8q + 4q
      }
    }
  }

  // -- the other waves' shares ------------------------------------------------------------------------------------------
  if (S > 1) {
    __syncthreads();                                             // every wave is done with its transpose region
    if (wv) {
#pragma unroll
      for (int i = 0; i < R; i++)
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int c = 0; c < 2; c++) lds[((i * 4 + m) * 2 + c) * 64 + lane] = A[i][m][c];
    }
    __syncthreads();
    if (wv) return;
#pragma unroll
    for (int w = 1; w < S; w++)
#pragma unroll
      for (int i = 0; i < R; i++)
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int c = 0; c < 2; c++) A[i][m][c] = csub_u(A[i][m][c] + lds_all[w * 4 * RP + ((i * 4 + m) * 2 + c) * 64 + lane], eightq);
  }

  // -- canonical results, 128-byte segments ----------------------------------------------------------------------------
  const uint32_t kbase = 2 * (L.lwA & 1) + (L.lwA >> 1);
#pragma unroll
  for (int c = 0; c < 2; c++) {
    gptr_t out = (gptr_t)g->out[c] + L.plA;
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int m = 0; m < 4; m++)
        st_stream(&out[((r0 + i) << 8) + 16 * (4 * m + kbase)], canon_any(A[i][m][c], q, nq, kc));
  }
}

}  // namespace

hipError_t ntt_fwd16_cols(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count, const SumSrc *d_sums,
                          uint64_t *launches) {
  if (!count) return hipSuccess;
  hipLaunchKernelGGL(ntt_col_pro_kernel, dim3(count * 16), dim3(256), 0, st, d_jobs, d_primes, count, d_sums);
  if (launches) ++*launches;
  return hipGetLastError();
}

hipError_t ntt_fwd16_rows(hipStream_t st, const LimbJob *d_inplace, const PrimeDev *d_primes, const TileDesc *d_uni_tiles, uint32_t n_uni,
                          const TileDesc *d_glb_tiles, uint32_t n_glb, const RowEpi *d_epis, uint64_t *launches) {
  if (n_uni) {
    hipLaunchKernelGGL((ntt_row_epi_kernel<true>), dim3(n_uni * 4), dim3(64), 0, st, d_inplace, d_primes, d_uni_tiles, d_epis);
    if (launches) ++*launches;
  }
  if (n_glb) {
    hipLaunchKernelGGL((ntt_row_epi_kernel<false>), dim3(n_glb * 4), dim3(64), 0, st, d_inplace, d_primes, d_glb_tiles, d_epis);
    if (launches) ++*launches;
  }
  return hipGetLastError();
}

// Rows per wave for a launch, from its groups (host side, when the launch is planned).  Batches of ciphertexts (>= 96 groups) fill the chip
// with four rows per wave and no idle lines.  Below that a wave's chain of tiles is what a launch costs, and the shapes differ in how many
// tiles carry a transform: with two rows per wave the left-over digits travel two to a tile, with one row per wave four to a tile.  Per two rows of a group with 4 nquad + tr transformed and np plain digits:
//   R = 2:  2 nquad + ceil(tr / 2) tiles with a transform, ceil((tr + np) / 2) - ceil(tr / 2) without
//   R = 1:  2 nquad + 2 [tr > 0] with, 2 (ceil((tr + np) / 4) - [tr > 0]) without          (a tile without ~ half a tile with).  HMult L = 24, alpha = 4 (tr = 1 / 2): R = 2, 0.239 against 0.243 ms; rotate L = 32 (tr = 3 / 0): R = 1,
// 0.315 against 0.344 ms (profiles/r04_w8_share_rowdot_rows.txt).  Very few groups (a rank's share at W = 8): R = 1 whatever the mix -- at two rows
// the launch is one wave per SIMD and its time one wave's chain.
uint32_t rowdot_rows(const RowDotGroup *h_groups, uint32_t ngroups) {
  if (ngroups >= 96) return 4;
  if (ngroups <= 10) return 1;
  uint32_t c1 = 0, c2 = 0;                                       // in half transforms
  for (uint32_t i = 0; i < ngroups; i++) {
    const uint32_t nt = h_groups[i].nt, np = h_groups[i].np, nquad = nt >> 2, tr = nt - 4 * nquad, has = tr ? 1u : 0u;
    c1 += 2 * (2 * nquad + 2 * has) + 2 * ((tr + np + 3) / 4 - has);
    c2 += 2 * (2 * nquad + (tr + 1) / 2) + ((tr + np + 1) / 2 - (tr + 1) / 2);
  }
  return c1 <= c2 ? 1 : 2;
}

hipError_t ntt_fwd16_rowdot(hipStream_t st, const RowDotGroup *d_groups, uint32_t ngroups, const PrimeDev *d_primes, uint32_t rows, uint64_t *launches) {
  if (!ngroups) return hipSuccess;
  // rows: rowdot_rows() of the launch's groups.  (ALOHA_ROWDOT_ROWS / ALOHA_ROWDOT_SPLIT: A/B timing switches)
  static const int forced = getenv("ALOHA_ROWDOT_ROWS") ? atoi(getenv("ALOHA_ROWDOT_ROWS")) : 0;
  static const int forced_split = getenv("ALOHA_ROWDOT_SPLIT") ? atoi(getenv("ALOHA_ROWDOT_SPLIT")) : 0;
  if (forced) rows = (uint32_t)forced;
  const bool split = forced_split ? forced_split == 2 : ngroups < 48;          // (S = 4, round 4: 0.292 against 0.271 ms; S = 1 equals S = 2 by now: profiles/r04_ks_kernel_experiments.txt)
  // one row per wave at W = 8 (3-5 groups): rank 0's share HMult L = 24 0.1027 -> 0.0978 ms, rotate L = 32 0.1164 -> 0.1056; four waves
  // per row instead of two: slower (profiles/r04_w8_share_rowdot_rows.txt)
  if (rows == 1) hipLaunchKernelGGL((ntt_rowdot_kernel<1, 2>), dim3(ngroups * 256), dim3(128), 0, st, d_groups, d_primes);
  else if (rows == 2 && split) hipLaunchKernelGGL((ntt_rowdot_kernel<2, 2>), dim3(ngroups * 128), dim3(128), 0, st, d_groups, d_primes);
  else if (rows == 2) hipLaunchKernelGGL((ntt_rowdot_kernel<2, 1>), dim3(ngroups * 128), dim3(64), 0, st, d_groups, d_primes);
  else hipLaunchKernelGGL((ntt_rowdot_kernel<4, 1>), dim3(ngroups * 64), dim3(64), 0, st, d_groups, d_primes);
  if (launches) ++*launches;
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
