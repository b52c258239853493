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
// (SURVEY 8d).  Timing experiments (the variant build of experiments/ntt_tile_knobs.hip, profiles/): memory
// skeleton ~195 us per pass per 1024 limbs, arithmetic ~150-180 us; in the row pass the transpose-back costs
// nothing, the final canonicalisation ~4 %, the per-lane second-step twiddles ~7 %.
// The tile bodies live in ntt_tiles.hpp (shared with the key-switch forms of ntt_ks.hip); the measured-and-rejected
// variants of these kernels -- the single-launch transforms through the Infinity Cache, five column workgroups per CU,
// wave priorities, ablations -- are in experiments/ntt_tile_knobs.hip, built only by `python -m aloha_amd.build --variant`.
#include <cstdlib>
#include <vector>

#include "kernels.hpp"
#include "modarith.hpp"
#include "ntt_device.hpp"
#include "ntt_tiles.hpp"

namespace aloha {
namespace k {

namespace {

// ---- column pass (N = 2^16): see col_tile_fwd / col_tile_inv in ntt_tiles.hpp -----------------------------------
template <bool INV, bool FIRST, bool LAST, bool SC = false>     // SC: jobs may carry their own last-stage scalar (LimbJob::sc)
__global__ __launch_bounds__(256) void ntt_col_kernel(const LimbJob *__restrict__ jobs, const PrimeDev *__restrict__ primes, uint32_t count) {
  __shared__ u64 sdata[LINES * PITCH];
  const int tid = threadIdx.x;
  TILE_STAMP(primes, 0, 7);
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
  if (!INV) col_tile_fwd<FIRST, LAST, false>(sdata, jb, pr, cg, tid, nullptr);
  else col_tile_inv<FIRST, LAST, SC>(sdata, jb, &jobs[limb], pr, cg, tid);
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

// ---- row pass: 16 lines of 256 contiguous points per tile, 4 lines per WAVE ------------------
// Lines of a tile are the same row of 2^pb limbs that share a prime (TileDesc), so with pb >= 2 a wave's four lines use
// ONE twiddle block: the first radix-16 step takes its 15 twiddles from SGPRs (scalar loads), the second reads its
// kh-dependent ones through the vector cache (16 distinct 16-byte words per load).  Mappings and the LDS image:
// ntt_tiles.hpp.  Launch shape: one WAVE per workgroup (64 threads, blockIdx = tile*4 + wave): the waves share
// nothing, so giving each its own workgroup lets the dispatcher refill a finished wave's slot at once and lifts the
// occupancy limit from 4 to 5 waves per SIMD (LDS: 8960 B per wave).  (A conflict-free B mapping was measured 0.8 %
// slower end to end, profiles/r01: the LDS is 24 % busy here, not a limiter.)
template <bool INV, bool UNIA, bool FIRST, bool LAST>
__device__ __forceinline__ void row_tile(u64 *lds, const LimbJob *__restrict__ jobs, const PrimeDev *__restrict__ primes,
                                         const TileDesc td, const uint32_t w, const int lane) {
  const uint32_t first_limb = td.first_limb, pb_log = td.pb_log, chunk = td.chunk;
  const uint32_t pb_mask = (1u << pb_log) - 1, rpt = 16u >> pb_log;
  const LimbJob j0 = jobs[first_limb];
  const PrimeDev *pr = &primes[j0.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t gtw = (ctw_t)(INV ? pr->tw.inv_blk : pr->tw.fwd_blk) + TW_BLK;   // row blocks
  const RowLanes L(lane);
  const uint32_t lineA = 4 * w + L.lwA, lineB = 4 * w + L.lwB;
  const uint32_t rowA = chunk * rpt + (lineA >> pb_log), rowB = chunk * rpt + (lineB >> pb_log);
  const uint32_t rowU = chunk * rpt + ((4 * w) >> pb_log);               // wave-uniform when pb_log >= 2
  ctw_t blkA = gtw + (size_t)(UNIA ? rowU : rowA) * TW_BLK;
  ctw_t blkB = gtw + (size_t)(UNIA ? rowU : rowB) * TW_BLK;
  const LimbJob jA = jobs[first_limb + (lineA & pb_mask)];
  cgptr_t src = (cgptr_t)jA.src + ((size_t)rowA << 8) + L.plA;
  gptr_t dst = (gptr_t)jA.dst + ((size_t)rowA << 8) + L.plA;

  u64 v[16];
  ROW_STAMP(pr, j0.prime, 0);
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = ld_stream(&src[16 * kk]);
  ROW_STAMP_VM(pr, j0.prime, 1);
  if (!INV) {
    row_fwd_core<UNIA, FIRST ? FWD_IN : FWD_MID, LAST>(v, lds, L, blkA, blkB, pr);
  } else {
    row_inv_core<UNIA, FIRST ? INV_IN : INV_MID>(v, lds, L, blkA, blkB, pr);
    if (LAST) {
      const u64 ni = pr->tw.n_inv, nip = pr->tw.n_inv_shoup;
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] = scale_canon(v[kk], ni, nip, q, nq);
    }
  }
  ROW_STAMP(pr, j0.prime, 4);
#pragma unroll
  for (int kk = 0; kk < 16; kk++) st_stream(&dst[16 * kk], v[kk]);
  ROW_STAMP(pr, j0.prime, 5);
  ROW_STAMP_VM(pr, j0.prime, 6);
}

// (the inverse instantiations settle at 100-134 VGPRs, 3-4 waves per SIMD; holding them to 96 costs 28 B of scratch and 2 %)
// The forward bundle tiles need 94-98 VGPRs by layout: 96 (5 waves per SIMD) as shipped.  Two registers more cost a fifth of
// the occupancy and 1 % of the headline; FORCING 96 (amdgpu_waves_per_eu(5)) where the allocation wants 98 costs 20 B of scratch and
// 3 % (profiles/r04_row_lds_layout_ab.txt).
template <bool INV, bool UNIA, bool FIRST, bool LAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1))) void ntt_row_kernel(const LimbJob *__restrict__ jobs,
                                                                                              const PrimeDev *__restrict__ primes,
                                                                                              const TileDesc *__restrict__ tiles) {
  __shared__ u64 sdata[4 * RP];
  ROW_STAMP(primes, 0, 7);
  row_tile<INV, UNIA, FIRST, LAST>(sdata, jobs, primes, tiles[blockIdx.x >> 2], blockIdx.x & 3, threadIdx.x);
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

hipError_t ntt_inv16_cols(hipStream_t st, const LimbJob *d_inplace, const PrimeDev *d_primes, uint32_t count, bool scaled, uint64_t *launches) {
  if (scaled) launch_cols<true, false, true, true>(st, d_inplace, d_primes, count, launches);
  else launch_cols<true, false, true>(st, d_inplace, d_primes, count, launches);
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
