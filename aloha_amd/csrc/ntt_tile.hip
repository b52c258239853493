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
// One workgroup (256 threads, 4 waves) owns a tile of 16 "lines" x 256 points = 32 KiB:
//   column tile (n1 = 8 only): 16 adjacent columns of one limb (128-byte segments per row)
//   row tile: 16 lines = (16 >> pb) rows x 2^pb limbs that share a prime (pb = 4..0), so one
//             4 KiB twiddle block feeds the same row of 2^pb limbs.
// A 256-point line transform is two radix-16 steps (4 butterfly stages each, 16 values per
// thread in registers) with a transpose through LDS in between, and one more so that global
// stores leave in the same 128-byte-segment pattern as the loads.
// Column passes for n1 < 8 (N < 2^16) run as radix-2 stage launches (ntt_radix2_stages).
//
// Twiddles: for both passes stage s' (0..7) of a 256-point line uses entry
//   L = 2^s' + (p >> (8 - s'))      p = point index inside the line
// of a 256-entry block of (w, floor(w*2^64/q)) pairs: block 0 is W[0..255] (column pass),
// block 1+r is the row-r slice W[2^(n1+s') + r*2^s' + i] re-indexed the same way (built by
// the host, vp_engine.cpp).  Tiles with <= 4 twiddle blocks stage them in LDS (the first
// radix-16 step then reads them at wave-uniform addresses = LDS broadcast); tiles with more
// (bundles of 1-2 limbs) read them through the vector cache.
//
// Arithmetic: lazy, see modarith.hpp.  Forward: relaxed CT butterflies, values < 16q, one
// borrow-subtract per element between radix-16 steps.  Inverse: Harvey GS butterflies, values
// < 2q, N^-1 fused into the last pass.  HBM traffic per pass: read + write of the limb
// (+ twiddle blocks); algorithmic bytes per transform: 16*N.
#include <vector>

#include "kernels.hpp"
#include "modarith.hpp"

namespace aloha {
namespace k {

namespace {

constexpr int LINES = 16;          // lines per tile
constexpr int PTS = 256;           // points per line
constexpr int PITCH = PTS + 1;     // LDS pitch in u64: odd => conflict-free 64-bit accesses
constexpr int TW_BLK = 256;        // twiddle entries per block

struct alignas(16) tw_t { u64 w, wp; };

// ---- butterflies ----------------------------------------------------------------------------
// Harvey GS: X, Y < 2q  ->  X' = X + Y (- 2q) < 2q,  Y' = (X - Y + 2q) * w  < 2q.
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, u64 w, u64 wp, u64 q, u64 twoq) {
  const u64 s = csub_borrow(X + Y, twoq);
  const u64 d = X + twoq - Y;
  Y = mul_shoup_2q(d, w, wp, q);
  X = s;
}

// radix-16 steps on 16 registers: sub-stage ss pairs v[i] with v[i + (8 >> ss)]; group g of
// sub-stage ss takes twiddle tw(ss, g).
template <typename TW>
__device__ __forceinline__ void fwd_radix16(u64 (&v)[16], const TW &tw, u64 nq, u64 twoq) {
#pragma unroll
  for (int ss = 0; ss < 4; ss++) {
    const int half = 8 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int i0 = g * 2 * half + jj;
        ct_bfly<false, true>(v[i0], v[i0 + half], t.w, t.wp, nq, twoq);
      }
    }
  }
}

template <typename TW>
__device__ __forceinline__ void inv_radix16(u64 (&v)[16], const TW &tw, u64 q, u64 twoq) {
#pragma unroll
  for (int ss = 3; ss >= 0; ss--) {
    const int half = 8 >> ss;
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const tw_t t = tw(ss, g);
#pragma unroll
      for (int jj = 0; jj < half; jj++) {
        const int i0 = g * 2 * half + jj;
        gs_bfly(v[i0], v[i0 + half], t.w, t.wp, q, twoq);
      }
    }
  }
}

// twiddle accessors over one 256-entry block (LDS or global)
struct TwStepA {                   // stages s' = 0..3: L = 2^ss + g   (same for every thread)
  const tw_t *blk;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return blk[(1 << ss) + g]; }
};
struct TwStepB {                   // stages s' = 4..7: L = 2^(4+ss) + kh*2^ss + g
  const tw_t *blk;
  int kh;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    return blk[(16 << ss) + (kh << ss) + g];
  }
};

// full canonicalisation of a value < 16q
__device__ __forceinline__ u64 canon16(u64 x, u64 q) {
  x = csub_borrow(x, q << 3);
  x = csub_borrow(x, q << 2);
  x = csub_borrow(x, q << 1);
  return csub_borrow(x, q);
}

}  // namespace

namespace {

// ---- the tile kernel ------------------------------------------------------------------------
// ROWPASS false: column tile (blockIdx -> limb, column group); true: row tile (TileDesc).
// INV selects GS.  TWLDS: twiddle blocks staged in LDS (needs rpt <= 4).
// FIRST: inputs come from memory (< 2^60 < 2q) and get the RTL's one conditional subtract
//        (modalu.sv:44-46); otherwise they are the previous pass's intermediate format.
// LAST : outputs are canonical; otherwise intermediate format (forward < 8q, inverse < 2q).
template <bool ROWPASS, bool INV, bool TWLDS, bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void ntt_tile_kernel(const LimbJob *__restrict__ jobs,
                                                       const PrimeDev *__restrict__ primes,
                                                       const TileDesc *__restrict__ tiles) {
  __shared__ u64 sdata[LINES * PITCH];
  __shared__ tw_t stw[TWLDS ? 4 * TW_BLK : 1];

  const int tid = threadIdx.x;
  const int fast = tid & 15, slow = tid >> 4;

  // ---- tile geometry (wave-uniform) ----
  uint32_t first_limb, pb_log, chunk;
  if (ROWPASS) {
    const TileDesc td = tiles[blockIdx.x];
    first_limb = td.first_limb; pb_log = td.pb_log; chunk = td.chunk;
  } else {
    first_limb = blockIdx.x >> 4; pb_log = 4; chunk = blockIdx.x & 15;    // 16 groups of 16 columns
  }
  const uint32_t pb_mask = (1u << pb_log) - 1;
  const uint32_t rpt = ROWPASS ? (16u >> pb_log) : 1u;                    // twiddle blocks in the tile
  const LimbJob j0 = jobs[first_limb];
  const PrimeDev *pr = &primes[j0.prime];
  const u64 q = pr->m.q, nq = pr->m.nq, twoq = q << 1;
  const tw_t *gtw = reinterpret_cast<const tw_t *>(INV ? pr->tw.inv_blk : pr->tw.fwd_blk) +
                    (ROWPASS ? (size_t)(1 + chunk * rpt) * TW_BLK : 0);

  if (TWLDS) {
    for (uint32_t b = 0; b < rpt; b++) stw[b * TW_BLK + tid] = gtw[b * TW_BLK + tid];
  }
  const tw_t *twbase = TWLDS ? stw : gtw;

  // ---- per-thread addressing ----
  // step-A register mapping: thread (line, pl) holds points p = pl + 16k; the lane-fast index
  // is the one that is contiguous in memory (points for rows, lines for columns).
  const int lineA = ROWPASS ? slow : fast;
  const int plA = ROWPASS ? fast : slow;
  const u64 *srcA; u64 *dstA; uint32_t rselA;
  if (ROWPASS) {
    const LimbJob jb = jobs[first_limb + ((uint32_t)lineA & pb_mask)];
    rselA = (uint32_t)lineA >> pb_log;
    const size_t off = (size_t)(chunk * rpt + rselA) << 8;                // row * 256
    srcA = jb.src + off; dstA = jb.dst + off;
  } else {
    rselA = 0;
    srcA = j0.src + chunk * 16 + lineA; dstA = j0.dst + chunk * 16 + lineA;
  }
  constexpr int PSH = ROWPASS ? 0 : 8;                                    // point stride = 1 or 256 words
  // step-B register mapping: thread (line = fast, kh = slow) holds points p = 16*kh + j
  const int lineB = fast, khB = slow;
  const uint32_t rselB = ROWPASS ? ((uint32_t)lineB >> pb_log) : 0;

  u64 v[16];
#pragma unroll
  for (int kk = 0; kk < 16; kk++) {
    u64 x = srcA[(size_t)(plA + 16 * kk) << PSH];
    if (FIRST) x = csub(x, q);
    v[kk] = x;
  }
  if (TWLDS) __syncthreads();                                             // twiddle blocks visible

  if (!INV) {
    // stages 0..3 in the A mapping, transpose, stages 4..7 in the B mapping
    fwd_radix16(v, TwStepA{twbase + rselA * TW_BLK}, nq, twoq);           // < in + 8q  (< 16q)
#pragma unroll
    for (int kk = 0; kk < 16; kk++) sdata[lineA * PITCH + plA + 16 * kk] = v[kk];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = csub_borrow(sdata[lineB * PITCH + 16 * khB + jj], q << 3);   // < 8q
    fwd_radix16(v, TwStepB{twbase + rselB * TW_BLK, khB}, nq, twoq);      // < 16q
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = LAST ? canon16(v[jj], q) : csub_borrow(v[jj], q << 3);
  } else {
    // stages 7..4 need the B mapping first
#pragma unroll
    for (int kk = 0; kk < 16; kk++) sdata[lineA * PITCH + plA + 16 * kk] = v[kk];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; jj++) v[jj] = sdata[lineB * PITCH + 16 * khB + jj];
    inv_radix16(v, TwStepB{twbase + rselB * TW_BLK, khB}, q, twoq);       // < 2q throughout
  }

  // ---- back to the A mapping (second inverse step / coalesced stores) ----
  // each thread rewrites exactly the LDS words it read in the B mapping: no barrier needed
  // before the writes, one before the A-mapping reads.
#pragma unroll
  for (int jj = 0; jj < 16; jj++) sdata[lineB * PITCH + 16 * khB + jj] = v[jj];
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < 16; kk++) v[kk] = sdata[lineA * PITCH + plA + 16 * kk];

  if (INV) {
    inv_radix16(v, TwStepA{twbase + rselA * TW_BLK}, q, twoq);
    if (LAST) {
      // fused N^-1: the reference halves at every stage (halfred.sv:23-26); the product of
      // the halvings is N^-1 and the canonical result is identical
      const u64 ni = pr->tw.n_inv, nip = pr->tw.n_inv_shoup;
#pragma unroll
      for (int kk = 0; kk < 16; kk++) v[kk] = csub(mul_shoup_2q(v[kk], ni, nip, q), q);
    }
  }
#pragma unroll
  for (int kk = 0; kk < 16; kk++) dstA[(size_t)(plA + 16 * kk) << PSH] = v[kk];
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

// Inverse stage with t = 2^u (h = N/2t groups): inputs < 2q, canonical out, no halving; the
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
  u64 U = csub(src[j], q), V = csub(src[j + t], q);
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
void build_row_tiles(const LimbJob *h_jobs, uint32_t count, unsigned log_rows,
                     std::vector<TileDesc> &lds_tiles, std::vector<TileDesc> &glb_tiles) {
  const uint32_t rows = 1u << log_rows;
  uint32_t i = 0;
  while (i < count) {
    uint32_t run = 1;
    while (i + run < count && h_jobs[i + run].prime == h_jobs[i].prime) run++;
    uint32_t done = 0;
    while (done < run) {
      uint32_t pb_log = 4;
      while ((1u << pb_log) > run - done) pb_log--;
      uint32_t rpt = 16u >> pb_log;
      if (rpt > rows) {                     // tiny transforms: fewer rows than a tile holds
        // not expressible as a full tile; signalled to the caller by an empty list
        lds_tiles.clear(); glb_tiles.clear();
        return;
      }
      std::vector<TileDesc> &dst = (rpt <= 4) ? lds_tiles : glb_tiles;
      for (uint32_t c = 0; c < rows / rpt; c++) dst.push_back(TileDesc{i + done, (uint16_t)pb_log, (uint16_t)c});
      done += 1u << pb_log;
    }
    i += run;
  }
}

template <bool ROWPASS, bool INV, bool FIRST, bool LAST>
static void launch_tiles(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes,
                         const TileDesc *d_tiles, uint32_t ntiles, bool twlds, uint64_t *launches) {
  if (!ntiles) return;
  if (twlds)
    hipLaunchKernelGGL((ntt_tile_kernel<ROWPASS, INV, true, FIRST, LAST>), dim3(ntiles), dim3(256), 0, st,
                       d_jobs, d_primes, d_tiles);
  else
    hipLaunchKernelGGL((ntt_tile_kernel<ROWPASS, INV, false, FIRST, LAST>), dim3(ntiles), dim3(256), 0, st,
                       d_jobs, d_primes, d_tiles);
  if (launches) ++*launches;
}

hipError_t ntt_radix2_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                            unsigned logn, unsigned stage_begin, unsigned stage_end, bool inverse,
                            bool first_reads_src, bool last_scales, uint64_t *launches) {
  // forward: stages s = stage_begin .. stage_end-1 (s = 0 is the stride-N/2 stage)
  // inverse: stages u = stage_begin .. stage_end-1 (u = 0 is the stride-1 stage)
  const dim3 grid((1u << logn) / 2 / 256, count), blk(256);
  for (unsigned s = stage_begin; s < stage_end; s++) {
    const int first = (first_reads_src && s == stage_begin) ? 1 : 0;
    if (!inverse)
      hipLaunchKernelGGL(ntt_radix2_fwd_stage, grid, blk, 0, st, d_jobs, d_primes, logn, s, first);
    else
      hipLaunchKernelGGL(ntt_radix2_inv_stage, grid, blk, 0, st, d_jobs, d_primes, logn, s, first,
                         (last_scales && s + 1 == stage_end) ? 1 : 0);
    if (launches) ++*launches;
  }
  return hipGetLastError();
}

hipError_t ntt_tile_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                          unsigned logn, bool inverse, const TileDesc *d_lds_tiles, uint32_t n_lds,
                          const TileDesc *d_glb_tiles, uint32_t n_glb, uint64_t *launches) {
  if (logn < 10 || logn > 16) return hipErrorInvalidValue;
  const unsigned n1 = logn - 8;
  const bool col_tiles = (n1 == 8);
  if (!inverse) {
    if (col_tiles) {
      launch_tiles<false, false, true, false>(st, d_jobs, d_primes, nullptr, count * 16, true, launches);
      launch_tiles<true, false, false, true>(st, d_jobs, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_tiles<true, false, false, true>(st, d_jobs, d_primes, d_glb_tiles, n_glb, false, launches);
    } else {
      hipError_t e = ntt_radix2_batch(st, d_jobs, d_primes, count, logn, 0, n1, false, true, false, launches);
      if (e != hipSuccess) return e;
      // radix-2 stages wrote canonical data to dst; the row tiles then run in place on dst
      launch_tiles<true, false, true, true>(st, d_jobs + count, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_tiles<true, false, true, true>(st, d_jobs + count, d_primes, d_glb_tiles, n_glb, false, launches);
    }
  } else {
    if (col_tiles) {
      launch_tiles<true, true, true, false>(st, d_jobs, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_tiles<true, true, true, false>(st, d_jobs, d_primes, d_glb_tiles, n_glb, false, launches);
      launch_tiles<false, true, false, true>(st, d_jobs + count, d_primes, nullptr, count * 16, true, launches);
    } else {
      launch_tiles<true, true, true, false>(st, d_jobs, d_primes, d_lds_tiles, n_lds, true, launches);
      launch_tiles<true, true, true, false>(st, d_jobs, d_primes, d_glb_tiles, n_glb, false, launches);
      hipError_t e = ntt_radix2_batch(st, d_jobs + count, d_primes, count, logn, 8, logn, true, true, true, launches);
      if (e != hipSuccess) return e;
    }
  }
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
