// kernels.hpp -- host-callable launchers of the gfx950 kernels (see *.hip for the bodies).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace aloha {
namespace k {

typedef unsigned long long u64;

// per-modulus constants passed by value (SGPRs)
struct ModCtx {
  u64 q;    // prime, 2^59 < q < 2^60
  u64 nq;   // 2^64 - q
  u64 mu;   // floor(2^121 / q)
  u64 kc;   // floor(2^59 / (hi32(q) + 1)): quotient estimator of canon_any (modarith.hpp)
};

// element-wise ALU ops, one per decodable modalu opcode (src/vp/vxu/modalu.sv:22-37)
enum EwOp : int {
  EW_MUL_VV = 0, EW_MUL_VS, EW_ADD_VV, EW_ADD_VS, EW_SUB_VV, EW_SUB_VS, EW_SUB_SV, EW_MOD, EW_CP
};

// d[i] = op(a[i], b[i] | s); s must already be normalised (< q), s_shoup = floor(s*2^64/q)
hipError_t ew(hipStream_t st, EwOp op, u64 *d, const u64 *a, const u64 *b, u64 s, u64 s_shoup,
              ModCtx m, uint32_t n);

// automorphism gather: out[j] = +-in[j * k_inv mod N], sign from (i*k mod 2N) >= N
// (src/vp/vxu/vxu_lane.sv:594-598); k already reduced mod N, k_inv = k^-1 mod N
hipError_t vaut(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 q, u64 k, u64 k_inv);
// extension: the same automorphism on an NTT-form limb -- out[j] = in[brv(((2 brv(j)+1) k mod 2N - 1)/2)], k mod N
hipError_t vautn(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 k);
// extension: out[j] = q where in[j] == 0 and j is a sign-flipped target of the automorphism k, else in[j] (vaut's flipped zero)
hipError_t vautz(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 q, u64 k, u64 k_inv);
// cyclic roll: out[j] = in[(j + s) mod N]  (vxu_lane.sv:595,:597)
hipError_t vroli(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 s);

// ---- batched forms used by the fusing sequencer (one launch for many limbs) ----------------------
struct EwJob {             // d = op(a, b | s) on n words under modulus m
  u64 *d;
  const u64 *a, *b;
  u64 s, s_shoup;
  ModCtx m;
  int op, pad;
};
struct AutJob {            // roll == 0: automorphism gather with (k, k_inv); 1: cyclic roll by k; 2: vautn (NTT-form permutation by k); 3: vautz
  u64 *out;
  const u64 *in;
  u64 q;
  uint32_t k, k_inv, roll, pad;
};
hipError_t ew_batch(hipStream_t st, const EwJob *d_jobs, uint32_t njobs, uint32_t n);
// two_pass: every job is an automorphism (no roll) -> two coalesced LDS passes for N >= 2^12 (ew_kernels.hip)
hipError_t aut_batch(hipStream_t st, const AutJob *d_jobs, uint32_t njobs, unsigned logn, bool two_pass);

// The tensor product of two ciphertexts under one modulus (the first step of a homomorphic product; no reference
// program ships): d0 = a0 .* b0, d1 = a0 .* b1 + a1 .* b0, d2 = a1 .* b1 -- as instructions two
// vfqmul.vv with stores and a two-term sum: 11 limb passes; here one pass over the four inputs, 7.
struct TensorJob {
  const u64 *a0, *a1, *b0, *b1;
  u64 *d0, *d1, *d2;
  ModCtx m;
};
hipError_t tensor_batch(hipStream_t st, const TensorJob *d_jobs, uint32_t njobs, uint32_t n);

// d = init + sum_t a[t] .* (b[t] | s[t])  (mod q), up to kDotTerms terms per launch: the key-switch
// inner product (keyswitch.mem:46-79) as ONE pass over its operands.
constexpr int kDotTerms = 64;
struct DotArgs {
  const u64 *a[kDotTerms];   // nullptr: term t is the constant s[t]
  const u64 *b[kDotTerms];   // nullptr: term t is a[t] times the scalar s[t] (vfqmul.vs)
  u64 s[kDotTerms];          // normalised scalar (< q) ...
  u64 sp[kDotTerms];         // ... and floor(s * 2^64 / q): scalar terms are Shoup products (14 instructions, Barrett: ~35)
  const u64 *init;         // may be null
  u64 *d;
  ModCtx m;
  int nterms, pad;
};
hipError_t dot(hipStream_t st, const DotArgs &args, uint32_t n);
// many independent dot products in ONE launch (blockIdx.y = job): all accumulators of a key-switch
// phase together fill the chip, where a single one occupies 128 workgroups
hipError_t dot_batch(hipStream_t st, const DotArgs *d_jobs, uint32_t njobs, uint32_t n);

// Sums of SCALED operands that share their operands (a basis conversion: for one digit, the same alpha
// coefficient-form limbs times a different scalar row under every target modulus).  As separate dot jobs
// each output re-reads the operands; a group reads them once per element and loops over its outputs:
//   d_j[i] = sum_t a_t[i] * s_jt  (+ cst_j)   mod q_j      for every job j of the group.
constexpr int kSumTerms = 8;
struct SumGroup {
  const u64 *a[kSumTerms];
  int nterms;              // 1..kSumTerms
  uint32_t first, count;   // jobs [first, first + count) of the job array
  uint32_t pad;
};
struct SumJob {
  u64 s[kSumTerms], sp[kSumTerms];   // scalars (< q) and their Shoup companions
  u64 cst;                           // constant term (< q), 0 if none
  u64 *d;
  ModCtx m;
};
hipError_t sum_groups(hipStream_t st, const SumGroup *d_groups, uint32_t ngroups, const SumJob *d_jobs, uint32_t n);

// ---- transforms --------------------------------------------------------------------------
// Device-resident twiddle set of one (q, N); tables hold (w, floor(w*2^64/q)) pairs.
struct TwiddleDev {
  const u64 *fwd;        // [N][2]   psi^bitrev(k)          (tf_rom_generator.sv:28-30)
  const u64 *inv;        // [N][2]   psi^-bitrev(k)         (:61-63)
  const u64 *fwd_blk;    // blocked copy for the tile kernels, see ntt_tile.hip
  const u64 *inv_blk;
  u64 n_inv, n_inv_shoup;
  u64 w1n_inv, w1n_inv_shoup;   // psi^-bitrev(1) * N^-1: twiddle of the last inverse stage with the scaling folded in
};

// One limb job of a batched transform.  Device job arrays hold 2*count entries: the first
// `count` are {src, dst}, the second `count` the in-place form {dst, dst} used by every pass
// after the first.
struct LimbJob {
  const u64 *src;
  u64 *dst;
  uint32_t prime;        // index into the PrimeDev table
  uint32_t scaled;       // inverse transforms only: 1 = multiply the result by the scalar folded into sc[] and add addc
  // `vintt; vfqmul.vs c; [vfqadd.vs a]` as ONE transform: the last inverse stage already multiplies its sums by N^-1
  // and its differences by w*N^-1 (ntt_device.hpp); with sc = {w*N^-1*c, its Shoup companion, N^-1*c, companion}
  // the extra scalar costs nothing, and the constant is one modular add on the canonical result.
  u64 sc[4];
  u64 addc;
  // forward transforms of the fused key-switch phases (N = 2^16, ntt_tile.hip / ntt_ks.hip); 0 = none, else index + 1:
  uint32_t pro;          // SumSrc: the limb to transform is a sum of scaled memory operands, formed in the column pass's loads
  uint32_t epi;          // RowEpi: the row pass ends in a sum of scaled operands with the transform as one of them
  // inverse transforms at N = 2^16 followed by `vautz` (automorphism exponent zk, its inverse mod N; 0 = none): canonical zeros
  // at sign-flipped targets leave as q
  uint32_t zk, zkinv;
};

// A limb that exists only as  sum_t a_t * s_t + cst  (mod q): the source of a column pass (LimbJob::pro).
struct SumSrc {
  const u64 *a[kSumTerms];
  u64 s[kSumTerms], sp[kSumTerms];   // scalars (< q) and their Shoup companions
  u64 cst;                           // constant term (< q)
  int nterms, pad;                   // 1..kSumTerms
};
// Epilogue of a forward row pass (LimbJob::epi):  out = t * s[0] + sum_i a_i * s[i + 1] + cst  (mod q), t = the transform
// of the limb -- the tail of the mod-down, (acc - t) * P^-1 + base (keyswitch.mem:102-117), without a pass of its own.
constexpr int kEpiOps = 3;
struct RowEpi {
  const u64 *a[kEpiOps];
  u64 s[kEpiOps + 1], sp[kEpiOps + 1];
  u64 cst;
  u64 *out;
  uint32_t nops;         // operands besides t: 0..kEpiOps
  uint32_t store_t;      // 1: the transform itself is stored to the job's dst as well
};
// One inner product with the key (keyswitch.mem:46-79) fused behind the row pass of its digits:
//   out_c = sum_d X_d .* KEY_{d,c}   c = 0, 1,   X_d = NTT(x_d) for the first nt digits (x_d: column-pass output, in
//   place at its dst), X_d = x_d as it stands for the next np digits (already in NTT form).
constexpr int kRowDotMax = 64;
struct RowDotGroup {
  const u64 *x[kRowDotMax];
  u64 *xout[kRowDotMax];         // transformed digits: where NTT(x_d) is stored, or null (not needed in memory)
  const u64 *key[kRowDotMax][2];
  u64 *out[2];
  uint32_t prime, nt, np, pad;
};
struct PrimeDev {
  ModCtx m;
  TwiddleDev tw;
};
constexpr uint32_t kPrimeTableEntries = 4096;   // entries of the device table of PrimeDev (the engine's twiddle-set cache indexes it)
constexpr uint32_t kTileTraceWords = 1u << 17;  // variant builds with -DALOHA_TILE_TRACE: clock stamps behind the table (ntt_tiles.hpp)

struct TileDesc {        // one row-pass tile of ntt_tile.hip
  uint32_t first_limb;   // first limb of the bundle (same prime, consecutive jobs)
  uint16_t pb_log;       // log2(limbs in the bundle): 4..0
  uint16_t chunk;        // rows [chunk*rpt, chunk*rpt + rpt), rpt = 16 >> pb_log
};

// Row-pass tile lists for a batch (limbs that share a prime must be adjacent to be bundled):
// tiles whose twiddle blocks fit LDS (bundles of >= 4 limbs) and the rest.
void build_row_tiles(const LimbJob *h_jobs, uint32_t begin, uint32_t count, unsigned log_rows,
                     std::vector<TileDesc> &uni_tiles, std::vector<TileDesc> &glb_tiles);

// Radix-2 stage launches straight on global memory, any N: bring-up path, on-device
// cross-check of the tile kernels and column passes of N < 2^16.
//   forward: stages s in [begin, end), s = 0 is the stride-N/2 stage; canonical in/out
//   inverse: stages u in [begin, end), u = 0 is the stride-1 stage; inputs < 2q, canonical
//            out, no per-stage halving; last_scales multiplies the final stage by N^-1
hipError_t ntt_radix2_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                            unsigned logn, unsigned stage_begin, unsigned stage_end, bool inverse,
                            bool first_reads_src, bool last_scales, uint64_t *launches);

// Tile transform of limbs [begin, begin+count) of a batch: d_jobs = the {src,dst} array,
// d_inplace = the {dst,dst} array (same indexing); the tile lists cover exactly these limbs.
hipError_t ntt_tile_batch(hipStream_t st, const LimbJob *d_jobs, const LimbJob *d_inplace,
                          const PrimeDev *d_primes, uint32_t begin, uint32_t count, unsigned logn,
                          bool inverse, const TileDesc *d_uni_tiles, uint32_t n_uni,
                          const TileDesc *d_glb_tiles, uint32_t n_glb, uint64_t *launches,
                          unsigned single_max = 13,    // largest log2 N transformed by the single-pass kernels (0: never)
                          bool scaled = false);       // inverse only: some job carries its own last-stage scalar (LimbJob::scaled)

// ---- the fused key-switch phases at N = 2^16 (forward transforms; ntt_tile.hip, ntt_ks.hip) -------------------------
// column pass of limbs [0, count) of d_jobs; jobs with `pro` take their input from d_sums
hipError_t ntt_fwd16_cols(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count, const SumSrc *d_sums,
                          uint64_t *launches);
// row pass over the tile lists (in place on the jobs' dst); jobs with `epi` end in their epilogue (d_epis)
hipError_t ntt_fwd16_rows(hipStream_t st, const LimbJob *d_inplace, const PrimeDev *d_primes, const TileDesc *d_uni_tiles, uint32_t n_uni,
                          const TileDesc *d_glb_tiles, uint32_t n_glb, const RowEpi *d_epis, uint64_t *launches);
// inverse column pass at N = 2^16, in place on the jobs' dst (`scaled`: some job carries its own last-stage scalar)
hipError_t ntt_inv16_cols(hipStream_t st, const LimbJob *d_inplace, const PrimeDev *d_primes, uint32_t count, bool scaled, uint64_t *launches);
// row pass of every group's transformed digits + inner product with the key, one launch
uint32_t rowdot_rows(const RowDotGroup *h_groups, uint32_t ngroups);   // rows per wave (1 / 2 / 4) for a launch over these groups
hipError_t ntt_fwd16_rowdot(hipStream_t st, const RowDotGroup *d_groups, uint32_t ngroups, const PrimeDev *d_primes, uint32_t rows, uint64_t *launches);

#ifdef ALOHA_EXPERIMENTS
// Fused forward transform for N = 2^16 (experiments/ntt_tile_knobs.hip, ntt_fused_fwd_kernel): both passes in ONE persistent launch,
// the intermediate handed from column tiles to row tiles through the Infinity Cache.  Needs a batch whose row tiles
// are all bundle tiles (>= 4 limbs per prime adjacent).  d_ctl: fused_ctl_words(count) u32 of device memory (zeroed by
// the launcher): ticket counters and one arrival counter per limb.  d_status: one sticky u32 owned by the caller,
// non-zero after a launch gave up on a bounded wait (its output is then incomplete).
constexpr uint32_t kFusedShards = 32;      // ticket counters (one 128-byte line each)
constexpr uint32_t kFusedCtrStride = 16;   // u32 words between arrival counters (64 bytes)
constexpr uint32_t kFusedStatsWord = kFusedShards * 32;   // ctl[kFusedStatsWord + 0/1]: tiles that waited / their poll iterations
void build_fused_schedule(const std::vector<TileDesc> &uni_tiles, unsigned log_rows, unsigned delay,
                          std::vector<uint64_t> &tickets);
size_t fused_ctl_words(uint32_t count);
uint32_t fused_workgroups_per_cu();      // resident workgroups of the fused kernel per CU (occupancy query)
hipError_t ntt_fused_fwd(hipStream_t st, const LimbJob *d_jobs, const LimbJob *d_inplace, const PrimeDev *d_primes,
                         const TileDesc *d_tiles, const uint64_t *d_tickets, uint32_t npos, uint32_t count,
                         uint32_t *d_ctl, uint32_t *d_status, uint32_t workgroups, uint64_t *launches);
#endif

// Single-pass transforms for N = 2^12 .. 2^14 (the limb stays in one CU's LDS: one read + one write of
// the limb, ntt_single.hip); d_jobs = the {src, dst} entries of the limbs to transform.
hipError_t ntt_single_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count,
                            unsigned logn, bool inverse, uint64_t *launches, bool scaled = false);

// ---- polynomial product at N = 2^14 as one launch (ntt_product.hip) ----------------------------------------------------
// d = iNTT(NTT(a) .* NTT(b)) under the job's modulus: `vle; vntt; vle; vntt; vfqmul.vv; vintt; vse` (BASELINE config 3).
struct ProdJob {
  const u64 *a, *b;
  u64 *d;
  uint32_t prime, pad;
};
constexpr uint32_t kProdMaxLimbs = 64;           // limbs per launch
// d_work: poly_product_work_words() words of scratch: three intermediates per limb slot, every word tagged in its top two
// bits with the launch that wrote it -- `tag` must differ from the tag the slots [0, count) carry (0 after a memset);
// d_status: one sticky u32, non-zero after a launch gave up on a bounded wait
size_t poly_product_work_words();
hipError_t poly_product_batch(hipStream_t st, const ProdJob *d_jobs, const PrimeDev *d_primes, uint32_t count, u64 *d_work, uint32_t *d_status,
                              uint32_t tag, uint64_t *launches);

// ---- measurement helpers (ew_kernels.hip) ---------------------------------------------------------------
// limb i of `count` limbs of n words at base + i*n: the first n accepted candidates (top 60 bits, < moduli[i]) of the
// splitmix64 stream seeded with seed_base + first_index + i
hipError_t fill_splitmix(hipStream_t st, u64 *base, const u64 *d_moduli, uint32_t count, uint32_t n, u64 seed_base,
                         u64 first_index);
// one launch of the VALU clock calibration: 8 waves per SIMD, 32 * iters v_mad_u64_u32 per wave
hipError_t valu_clock(hipStream_t st, u64 *d_sink, uint32_t simds, uint32_t iters);

// ---- CKKS encoder block (src/encoder), FP64: see encode_kernels.hip --------------------------------
struct EncodeTables {      // per ring size N = 2^logn, n = N/2 slots; built by the host (vp_engine.cpp)
  const double *w;         // [n/2][2]  exp(-2*pi*i*k/n)
  const double *twist;     // [n][2]    exp(-i*pi*k/N)
  const uint32_t *src;     // [n]       t -> slot index j, bit 31 = take conj(z_j)   (y_t of the derivation)
};
// nplain plaintexts of N doubles each (device) -> nplain*nlimbs limbs of N residues at d_out, coefficient
// form; d_work: nplain*n complex doubles, needed only for logn > 14
hipError_t encode(hipStream_t st, const double *d_re_im, const EncodeTables &tb, void *d_work, u64 *d_out,
                  const u64 *d_moduli, uint32_t nlimbs, uint32_t nplain, unsigned logn, double scale,
                  uint64_t *launches);

// the inverse map: one coefficient-form limb (modulus q) per plaintext, `limb_stride_words` apart, -> nplain * N
// doubles (slots, (re, im) interleaved) at d_out; d_work: nplain * n complex doubles
hipError_t decode(hipStream_t st, const u64 *d_limbs, size_t limb_stride_words, const EncodeTables &tb, void *d_work,
                  double *d_out, u64 q, uint32_t nplain, unsigned logn, double scale, uint64_t *launches);

}  // namespace k
}  // namespace aloha
