/*
 * include/aloha_vp.h -- C ABI of libaloha_vp.so: the ALOHA vector processor (VP) lane
 * datapath on MI355X (gfx950).  Plain pointers and sizes only; no C++/torch types.
 *
 * This is the drop-in boundary for the reference's hot path.  In the reference the VP is
 * driven through module ports / test-bench tasks (there is no software driver):
 *
 *   reference interface (relative to AntChainOpenLabs/ALOHA)          replaced by
 *   ---------------------------------------------------------------    --------------------
 *   vp_top_full ports  src/vp/top/vp_top_full.sv:15-41                 aloha_vp_create/destroy
 *   inst_rom $readmemh image  src/vp/sequncer/inst_rom.v:32-35         aloha_vp_load_isram[_hex]
 *   CSR pins pc/src0/src1/rslt/ksk/step + start/done
 *     src/top/h2_top_no_axilite.sv:60-79,:145-157;
 *     TB task run_vp  sim/top/top_noaxilite_tb.sv:396-417             aloha_vp_run / aloha_vp_sync
 *   DMA <-> SPM / KSK memory  run_load_cipher/run_store_cipher/
 *     copy_spm_to_dram/load_ksk  top_noaxilite_tb.sv:372-394,:450-520  aloha_vp_write / aloha_vp_read
 *   instruction semantics  src/vp/sequncer/expander.v:65-700           (executed inside run)
 *   (two extensions, absent from every reference image: funct6 011101 `vautn`, the automorphism of vaut on an
 *    NTT-form limb, and funct6 100001 `vautz`, which stores q where vaut would have stored a sign-flipped zero as q
 *    (vxu_lane.sv:598) -- with it iNTT(vautn(x)) equals the reference's vaut(iNTT(x)) word for word)
 *
 * Memory model: two row-addressed spaces, rows of 128 x u64 = 1024 bytes in
 * plain little-endian linear order (src/mem_buf/spm.sv:118-126, ksk_mem.sv:46-54).  A
 * polynomial limb of N coefficients occupies N/128 consecutive rows, coefficient i at byte
 * 8*i.  All residues are canonical or at least < 2^60 (< 2q): the RTL's ALU normalises its
 * inputs with ONE conditional subtraction (src/vp/vxu/modalu.sv:44-46).
 *
 * Errors: the RTL signals nothing and leaves bad encodings undefined (unknown funct6 is a
 * silent no-op bundle, expander.v:524-532).  This library instead returns a status and
 * REJECTS what the RTL leaves undefined.  Threading: a handle is single-threaded, like the
 * one-instruction-in-flight sequencer (seq_top.v:343-413); internally work is enqueued on a
 * HIP stream, aloha_vp_run may return before completion, aloha_vp_sync / aloha_vp_read block.
 *
 * One kernel of the library waits INSIDE its grid: the polynomial product at N = 2^14 as one launch (a loop of
 * `vle; vntt; vle; vntt; vfqmul.vv; vintt; vse` whose destination is none of its operands).  Its later phases wait for the earlier
 * ones, which makes progress while every XCD dispatches a queue's workgroups in grid order -- observed on MI355X, not promised by
 * HIP.  Every wait is bounded (1 s); a launch that gave up is reported as ALOHA_ERR_DEVICE by the next aloha_vp_sync / aloha_vp_read,
 * its operands are intact, and the handle runs its later products as five ordinary launches.  At most one handle per device has
 * such a launch in flight; a second handle's products take the five-launch form meanwhile (aloha_vp_stats.product_fallbacks).
 */
#ifndef ALOHA_VP_H
#define ALOHA_VP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALOHA_VP_ABI_VERSION 4

typedef enum {
  ALOHA_OK = 0,
  ALOHA_ERR_ARG = 1,      /* null / out-of-range argument                                  */
  ALOHA_ERR_VL = 2,       /* VL != 64 * 2^k, k in [10,16]   (ntt_fsm.sv:59-65)              */
  ALOHA_ERR_MODULUS = 3,  /* q not a prime in (2^59, 2^60)  (mod width hard-wired to 60,
                             vxu_lane.sv:539)                                               */
  ALOHA_ERR_ROOT = 4,     /* q != 1 mod 2N at vntt/vintt: no 2N-th root of unity            */
  ALOHA_ERR_DECODE = 5,   /* unknown funct6 / funct3        (expander.v:524-532)            */
  ALOHA_ERR_BANK = 6,     /* .vv operands in the same register bank (expander.v:183-200)    */
  ALOHA_ERR_ADDR = 7,     /* memory row range outside the configured space                  */
  ALOHA_ERR_AUT = 8,      /* even automorphism exponent (not a permutation)                 */
  ALOHA_ERR_NOBREAK = 9,  /* ran off the instruction ROM without `break`                    */
  ALOHA_ERR_DEVICE = 10,  /* HIP runtime error / no gfx950 device (never a CPU fallback)    */
  ALOHA_ERR_NOMEM = 11
} aloha_status;

enum { ALOHA_SPACE_SPM = 0, ALOHA_SPACE_KSK = 1 };
#define ALOHA_ROW_BYTES 1024u   /* 128 lanes x 8 B: vp_defines.vh:25,:62 */
#define ALOHA_IRAM_DEPTH 4096u  /* vp_defines.vh:31: the reference's instruction ROM, and the default here */
#define ALOHA_IRAM_DEPTH_MAX 65536u /* aloha_vp_config.iram_depth: a deeper ROM (same word format, same entry points) for programs the
                                       reference's 4096 words cannot hold in one piece -- a key-switch phase at L = 32 */
#define ALOHA_NVREG 32u

typedef struct aloha_vp aloha_vp;

typedef struct {
  uint32_t abi_version;   /* ALOHA_VP_ABI_VERSION                                            */
  int32_t device;         /* HIP device ordinal                                              */
  uint64_t spm_rows;      /* rows in the scratch-pad space (reference: 2^16, spm.sv:12-21)   */
  uint64_t ksk_rows;      /* rows in the key-switch-key space (ksk_mem.sv:12-21)             */
  uint32_t max_log_n;     /* largest ring size the register file is sized for (10..16)       */
  uint32_t flags;         /* ALOHA_FLAG_*                                                    */
  void *stream;           /* hipStream_t to enqueue on, or NULL for a private stream         */
  uint32_t iram_depth;    /* instruction ROM words: 0 = ALOHA_IRAM_DEPTH (4096, inst_rom.v:32-35), up to ALOHA_IRAM_DEPTH_MAX (ABI 3) */
  uint32_t reserved;
} aloha_vp_config;

#define ALOHA_FLAG_NO_FUSION 1u    /* one kernel per instruction (debug / parity bisecting)  */
#define ALOHA_FLAG_RADIX2_ONLY 2u /* transforms by radix-2 stage launches only (cross-check) */
#define ALOHA_FLAG_NTT_TWO_PASS 4u  /* N = 2^12, 2^13 through the two-pass tile kernels too (default: single pass); N = 2^16
                                       forward batches never as one fused launch (ALOHA_NTT_FUSED, off by default) */
#define ALOHA_FLAG_NTT_SINGLE_14 8u /* N = 2^14 through the single-pass kernel as well (default: two passes)        */

/* ---- lifecycle ----------------------------------------------------------------------- */
aloha_status aloha_vp_create(const aloha_vp_config *cfg, aloha_vp **out);
void aloha_vp_destroy(aloha_vp *vp);
const char *aloha_vp_last_error(const aloha_vp *vp);      /* text of the last failure      */
const char *aloha_status_string(aloha_status s);
uint32_t aloha_vp_abi_version(void);
/* the configuration the handle was created with (ABI 3): callers that lay data out in the handle's memories -- the
 * key-switch host below -- check their needs against it instead of finding out by an address error */
aloha_status aloha_vp_get_config(const aloha_vp *vp, aloha_vp_config *out);

/* ---- memory (replaces the AXI DMA paths) ------------------------------------------------ */
aloha_status aloha_vp_write(aloha_vp *vp, int space, uint64_t row, const void *host_src, uint64_t bytes);
aloha_status aloha_vp_read(aloha_vp *vp, int space, uint64_t row, void *host_dst, uint64_t bytes);
/* device address of a row, for callers that fill/consume HBM directly (zero copy) */
void *aloha_vp_device_ptr(aloha_vp *vp, int space, uint64_t row);
/* the HIP stream (hipStream_t) all work of the handle is enqueued on, and its device ordinal */
void *aloha_vp_stream(aloha_vp *vp);
int aloha_vp_device(const aloha_vp *vp);

/* ---- instruction ROM ---------------------------------------------------------------------- */
/* n words of {inst32, imm[63:32], imm[31:0]} placed at ROM address base_pc */
aloha_status aloha_vp_load_isram(aloha_vp *vp, uint32_t base_pc, const uint32_t *words, uint32_t n);
/* the reference's $readmemh text, one 24-hex-digit word per line, loaded unchanged */
aloha_status aloha_vp_load_isram_hex(aloha_vp *vp, uint32_t base_pc, const char *text);

/* ---- run: the six CSRs + start (top_noaxilite_tb.sv:396-417) -------------------------------- */
aloha_status aloha_vp_run(aloha_vp *vp, uint32_t pc, uint64_t src0_row, uint64_t src1_row,
                          uint64_t rslt_row, uint64_t ksk_row, uint64_t rot_step);
aloha_status aloha_vp_sync(aloha_vp *vp);   /* `done` */
/* Scratch window (ABI 3; no counterpart in the reference, whose memories are all observable): the caller declares SPM rows
 * [row, row + nrows) as its own scratch.  Rows of the window that a run WRITES have undefined contents once that run has
 * returned; everything else -- every byte a `vse` writes outside the window, in program order -- is unchanged.  It lets the
 * sequencer keep a value that one instruction stores and a later instruction of the SAME run loads in registers (the digit
 * limbs between their transform and the inner product with the key, keyswitch.mem:46-79).  nrows = 0 clears the window
 * (the default: none).  The window is part of what a recorded run is keyed on. */
aloha_status aloha_vp_set_scratch(aloha_vp *vp, uint64_t row, uint64_t nrows);

/* ---- batched transforms on rows (the fused form of N x {vsetq; vle; vntt|vintt; vse}) ---------
 * count limbs of 2^log_n coefficients; limb i is read at src_row + i*(N/128) and written at
 * dst_row + i*(N/128) (src_row == dst_row allowed; ranges that overlap otherwise are rejected
 * with ALOHA_ERR_ARG: the limbs are transformed concurrently), under modulus moduli[i].  Identical
 * results to running the encode_post-style program (encode_post.mem:1-12) limb by limb. */
aloha_status aloha_vp_ntt_rows(aloha_vp *vp, int space, uint64_t dst_row, uint64_t src_row,
                               uint32_t log_n, const uint64_t *moduli, uint32_t count, int inverse);

/* ---- CKKS encoder block (replaces src/encoder; encoder_top.sv:12-52, DMA command of the
 * test-bench task run_encode, top_noaxilite_tb.sv:419-448) -------------------------------------------
 * nplain cleartexts of N = 2^log_n doubles each -- N/2 complex slots, (re, im) interleaved, exactly the
 * DRAM image the test bench hands the encoder -- become nplain * nlimbs limbs in COEFFICIENT form at
 * dst_row (plaintext p, limb l at dst_row + (p*nlimbs + l)*(N/128), under moduli[l]); the caller then
 * runs encode_post / aloha_vp_ntt_rows, as run_encode does.  Convention of the reference's software
 * model (tv/ encode pairs): m(zeta^(3^j)) = scale * z_j, scale = 2^38 in tv/, round to nearest,
 * negative coefficients stored as q - |x|.  Double precision; agreement with the golden pairs is by
 * tolerance (their cleartexts carry 6 digits), and the RTL's own single-precision results differ from
 * its model everywhere.  |scale * z| must stay below 2^62. */
aloha_status aloha_vp_encode(aloha_vp *vp, int space, uint64_t dst_row, const double *host_re_im,
                             uint32_t nplain, uint32_t log_n, const uint64_t *moduli, uint32_t nlimbs,
                             double scale);

/* The inverse map (no counterpart in the reference, which has an encoder only): nplain plaintexts, each given
 * by ONE coefficient-form limb under `modulus` (plaintext p at src_row + p*limb_stride*(N/128); valid while every
 * coefficient's centred value is below modulus/2), to nplain * N doubles (slots, (re, im) interleaved). */
aloha_status aloha_vp_decode(aloha_vp *vp, int space, uint64_t src_row, uint64_t limb_stride, double *host_re_im,
                             uint32_t nplain, uint32_t log_n, uint64_t modulus, double scale);

/* ---- measurement helpers (no counterpart in the reference) -------------------------------------
 * Synthetic inputs generated in HBM: limb i of `count` limbs of N = 2^log_n words at row + i*(N/128) receives the
 * first N accepted candidates of the splitmix64 stream seeded with seed_base + first_index + i, where a candidate is
 * the top 60 bits of an output and is accepted when < moduli[i] (uniform canonical residues; 2^56 <= modulus <= 2^60). */
aloha_status aloha_vp_fill_synthetic(aloha_vp *vp, int space, uint64_t row, uint32_t log_n, const uint64_t *moduli,
                                     uint32_t count, uint64_t seed_base, uint64_t first_index);
/* Effective VALU clock in GHz under an integer multiply-add load, measured on the handle's device for about `ms`
 * milliseconds: prices a kernel's VALU instruction count (the second bound next to HBM bandwidth). */
aloha_status aloha_vp_valu_clock(aloha_vp *vp, double ms, double *ghz);

/* ---- the L-limb key-switch as a C host (keyswitch_host.cpp) ------------------------------------------------------
 * The reference's only cross-limb step is its rotate / key-switch micro-program (keyswitch.mem:1-122, driven by the
 * test-bench task run_rotate, top_noaxilite_tb.sv:522-532), written for 2 ciphertext primes + 1 special prime.  These
 * entry points generate that program for L primes, K special primes and alpha primes per digit (alpha = K = 1: the
 * reference's algorithm), as HE-ISA phase programs run on an aloha_vp handle, with limb j living on rank j mod world
 * and the two exchange steps (the digits, keyswitch.mem:4-43; the mod-down terms, :80-83) as ncclAllGather on the
 * handle's stream.  RCCL is loaded on first use; world = 1 needs none.
 *   ciphertext: 2 components x L limbs of N words, NTT form, c0 limbs then c1 limbs       (tv/README.md:4-9)
 *   key: (L+K) x dnum x 2 limbs ordered [modulus][digit][component], dnum = ceil(L/alpha) (keyswitch.mem:46-77)
 *   outputs: the OWNED limbs only, [component][owned index][N]; aloha_ks_owned lists the limb indices. */
typedef struct aloha_ks aloha_ks;
typedef struct {
  uint32_t log_n;            /* N = 2^log_n                                                           */
  uint32_t L, K, alpha;      /* ciphertext primes, special primes, primes per digit                   */
  uint32_t world, rank;      /* ranks sharing the limbs, this rank                                    */
  uint32_t batch;            /* ciphertexts switched per call under the one key (slots 0 .. batch-1)  */
  uint32_t flags;            /* ALOHA_KS_*                                                            */
  const uint64_t *moduli;    /* L primes                                                              */
  const uint64_t *specials;  /* K primes                                                              */
  uint64_t base_row;         /* first SPM row the handle may use                                      */
  uint32_t rom_depth;        /* instruction ROM words the phase programs may fill: 0 = ALOHA_IRAM_DEPTH; with a handle, at most
                                that handle's iram_depth.  A phase that does not fit is cut into several programs (by modulus).   */
  uint32_t reserved;
} aloha_ks_config;
/* alpha = 1 only: aloha_ks_rotate runs the reference's own phase 1 -- iNTT, vaut in the coefficient domain, NTT
 * (keyswitch.mem:4-7,:24-27,:100-121) -- instead of the NTT-domain automorphism.  Same words, except that a zero coefficient
 * of iNTT(c1) at a sign-flipped position is lifted as q_d (what the RTL's vaut stores, vxu_lane.sv:598), not as 0. */
#define ALOHA_KS_REFERENCE_PHASE1 1u
typedef struct {
  uint32_t dnum, owned_ct, owned_all, rows_per_limb, resident, pad;
  uint64_t spm_rows, ksk_rows;           /* what the aloha_vp must provide (ksk_rows: ONE key region)  */
  uint64_t bytes_sent, bytes_received;   /* exchange payload so far                                   */
  uint64_t key_rows;                     /* KSK rows of one key region (hoisted rotations keep one key per region) */
} aloha_ks_info_t;
aloha_status aloha_ks_rows_needed(const aloha_ks_config *cfg, uint64_t *spm_rows, uint64_t *ksk_rows);
/* vp may be NULL: a plan only -- programs are generated and can be inspected, nothing runs */
aloha_status aloha_ks_create(aloha_vp *vp, const aloha_ks_config *cfg, aloha_ks **out);
void aloha_ks_destroy(aloha_ks *ks);
const char *aloha_ks_last_error(const aloha_ks *ks);
aloha_status aloha_ks_info(const aloha_ks *ks, aloha_ks_info_t *out);
aloha_status aloha_ks_owned(const aloha_ks *ks, uint32_t *ct_indices /* owned_ct */, uint32_t *all_indices /* owned_all */);
/* instruction words of program `index` of a phase ("rot1", "mul1", "mid", "terms", "fin_rot", "fin_mul", "hoist1", "modup",
 * "inner_h", "fin_h", "rescale1", "rescale2"; after aloha_ks_hoist(.., 1): "hoist1_1", "modup_1"; after
 * aloha_ks_rotate_hoisted_many: "many", "many_terms", "many_fin" -- both also generate on a plan); *ninst: in =
 * capacity of `words` in instructions (3 words each), out = instructions of the program (0: no such program) */
aloha_status aloha_ks_program(const aloha_ks *ks, const char *phase, uint32_t index, uint32_t *words, uint32_t *ninst);
/* exchange: a 128-byte ncclUniqueId made on one rank and given to every rank's aloha_ks_comm_init (ncclCommInitRank over
 * cfg.world ranks on the handle's device), or an existing ncclComm_t borrowed with aloha_ks_set_comm */
aloha_status aloha_ks_nccl_unique_id(void *id128);
aloha_status aloha_ks_comm_init(aloha_ks *ks, const void *id128);
aloha_status aloha_ks_set_comm(aloha_ks *ks, void *nccl_comm);
/* exchanges as world-1 direct ncclSend/ncclRecv pairs per rank in one group (one per xGMI link) instead of ncclAllGather;
 * same words either way */
aloha_status aloha_ks_set_pairwise(aloha_ks *ks, int on);
aloha_status aloha_ks_load_key(aloha_ks *ks, const uint64_t *key_full);
aloha_status aloha_ks_load_ct(aloha_ks *ks, int which /* 0: the ciphertext, 1: the second factor of multiply */, uint32_t slot,
                              const uint64_t *ct_full);
aloha_status aloha_ks_read_out(aloha_ks *ks, uint32_t slot, uint64_t *out);
/* rotate by X -> X^k and switch the key / multiply the two loaded ciphertexts and relinearise; asynchronous on the
 * handle's stream like aloha_vp_run */
aloha_status aloha_ks_rotate(aloha_ks *ks, uint64_t k);
aloha_status aloha_ks_multiply(aloha_ks *ks);
/* For hosts that carry the exchanges themselves (MPI, sockets, gloo in a test): the same key-switch stage by stage -- op 0
 * rotate (by k), 1 multiply; stage 0, then all-gather buffer 0, stage 1, all-gather buffer 1, stage 2.  A buffer is `world`
 * chunks of chunk_words words from SPM row row0; rank r owns chunk r. */
aloha_status aloha_ks_stage(aloha_ks *ks, int op, int stage, uint64_t k);
aloha_status aloha_ks_gather_buffer(const aloha_ks *ks, int which, uint64_t *row0, uint64_t *chunk_words);
/* Hoisted rotations (BASELINE config 5 is a BATCH of rotations): decompose the ciphertext once -- aloha_ks_hoist; with
 * first_slot_only the other batch slots become ROTATION slots -- then rotate by many steps, each with its own key stored
 * by aloha_ks_load_key_hoisted(key, k, region) in KSK region `region` (regions are key_rows apart; the aloha_vp needs
 * regions x ksk_rows): one step at a time (aloha_ks_rotate_hoisted) or `count` <= batch steps in ONE pass of the phases,
 * OUT slot b = rotation by steps[b] with the key of regions[b] (NULL: region b). */
aloha_status aloha_ks_load_key_hoisted(aloha_ks *ks, const uint64_t *key_full, uint64_t k, uint32_t region);
aloha_status aloha_ks_hoist(aloha_ks *ks, int first_slot_only);
aloha_status aloha_ks_rotate_hoisted(aloha_ks *ks, uint64_t k, uint32_t region);
aloha_status aloha_ks_rotate_hoisted_many(aloha_ks *ks, const uint64_t *steps, const uint32_t *regions, uint32_t count);
/* rounded division by the last ciphertext prime (the step after a product in leveled CKKS): OUT limbs 0 .. L-2 */
aloha_status aloha_ks_rescale(aloha_ks *ks);

/* ---- parameters ------------------------------------------------------------------------------ */
/* minimal primitive 2N-th root of unity mod q -- the rule behind the reference's hard-coded
 * psi values (tf_rom_generator.sv:75) */
aloha_status aloha_find_psi(uint64_t q, uint32_t log_n, uint64_t *psi);
/* idx-th largest prime p < 2^60 with p == 1 (mod 2^(log_n+1)) */
uint64_t aloha_nth_prime(uint32_t log_n, uint32_t idx);

/* ---- introspection ------------------------------------------------------------------------- */
typedef struct {
  uint64_t instructions;    /* instructions retired since create                             */
  uint64_t kernel_launches; /* HIP kernel launches issued                                    */
  uint64_t twiddle_builds;  /* (q, N) twiddle tables generated                               */
  uint64_t fused_launches;  /* variant builds: N = 2^16 forward transforms run as ONE launch (ABI 2)   */
  uint64_t graph_launches;  /* aloha_vp_run calls served by one hipGraphLaunch of a recorded run (ABI 2) */
  uint64_t twiddle_generations; /* times the bounded twiddle cache was emptied to make room (ABI 2)     */
  uint64_t ks_fused_launches;   /* of kernel_launches: key-switch kernels that absorbed neighbouring sums (ABI 3)  */
  uint64_t product_fallbacks;   /* polynomial products (N = 2^14) run as five launches because another handle's one-launch product was
                                   in flight on the device, or this handle's status word was once set (ABI 4)       */
  uint64_t flow_runs;           /* aloha_vp_run calls sequenced by the program's dependency graph (rings up to N = 2^13; ABI 4)  */
} aloha_vp_stats;
aloha_status aloha_vp_get_stats(const aloha_vp *vp, aloha_vp_stats *out);
/* copy a vector register to the host (debug; not architecturally visible in the reference) */
aloha_status aloha_vp_peek_vreg(aloha_vp *vp, uint32_t reg, uint64_t *host_dst, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ALOHA_VP_H */
