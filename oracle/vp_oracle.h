/*
 * oracle/vp_oracle.h -- CPU restatement of the ALOHA vector-processor (VP) lane datapath.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under aloha_amd/ (the product) may include, link or
 * execute this code.  It is imported only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, always as the checker, never as the thing shipped.
 *
 * The reference (/root/reference) is synthesizable SystemVerilog; no Verilog simulator
 * exists in the build image, so the reference itself cannot be executed (oracle/_ref is
 * therefore empty).  This file restates, in plain C, the arithmetic the RTL
 * performs, function by function, each citing the reference file:line it follows.  It is
 * pinned by tests/test_oracle_golden.py against the reference's own tv/ golden vectors.
 */
#ifndef VP_ORACLE_H
#define VP_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (same numeric values as include/aloha_vp.h so tests can compare) */
enum {
  ORC_OK = 0,
  ORC_ERR_ARG = 1,        /* bad argument (null, range) */
  ORC_ERR_VL = 2,         /* VL not 64*2^k, k in [10,16] */
  ORC_ERR_MODULUS = 3,    /* q not in (2^59, 2^60), or not prime */
  ORC_ERR_ROOT = 4,       /* q != 1 mod 2N at an NTT */
  ORC_ERR_DECODE = 5,     /* unknown funct6 / funct3 */
  ORC_ERR_BANK = 6,       /* .vv operands in the same register bank (RTL: undefined) */
  ORC_ERR_ADDR = 7,       /* memory row out of range */
  ORC_ERR_AUT = 8,        /* even automorphism exponent */
  ORC_ERR_NOBREAK = 9     /* ran off the instruction ROM without `break` */
};

#define ORC_LANES 128u         /* src/vp/include/vp_defines.vh:25 */
#define ORC_ROW_BYTES 1024u    /* 128 lanes x 8 B, src/mem_buf/spm.sv:12-21 */
#define ORC_NVREG 32u
#define ORC_NMAX 65536u
#define ORC_IRAM_DEPTH 4096u   /* vp_defines.vh:31: the reference's ROM, the default of a machine */
#define ORC_IRAM_DEPTH_MAX 65536u /* orc_vp_set_iram_depth: a deeper ROM for the test programs the engine runs with one */

/* ---- scalar arithmetic ------------------------------------------------------------- */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);           /* exact a*b mod q */
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);                        /* q prime */
int      orc_is_prime(uint64_t n);
/* bit-faithful restatement of the RTL Barrett multiplier, src/vp/vxu/modmul.sv:150-251 */
uint64_t orc_barrett_rtl(uint64_t a, uint64_t b, uint64_t q, uint64_t mu);
uint64_t orc_barrett_mu(uint64_t q);                                /* floor(2^121/q) */
uint64_t orc_half(uint64_t x, uint64_t q);                          /* halfred.sv:23-26 */
/* modalu opcode model (modalu.sv:22-37,44-46,196-249; TB model modalu_tb.sv:133-165) */
int orc_alu(unsigned opcode, uint64_t a, uint64_t b, uint64_t s, uint64_t q,
            uint64_t *res0, uint64_t *res1);

/* ---- parameters -------------------------------------------------------------------- */
/* minimal primitive 2N-th root of unity mod q (tf_rom_generator.sv:75) */
int orc_find_psi(uint64_t q, unsigned logn, uint64_t *psi);
/* the i-th (0-based) largest prime p < 2^60 with p == 1 mod 2^(logn+1) */
uint64_t orc_nth_prime(unsigned logn, unsigned idx);
/* tab[k] = psi^bitrev(k,logn), itab[k] = psi^-bitrev(k,logn)  (tf_rom_generator.sv:28-63) */
void orc_make_tables(uint64_t q, uint64_t psi, unsigned logn, uint64_t *tab, uint64_t *itab);

/* ---- vector kernels (one limb, n = 2^logn u64 words) -------------------------------- */
void orc_ntt(uint64_t *a, unsigned logn, uint64_t q, const uint64_t *tab);   /* vntt  */
void orc_intt(uint64_t *a, unsigned logn, uint64_t q, const uint64_t *itab); /* vintt */
void orc_vaut(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t q, uint64_t k);
int orc_set_threads(int n);   /* OpenMP threads of orc_ntt_batch; 0 = all cores; returns the count in effect */
void orc_vautn(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t k);   /* extension: NTT-domain automorphism */
void orc_vautz(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t q, uint64_t k);   /* extension: vaut's flipped zeros (q) put back */
void orc_vroli(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t s);
/* convenience: table build + transform of `count` limbs stored back to back */
int orc_ntt_batch(uint64_t *a, unsigned logn, uint64_t q, unsigned count, int inverse);

/* ---- the VP as an ISA machine -------------------------------------------------------- */
typedef struct orc_vp orc_vp;
orc_vp *orc_vp_create(uint64_t spm_rows, uint64_t ksk_rows);
int orc_vp_set_iram_depth(orc_vp *vp, uint32_t depth);       /* ORC_IRAM_DEPTH .. ORC_IRAM_DEPTH_MAX */
void    orc_vp_destroy(orc_vp *vp);
/* space 0 = SPM, 1 = KSK memory; row = 1 KiB row index */
int orc_vp_write(orc_vp *vp, int space, uint64_t row, const void *src, uint64_t bytes);
int orc_vp_read(orc_vp *vp, int space, uint64_t row, void *dst, uint64_t bytes);
/* words: n entries of {inst32, imm_hi32, imm_lo32} written at ROM address base_pc */
int orc_vp_load_isram(orc_vp *vp, uint32_t base_pc, const uint32_t *words, uint32_t n);
/* run from pc until `break` with the six CSRs of top_noaxilite_tb.sv:396-417 */
int orc_vp_run(orc_vp *vp, uint32_t pc, uint64_t src0_row, uint64_t src1_row,
               uint64_t rslt_row, uint64_t ksk_row, uint64_t step);
/* debug access to a vector register (not architecturally visible in the reference) */
int orc_vp_peek_vreg(orc_vp *vp, unsigned r, uint64_t *dst, uint64_t n);
uint64_t orc_vp_instret(orc_vp *vp);

#ifdef __cplusplus
}
#endif
#endif
