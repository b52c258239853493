/*
 * oracle/vp_oracle.c -- CPU restatement of the ALOHA VP lane datapath (see vp_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  The product (aloha_amd/) never links or calls this file.
 *
 * Reference citations are relative to /root/reference.  Domain of definition: every
 * residue in memory / registers is < 2^60 < 2q (the RTL's modalu normalises its inputs
 * with ONE conditional subtraction, src/vp/vxu/modalu.sv:44-46, and the reference's own
 * ALU test bench constrains operands to < 2^60, sim/vp/vxu/modalu_tb.sv:82-92).
 */
#include "vp_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------ */
/* scalar arithmetic                                                                     */
/* ------------------------------------------------------------------------------------ */

/* modalu_tb.sv:46-55 (AluFunc::modmul): (opa*opb) % mod in 128 bit */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

/* tf_rom_generator.sv:32-46 (powerMod) */
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = orc_mulmod(r, a, q);
    a = orc_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a, q - 2, q); }

/* deterministic Miller-Rabin for 64-bit integers (bases 2..37) */
int orc_is_prime(uint64_t n) {
  static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return 0;
  for (int i = 0; i < 12; i++) {
    if (n == bases[i]) return 1;
    if (n % bases[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  unsigned s = 0;
  while (!(d & 1)) { d >>= 1; s++; }
  for (int i = 0; i < 12; i++) {
    uint64_t x = orc_powmod(bases[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (unsigned r = 1; r < s; r++) {
      x = orc_mulmod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}

/* vsetiq operand used by the shipped programs: floor(2^(2w+1)/q), w = 60
 * (modalu_tb.sv:113-119 cal_imod; keyswitch.mem:3,8) */
uint64_t orc_barrett_mu(uint64_t q) { return (uint64_t)((((u128)1) << 121) / q); }

/* src/vp/vxu/modmul.sv:150-251.  mod_width is hard-wired to 60 (vxu_lane.sv:539).
 *   prod       = a*b                                   (mul0, :60-98)
 *   prod_shift = (prod >> (w-2)) truncated to 64 bit   (:150)
 *   mid        = prod_shift * mu                       (mul1, :153-165)
 *   mid_shift  = (mid >> (w+3)) truncated to 64 bit    (:172)
 *   estim      = mid_shift * q                         (mul2, :175-187)
 *   mask       = 1 << (w+1)                            (:195)
 *   diff       = ((prod & (mask-1) | mask) - (estim & (mask-1))) & (mask-1)   (:216-222)
 *   res        = diff < q ? diff : diff - q            (:241-251)                       */
uint64_t orc_barrett_rtl(uint64_t a, uint64_t b, uint64_t q, uint64_t mu) {
  const unsigned w = 60;
  u128 prod = (u128)a * b;
  uint64_t prod_shift = (uint64_t)(prod >> (w - 2));
  u128 mid = (u128)prod_shift * mu;
  uint64_t mid_shift = (uint64_t)(mid >> (w + 3));
  u128 estim = (u128)mid_shift * q;
  uint64_t mask = 1ull << (w + 1);
  uint64_t dx = (uint64_t)prod & (mask - 1);
  uint64_t dy = (uint64_t)estim & (mask - 1);
  uint64_t diff = ((dx | mask) - dy) & (mask - 1);
  return diff < q ? diff : diff - q;
}

/* src/vp/vxu/halfred.sv:23-26: x/2 mod q = (x>>1) + (x&1 ? (q+1)/2 : 0) */
uint64_t orc_half(uint64_t x, uint64_t q) { return (x >> 1) + ((x & 1) ? ((q + 1) >> 1) : 0); }

/* modalu.sv:44-46 -- every ALU input passes ONE conditional subtraction */
static inline uint64_t norm1(uint64_t x, uint64_t q) { return x >= q ? x - q : x; }
/* modalu.sv:223-224 (add_sum is 65 bit wide, result truncated to 64) */
static inline uint64_t addred(uint64_t a, uint64_t b, uint64_t q) {
  u128 s = (u128)a + b;
  return (uint64_t)(s >= q ? s - q : s);
}
/* modalu.sv:247 */
static inline uint64_t subred(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : q + a - b; }

/* modalu.sv:196-205 + output mux :351-377; TB model modalu_tb.sv:133-165 */
int orc_alu(unsigned opcode, uint64_t a_i, uint64_t b_i, uint64_t s_i, uint64_t q,
            uint64_t *res0, uint64_t *res1) {
  uint64_t a = norm1(a_i, q), b = norm1(b_i, q), s = norm1(s_i, q);
  uint64_t r0 = 0, r1 = 0;
  switch (opcode) {
    case 0x00: r0 = orc_mulmod(a, b, q); break;                 /* modmul.vv */
    case 0x04: r0 = orc_mulmod(a, s, q); break;                 /* modmul.vs */
    case 0x01: r0 = addred(a, b, q); break;                     /* modadd.vv */
    case 0x05: r0 = addred(a, s, q); break;                     /* modadd.vs */
    case 0x02: r0 = subred(a, b, q); break;                     /* modsub.vv */
    case 0x06: r0 = subred(a, s, q); break;                     /* modsub.vs */
    case 0x0a: r0 = subred(s, a, q); break;                     /* modsub.sv */
    case 0x03: r0 = orc_mulmod(a, 1, q); break;                 /* mod (a*1, modalu.sv:164) */
    case 0x15: r0 = addred(orc_mulmod(a, b, q), s, q); break;   /* madd.vs */
    case 0x16: r0 = subred(orc_mulmod(a, b, q), s, q); break;   /* msub.vs */
    case 0x1a: r0 = subred(s, orc_mulmod(a, b, q), q); break;   /* msub.sv */
    case 0x10: {                                                /* ct.vvs  */
      uint64_t v = orc_mulmod(b, s, q);
      r0 = addred(a, v, q);
      r1 = subred(a, v, q);
      break;
    }
    case 0x13: {                                                /* gs.vvs  */
      r0 = orc_half(addred(a, b, q), q);
      r1 = orc_half(orc_mulmod(subred(a, b, q), s, q), q);
      break;
    }
    case 0x11: r0 = orc_mulmod(subred(a, b, q), s, q); break;   /* vvs */
    default: return ORC_ERR_DECODE;
  }
  if (res0) *res0 = r0;
  if (res1) *res1 = r1;
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------ */
/* parameters                                                                            */
/* ------------------------------------------------------------------------------------ */

/* tf_rom_generator.sv:49-59 (BitReverse) */
static inline uint32_t bitrev(uint32_t x, unsigned len) {
  uint32_t r = 0;
  for (unsigned i = 0; i < len; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* The reference hard-codes psi per prime (tf_rom_generator.sv:75).  All three shipped
 * values are the numerically smallest primitive 2N-th root of unity mod q,
 * which is the rule restated here for arbitrary q. */
int orc_find_psi(uint64_t q, unsigned logn, uint64_t *psi) {
  if (logn < 1 || logn > 16) return ORC_ERR_VL;
  uint64_t two_n = 2ull << logn, n = 1ull << logn;
  if ((q - 1) % two_n) return ORC_ERR_ROOT;
  uint64_t g = 0;
  for (uint64_t x = 2; x < 1000; x++) {
    uint64_t r = orc_powmod(x, (q - 1) / two_n, q);
    if (orc_powmod(r, n, q) == q - 1) { g = r; break; }
  }
  if (!g) return ORC_ERR_MODULUS;
  /* all primitive 2N-th roots are the odd powers of g */
  uint64_t g2 = orc_mulmod(g, g, q), cur = g, best = g;
  for (uint64_t i = 1; i < n; i++) {
    cur = orc_mulmod(cur, g2, q);
    if (cur < best) best = cur;
  }
  *psi = best;
  return ORC_OK;
}

/* synthetic moduli for sizes the reference ships no prime for */
uint64_t orc_nth_prime(unsigned logn, unsigned idx) {
  uint64_t two_n = 2ull << logn;
  uint64_t p = ((1ull << 60) - 1) / two_n * two_n + 1;
  while (p >= (1ull << 60)) p -= two_n;
  for (;;) {
    if (orc_is_prime(p)) {
      if (idx == 0) return p;
      idx--;
    }
    p -= two_n;
  }
}

/* tf_rom_generator.sv:28-30,:111 and :61-63,:148 */
void orc_make_tables(uint64_t q, uint64_t psi, unsigned logn, uint64_t *tab, uint64_t *itab) {
  uint32_t n = 1u << logn;
  uint64_t ipsi = orc_invmod(psi, q);
  /* powers in natural order, then scatter to bit-reversed slots */
  uint64_t p = 1, ip = 1;
  for (uint32_t e = 0; e < n; e++) {
    uint32_t k = bitrev(e, logn);
    if (tab) tab[k] = p;
    if (itab) itab[k] = ip;
    p = orc_mulmod(p, psi, q);
    ip = orc_mulmod(ip, ipsi, q);
  }
}

/* ------------------------------------------------------------------------------------ */
/* vector kernels                                                                        */
/* ------------------------------------------------------------------------------------ */

/* vntt.  The RTL runs log2 N stages of the CT butterfly (modalu.sv:160-165,:223-249,
 * opcode 10000) in a constant-geometry order (ntt_fsm.sv:71-81, iconn_shuffle.sv:33) with
 * twiddle psi^bitrev(m+i) (tf_rom_generator.sv:28-30,:105-111).  Its result is the
 * standard bit-reversed-twiddle CT network: natural-order in, bit-reversed out, every value
 * canonical after each stage. */
void orc_ntt(uint64_t *a, unsigned logn, uint64_t q, const uint64_t *tab) {
  uint32_t n = 1u << logn, t = n;
  for (uint32_t m = 1; m < n; m <<= 1) {
    t >>= 1;
    for (uint32_t i = 0; i < m; i++) {
      uint64_t w = tab[m + i];
      uint32_t j1 = 2 * i * t;
      for (uint32_t j = j1; j < j1 + t; j++) {
        uint64_t r0, r1;
        orc_alu(0x10, a[j], a[j + t], w, q, &r0, &r1);
        a[j] = r0;
        a[j + t] = r1;
      }
    }
  }
}

/* vintt.  GS butterfly with per-stage halving (modalu.sv:152,:264-328, halfred.sv:23-26,
 * opcode 10011), inverse table tf_rom_generator.sv:61-63,:147-148.  Bit-reversed in,
 * natural out, net scaling N^-1 with no separate multiply. */
void orc_intt(uint64_t *a, unsigned logn, uint64_t q, const uint64_t *itab) {
  uint32_t n = 1u << logn, t = 1;
  for (uint32_t m = n; m > 1; m >>= 1) {
    uint32_t h = m >> 1;
    for (uint32_t i = 0; i < h; i++) {
      uint64_t w = itab[h + i];
      uint32_t j1 = 2 * i * t;
      for (uint32_t j = j1; j < j1 + t; j++) {
        uint64_t r0, r1;
        orc_alu(0x13, a[j], a[j + t], w, q, &r0, &r1);
        a[j] = r0;
        a[j + t] = r1;
      }
    }
    t <<= 1;
  }
}

/* vaut.  vxu_lane.sv:594-598: element i goes to (i*k) mod N, negated as (q - x) when
 * (i*k) mod 2N >= N.  The exponent is first reduced mod N: the RTL keeps only
 * log2(VLMAX/64) = 13 bits of the scalar (vxu_lane.sv:594) and the block test bench's
 * model does `scalar % (vl/64)` (sim/vp/vxu/vxu_top_tb.sv:603).  The negation is the raw
 * 64-bit `q - x` with no normalisation, so a flipped 0 is stored as q (vxu_lane.sv:598). */
void orc_vaut(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t q, uint64_t k) {
  uint64_t n = 1ull << logn;
  uint64_t ke = k & (n - 1);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t t = (i * ke) & (2 * n - 1);
    out[t & (n - 1)] = (t >= n) ? q - in[i] : in[i];
  }
}

/* vautn -- EXTENSION (funct6 011101, not in the reference): the automorphism of vaut applied to a
 * limb in NTT form, i.e. to the bit-reversed evaluations that vntt produces.  vntt puts
 * x(psi^(2*brv(j)+1)) at index j (tf_rom_generator.sv:28-30 twiddle order), and
 * sigma_k(x)(psi^e) = x(psi^(e*k)), so the result is a pure permutation, no signs:
 *     out[j] = in[brv(((2*brv(j)+1) * k mod 2N - 1) / 2)]
 * with the exponent reduced mod N exactly as vaut reduces it, so that
 * vntt(vaut(x, k)) == vautn(vntt(x), k) for every operand k (tests). */
static uint64_t brv(uint64_t x, unsigned bits) {
  uint64_t r = 0;
  for (unsigned b = 0; b < bits; b++) r |= ((x >> b) & 1) << (bits - 1 - b);
  return r;
}
void orc_vautn(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t k) {
  uint64_t n = 1ull << logn;
  uint64_t ke = k & (n - 1);
  for (uint64_t j = 0; j < n; j++) {
    uint64_t e = ((2 * brv(j, logn) + 1) * ke) & (2 * n - 1);
    out[j] = in[brv((e - 1) >> 1, logn)];
  }
}

/* vautz -- EXTENSION (funct6 100001, not in the reference): the one place where the coefficient-domain vaut differs from
 * the same automorphism carried out in NTT form.  vaut stores a sign-flipped ZERO as q (vxu_lane.sv:598: the raw q - x);
 * y = vintt(vautn(vntt-form x, k)) holds 0 there.  vautz puts the q back:
 *     out[t] = q  if in[t] == 0 and t is a sign-flipped target of the automorphism k (t = i*k mod N with i*k mod 2N >= N),
 *     out[t] = in[t] otherwise (no normalisation),
 * so that vautz(vintt(vautn(X, k)), k) == vaut(vintt(X), k) word for word for canonical X (tests): the rotate program of
 * keyswitch.mem:4-7,:24-27 without its coefficient-domain permutation. */
void orc_vautz(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t q, uint64_t k) {
  uint64_t n = 1ull << logn;
  uint64_t ke = k & (n - 1);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t t = (i * ke) & (2 * n - 1), pos = t & (n - 1);
    out[pos] = (t >= n && in[pos] == 0) ? q : in[pos];
  }
}

/* vroli.  vxu_lane.sv:595,:597; model vxu_top_tb.sv:619-623: out[(i + N - s) mod N] = in[i] */
void orc_vroli(uint64_t *out, const uint64_t *in, unsigned logn, uint64_t s) {
  uint64_t n = 1ull << logn;
  uint64_t se = s & (n - 1);
  for (uint64_t i = 0; i < n; i++) out[(i + n - se) & (n - 1)] = in[i];
}

#ifdef _OPENMP
#include <omp.h>
#endif
/* threads used by orc_ntt_batch (0 = every core the process may run on); returns the count in effect */
int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n <= 0) n = omp_get_num_procs();
  omp_set_num_threads(n);
  return n;
#else
  (void)n;
  return 1;
#endif
}

int orc_ntt_batch(uint64_t *a, unsigned logn, uint64_t q, unsigned count, int inverse) {
  uint64_t psi;
  int rc = orc_find_psi(q, logn, &psi);
  if (rc) return rc;
  uint32_t n = 1u << logn;
  uint64_t *tab = (uint64_t *)malloc(sizeof(uint64_t) * n);
  if (!tab) return ORC_ERR_ARG;
  if (inverse) orc_make_tables(q, psi, logn, NULL, tab);
  else orc_make_tables(q, psi, logn, tab, NULL);
  /* limbs are independent: the CPU baseline of bench.py runs this loop on all host cores (OpenMP,
   * OMP_NUM_THREADS); one thread gives the scalar-port figure */
#pragma omp parallel for schedule(static) if (count > 1)
  for (unsigned c = 0; c < count; c++) {
    if (inverse) orc_intt(a + (size_t)c * n, logn, q, tab);
    else orc_ntt(a + (size_t)c * n, logn, q, tab);
  }
  free(tab);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------ */
/* the VP as an ISA machine                                                              */
/* ------------------------------------------------------------------------------------ */

struct orc_vp {
  uint64_t spm_rows, ksk_rows;
  uint64_t *spm, *ksk;               /* rows of 128 u64, plain little-endian linear order
                                        (spm.sv:118-126, ksk_mem.sv:46-54) */
  uint32_t inst[ORC_IRAM_DEPTH_MAX]; /* inst_rom.v:32-35 (the first iram_depth words are the ROM) */
  uint64_t imm[ORC_IRAM_DEPTH_MAX];
  uint32_t iram_depth;
  uint64_t vl_bits, q, mu;           /* vxu_top.sv:106-118 */
  uint64_t *vreg[ORC_NVREG];
  /* twiddle cache for the current (q, logn) */
  uint64_t tw_q; unsigned tw_logn; uint64_t *tab, *itab;
  uint64_t instret;
};

orc_vp *orc_vp_create(uint64_t spm_rows, uint64_t ksk_rows) {
  orc_vp *vp = (orc_vp *)calloc(1, sizeof(orc_vp));
  if (!vp) return NULL;
  vp->spm_rows = spm_rows;
  vp->ksk_rows = ksk_rows;
  vp->iram_depth = ORC_IRAM_DEPTH;
  vp->spm = (uint64_t *)calloc(spm_rows ? spm_rows : 1, ORC_ROW_BYTES);
  vp->ksk = (uint64_t *)calloc(ksk_rows ? ksk_rows : 1, ORC_ROW_BYTES);
  for (unsigned r = 0; r < ORC_NVREG; r++) vp->vreg[r] = (uint64_t *)calloc(ORC_NMAX, 8);
  vp->tab = (uint64_t *)calloc(ORC_NMAX, 8);
  vp->itab = (uint64_t *)calloc(ORC_NMAX, 8);
  return vp;
}

int orc_vp_set_iram_depth(orc_vp *vp, uint32_t depth) {
  if (!vp || depth < ORC_IRAM_DEPTH || depth > ORC_IRAM_DEPTH_MAX) return ORC_ERR_ARG;
  vp->iram_depth = depth;
  return ORC_OK;
}

void orc_vp_destroy(orc_vp *vp) {
  if (!vp) return;
  free(vp->spm); free(vp->ksk);
  for (unsigned r = 0; r < ORC_NVREG; r++) free(vp->vreg[r]);
  free(vp->tab); free(vp->itab);
  free(vp);
}

static uint64_t *space_ptr(orc_vp *vp, int space, uint64_t row, uint64_t bytes) {
  uint64_t rows = space == 0 ? vp->spm_rows : vp->ksk_rows;
  uint64_t *base = space == 0 ? vp->spm : vp->ksk;
  if (space != 0 && space != 1) return NULL;
  if (row > rows || bytes > (rows - row) * (uint64_t)ORC_ROW_BYTES) return NULL;
  return base + row * ORC_LANES;
}

int orc_vp_write(orc_vp *vp, int space, uint64_t row, const void *src, uint64_t bytes) {
  if (!vp || !src) return ORC_ERR_ARG;
  uint64_t *p = space_ptr(vp, space, row, bytes);
  if (!p) return ORC_ERR_ADDR;
  memcpy(p, src, bytes);
  return ORC_OK;
}

int orc_vp_read(orc_vp *vp, int space, uint64_t row, void *dst, uint64_t bytes) {
  if (!vp || !dst) return ORC_ERR_ARG;
  uint64_t *p = space_ptr(vp, space, row, bytes);
  if (!p) return ORC_ERR_ADDR;
  memcpy(dst, p, bytes);
  return ORC_OK;
}

int orc_vp_load_isram(orc_vp *vp, uint32_t base_pc, const uint32_t *words, uint32_t n) {
  if (!vp || !words) return ORC_ERR_ARG;
  if (base_pc > vp->iram_depth || n > vp->iram_depth - base_pc) return ORC_ERR_ADDR;
  for (uint32_t i = 0; i < n; i++) {
    vp->inst[base_pc + i] = words[3 * i];
    vp->imm[base_pc + i] = ((uint64_t)words[3 * i + 1] << 32) | words[3 * i + 2];
  }
  return ORC_OK;
}

static int vl_logn(uint64_t vl_bits, unsigned *logn) {
  if (vl_bits % 64) return ORC_ERR_VL;
  uint64_t n = vl_bits / 64;
  for (unsigned l = 10; l <= 16; l++)
    if (n == (1ull << l)) { *logn = l; return ORC_OK; }
  return ORC_ERR_VL;                       /* ntt_fsm.sv:59-65 admits log2 N in 10..16 */
}

static int ensure_tables(orc_vp *vp, unsigned logn) {
  if (vp->tw_q == vp->q && vp->tw_logn == logn) return ORC_OK;
  uint64_t psi;
  int rc = orc_find_psi(vp->q, logn, &psi);
  if (rc) return rc;
  orc_make_tables(vp->q, psi, logn, vp->tab, vp->itab);
  vp->tw_q = vp->q;
  vp->tw_logn = logn;
  return ORC_OK;
}

/* vle/vse address: addr_gen.v:44 (imm + cnt<<10), pointer select vp_top_full.sv:105-117:
 * row = imm[..:10] + PTR[imm[63:48]]; PTR 0 src0, 1 src1, 2 rslt, others 0; loads with
 * select 15 read the KSK memory at csr_ksk_ptr (:113-114).  The RTL keeps 16 row bits;
 * here the row is bounds-checked against the configured memory instead of wrapping. */
static uint64_t *mem_row(orc_vp *vp, uint64_t imm, int is_load, const uint64_t csr[4],
                         uint64_t nrows_needed) {
  unsigned sel = (unsigned)(imm >> 48);
  uint64_t off = (imm & 0xFFFFFFFFFFFFull) >> 10;
  if (is_load && sel == 15) return space_ptr(vp, 1, off + csr[3], nrows_needed * ORC_ROW_BYTES);
  uint64_t base = sel == 0 ? csr[0] : sel == 1 ? csr[1] : sel == 2 ? csr[2] : 0;
  return space_ptr(vp, 0, off + base, nrows_needed * ORC_ROW_BYTES);
}

int orc_vp_run(orc_vp *vp, uint32_t pc, uint64_t src0_row, uint64_t src1_row,
               uint64_t rslt_row, uint64_t ksk_row, uint64_t step) {
  if (!vp) return ORC_ERR_ARG;
  const uint64_t csr[4] = {src0_row, src1_row, rslt_row, ksk_row};
  for (;; pc++) {
    if (pc >= vp->iram_depth) return ORC_ERR_NOBREAK;
    /* field split: expander.v:123-130 */
    uint32_t inst = vp->inst[pc];
    uint64_t imm = vp->imm[pc];
    unsigned funct6 = inst >> 26, vs2 = (inst >> 20) & 31, vs1 = (inst >> 15) & 31;
    unsigned funct3 = (inst >> 12) & 7, vd = (inst >> 7) & 31;
    vp->instret++;
    uint64_t q = vp->q;
    unsigned logn = 0;
    uint64_t n = 0;
    int is_vec = !(funct6 == 0x04 || funct6 == 0x08 || funct6 == 0x0c || funct6 == 0x10 || funct6 == 0x00);
    if (is_vec) {
      int rc = vl_logn(vp->vl_bits, &logn);
      if (rc) return rc;
      n = 1ull << logn;
      if (funct6 != 0x03 && funct6 != 0x07 && funct6 != 0x19 && funct6 != 0x1d) {     /* arithmetic needs q */
        if (q <= (1ull << 59) || q >= (1ull << 60)) return ORC_ERR_MODULUS;
      }
    }
    switch (funct6) {
      case 0x10: return ORC_OK;                                   /* break  expander.v:141-151 */
      case 0x00: break;                                           /* nop    :69 */
      case 0x04: {                                                /* vsetvl :156-159 */
        unsigned l; int rc = vl_logn(imm, &l); if (rc) return rc;
        vp->vl_bits = imm; break;
      }
      case 0x08:                                                  /* vsetq  :160-163 */
        if (imm <= (1ull << 59) || imm >= (1ull << 60) || !orc_is_prime(imm)) return ORC_ERR_MODULUS;
        vp->q = imm; break;
      case 0x0c: vp->mu = imm; break;                             /* vsetiq :164-167 */
      case 0x01: case 0x05: case 0x09: {                          /* vfqmul/vfqadd/vfqsub :178-373 */
        unsigned op;
        const uint64_t *a, *b = NULL; uint64_t s = 0;
        if (funct3 == 0) {                                        /* .vv */
          if (((vs1 ^ vs2) & 1) == 0) return ORC_ERR_BANK;        /* one read port per bank */
          a = vp->vreg[vs1]; b = vp->vreg[vs2];
          op = funct6 == 0x01 ? 0x00 : funct6 == 0x05 ? 0x01 : 0x02;
        } else if (funct3 == 1) {                                 /* .vs */
          a = vp->vreg[vs1]; s = imm;
          op = funct6 == 0x01 ? 0x04 : funct6 == 0x05 ? 0x05 : 0x06;
        } else if (funct3 == 2 && funct6 == 0x09) {               /* vfqsub.sv: imm - vs2 */
          a = vp->vreg[vs2]; s = imm; op = 0x0a;
        } else return ORC_ERR_DECODE;
        if (s >= 2 * q) return ORC_ERR_ARG;                       /* scalar outside the ALU's domain */
        uint64_t *d = vp->vreg[vd];
        for (uint64_t i = 0; i < n; i++) orc_alu(op, a[i], b ? b[i] : 0, s, q, &d[i], NULL);
        break;
      }
      case 0x0d: {                                                /* vfqmod :374-395 */
        uint64_t *d = vp->vreg[vd]; const uint64_t *a = vp->vreg[vs1];
        for (uint64_t i = 0; i < n; i++) orc_alu(0x03, a[i], 0, 0, q, &d[i], NULL);
        break;
      }
      case 0x11: {                                                /* vcp = a + 0  :396-417 */
        uint64_t *d = vp->vreg[vd]; const uint64_t *a = vp->vreg[vs1];
        for (uint64_t i = 0; i < n; i++) orc_alu(0x05, a[i], 0, 0, q, &d[i], NULL);
        break;
      }
      case 0x15: case 0x19: case 0x1d: case 0x21: {               /* vaut / vroli :460-501,:549-558; vautn, vautz: extensions */
        uint64_t k = step + imm;                                  /* scalar_iconn = csr_step + imm */
        uint64_t *tmp = (uint64_t *)malloc(n * 8);
        if (!tmp) return ORC_ERR_ARG;
        if (funct6 == 0x15) {
          if (!(k & 1)) { free(tmp); return ORC_ERR_AUT; }
          orc_vaut(tmp, vp->vreg[vs1], logn, q, k);
        } else if (funct6 == 0x1d) {
          if (!(k & 1)) { free(tmp); return ORC_ERR_AUT; }
          orc_vautn(tmp, vp->vreg[vs1], logn, k);
        } else if (funct6 == 0x21) {
          if (!(k & 1)) { free(tmp); return ORC_ERR_AUT; }
          orc_vautz(tmp, vp->vreg[vs1], logn, q, k);
        } else {
          orc_vroli(tmp, vp->vreg[vs1], logn, k);
        }
        memcpy(vp->vreg[vd], tmp, n * 8);
        free(tmp);
        break;
      }
      case 0x02: case 0x06: {                                     /* vntt / vintt :418-459 */
        int rc = ensure_tables(vp, logn); if (rc) return rc;
        if (vd != vs1) memcpy(vp->vreg[vd], vp->vreg[vs1], n * 8);
        if (funct6 == 0x02) orc_ntt(vp->vreg[vd], logn, q, vp->tab);
        else orc_intt(vp->vreg[vd], logn, q, vp->itab);
        /* The RTL ping-pongs between vs1 and vd once per stage (ntt_fsm.sv:80,
         * vxu_lane.sv:315-318), so vs1 is left holding an intermediate stage: its
         * contents are UNDEFINED afterwards.  Poison it so a program relying on it fails. */
        if (vd != vs1) memset(vp->vreg[vs1], 0xEE, n * 8);
        break;
      }
      case 0x03: {                                                /* vle :647-661,:681-686 */
        uint64_t *p = mem_row(vp, imm, 1, csr, n / ORC_LANES);
        if (!p) return ORC_ERR_ADDR;
        memcpy(vp->vreg[vd], p, n * 8);
        break;
      }
      case 0x07: {                                                /* vse (source = vs1) :502-523,:687-692 */
        uint64_t *p = mem_row(vp, imm, 0, csr, n / ORC_LANES);
        if (!p) return ORC_ERR_ADDR;
        memcpy(p, vp->vreg[vs1], n * 8);
        break;
      }
      default: return ORC_ERR_DECODE;                             /* RTL: silent no-op (:524-532) */
    }
  }
}

int orc_vp_peek_vreg(orc_vp *vp, unsigned r, uint64_t *dst, uint64_t n) {
  if (!vp || !dst || r >= ORC_NVREG || n > ORC_NMAX) return ORC_ERR_ARG;
  memcpy(dst, vp->vreg[r], n * 8);
  return ORC_OK;
}

uint64_t orc_vp_instret(orc_vp *vp) { return vp ? vp->instret : 0; }
