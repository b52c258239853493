/* tests/c/oracle_sanitize.c -- the oracle's C restatement (oracle/vp_oracle.c) under AddressSanitizer and UBSan: the
 * reference's golden rotate through the ISA machine with the reference's keyswitch.mem image (every instruction class of
 * the shipped programs: vsetvl/vsetq/vsetiq, vle/vse with all pointer selects, vntt/vintt, vaut, vcp/vfqmod, vfqmul/add/sub,
 * break), transform round trips at the ends of the size range, the automorphism identity of the vautn extension, the
 * RTL Barrett form against exact arithmetic.  Built and run by tests/test_c_host.py (gcc, no GPU, no Python). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/vp_oracle.c"

static uint64_t *slurp(const char *dir, const char *rel, size_t bytes) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, rel);
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint64_t *p = (uint64_t *)malloc(bytes);
  if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read\n", path); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char **argv) {
  const char *dir = argc > 1 ? argv[1] : "tests/golden";
  enum { N = 8192 };
  /* 1. golden rotate through the machine with the reference's own program image at PC 256 */
  uint64_t *ct = slurp(dir, "case0/kernel1_ct_before_rotate.bin", 4 * N * 8);
  uint64_t *expct = slurp(dir, "case0/kernel1_ct_after_rotate.bin", 4 * N * 8);
  uint64_t *ksk = slurp(dir, "case0/ksk_step2.bin", 12 * N * 8);
  uint32_t *img = (uint32_t *)slurp(dir, argv[2], (size_t)atoi(argv[3]) * 12);
  orc_vp *vp = orc_vp_create(1 << 12, 1 << 12);
  if (!vp) return 1;
  if (orc_vp_load_isram(vp, 256, img, (uint32_t)atoi(argv[3]))) return 1;
  if (orc_vp_write(vp, 1, 0, ksk, 12 * N * 8) || orc_vp_write(vp, 0, 0, ct, 4 * N * 8)) return 1;
  if (orc_vp_run(vp, 256, 0x000, 0, 0x100, 0, 9)) { fprintf(stderr, "run failed\n"); return 1; }
  uint64_t *out = (uint64_t *)malloc(4 * N * 8);
  if (orc_vp_read(vp, 0, 0x100, out, 4 * N * 8)) return 1;
  if (memcmp(out, expct, 4 * N * 8)) { fprintf(stderr, "golden rotate mismatch\n"); return 1; }
  orc_vp_destroy(vp);
  /* 2. transforms at both ends of the size range, vaut / vautn identity, roll */
  for (unsigned logn = 10; logn <= 16; logn += 6) {
    const uint32_t n = 1u << logn;
    const uint64_t q = orc_nth_prime(logn, 1);
    uint64_t psi;
    if (orc_find_psi(q, logn, &psi)) return 1;
    uint64_t *tab = malloc(8ull * n), *itab = malloc(8ull * n), *a = malloc(8ull * n), *b = malloc(8ull * n), *c = malloc(8ull * n);
    orc_make_tables(q, psi, logn, tab, itab);
    uint64_t x = 88172645463325252ull;
    for (uint32_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; a[i] = x % q; }
    memcpy(b, a, 8ull * n);
    orc_ntt(b, logn, q, tab); orc_intt(b, logn, q, itab);
    if (memcmp(a, b, 8ull * n)) { fprintf(stderr, "round trip mismatch at 2^%u\n", logn); return 1; }
    const uint64_t k = 5;                                       /* vntt(vaut(a, k)) == vautn(vntt(a), k) */
    orc_vaut(b, a, logn, q, k); orc_ntt(b, logn, q, tab);
    memcpy(c, a, 8ull * n); orc_ntt(c, logn, q, tab);
    uint64_t *d = malloc(8ull * n);
    orc_vautn(d, c, logn, k);
    if (memcmp(b, d, 8ull * n)) { fprintf(stderr, "vautn identity fails at 2^%u\n", logn); return 1; }
    orc_vroli(d, a, logn, 3);
    if (d[0] != a[3]) return 1;
    free(tab); free(itab); free(a); free(b); free(c); free(d);
  }
  /* 3. the RTL's Barrett form == exact product, operands up to 2^60 - 1 */
  {
    const uint64_t q = 576460825317867521ull, mu = orc_barrett_mu(q);
    uint64_t x = 2463534242ull;
    for (int i = 0; i < 200000; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const uint64_t a = x % q, b = (x * 0x9E3779B97F4A7C15ull) % q;
      if (orc_barrett_rtl(a, b, q, mu) != orc_mulmod(a, b, q)) { fprintf(stderr, "barrett mismatch\n"); return 1; }
    }
  }
  free(ct); free(expct); free(ksk); free(img); free(out);
  printf("oracle_sanitize ok\n");
  return 0;
}
