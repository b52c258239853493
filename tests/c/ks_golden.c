/* tests/c/ks_golden.c -- a plain C99 host of the C ABI (no Python, no C++): the reference's golden rotate
 * (tv/case0_4_4 kernel1: rotation step 2, k = 3^2; fixtures under tests/golden/case0) through aloha_ks_* with the
 * reference's own primes and key -- run_rotate of sim/top/top_noaxilite_tb.sv:522-532 as a C program -- then the same
 * rotate through the reference's instruction image (aloha_vp_load_isram + aloha_vp_run at PC 256), both compared word
 * for word with the golden output.  Built and run by tests/test_c_host.py:
 *     gcc -std=c99 -Wall -Werror -I include tests/c/ks_golden.c -L aloha_amd/lib -laloha_vp -o ks_golden
 *     ./ks_golden tests/golden                                                                        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aloha_vp.h"

#define N 8192
#define CHECK(x) do { aloha_status s_ = (x); if (s_) { fprintf(stderr, "%s -> %s\n", #x, aloha_status_string(s_)); return 1; } } while (0)

static uint64_t *slurp(const char *dir, const char *rel, size_t words) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, rel);
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint64_t *p = (uint64_t *)malloc(words * 8);
  if (fread(p, 8, words, f) != words) { fprintf(stderr, "%s: short read\n", path); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char **argv) {
  const char *dir = argc > 1 ? argv[1] : "tests/golden";
  const uint64_t q[2] = {576460825317867521ull, 576460924102115329ull};   /* src/top/h2_top_no_axilite.sv:31-32 */
  const uint64_t p[1] = {576462951330889729ull};                          /* keyswitch.mem:11 */
  uint64_t *ct = slurp(dir, "case0/kernel1_ct_before_rotate.bin", 4 * N);
  uint64_t *exp = slurp(dir, "case0/kernel1_ct_after_rotate.bin", 4 * N);
  uint64_t *ksk = slurp(dir, "case0/ksk_step2.bin", 12 * N);
  uint64_t *out = (uint64_t *)malloc(4 * N * 8);
  if (aloha_vp_abi_version() != ALOHA_VP_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }

  /* 1. the C host of the key-switch: alpha = K = 1 is the reference's algorithm */
  aloha_ks_config kc;
  memset(&kc, 0, sizeof kc);
  kc.log_n = 13; kc.L = 2; kc.K = 1; kc.alpha = 1; kc.world = 1; kc.rank = 0; kc.batch = 1; kc.moduli = q; kc.specials = p;
  kc.flags = ALOHA_KS_REFERENCE_PHASE1;            /* phase 1 as keyswitch.mem writes it (vaut in the coefficient domain) */
  uint64_t spm_rows = 0, ksk_rows = 0;
  CHECK(aloha_ks_rows_needed(&kc, &spm_rows, &ksk_rows));
  aloha_vp_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = ALOHA_VP_ABI_VERSION; cfg.device = 0; cfg.spm_rows = spm_rows > 65536 ? spm_rows : 65536;
  cfg.ksk_rows = ksk_rows > 65536 ? ksk_rows : 65536; cfg.max_log_n = 13;
  aloha_vp *vp = NULL;
  aloha_ks *ks = NULL;
  CHECK(aloha_vp_create(&cfg, &vp));
  CHECK(aloha_ks_create(vp, &kc, &ks));
  CHECK(aloha_ks_load_key(ks, ksk));
  CHECK(aloha_ks_load_ct(ks, 0, 0, ct));
  CHECK(aloha_ks_rotate(ks, 9));
  CHECK(aloha_ks_read_out(ks, 0, out));
  if (memcmp(out, exp, 4 * N * 8) != 0) { fprintf(stderr, "aloha_ks_rotate differs from the golden rotate\n"); return 1; }
  aloha_ks_info_t info;
  CHECK(aloha_ks_info(ks, &info));
  aloha_ks_destroy(ks);

  /* 2. the reference's own program image: isram words of keyswitch.mem at PC 256, CSRs as run_rotate sets them */
  if (argc > 2) {
    FILE *f = fopen(argv[2], "rb");                              /* packed {inst, imm_hi, imm_lo} words (tests/golden/isram) */
    if (!f) { perror(argv[2]); return 2; }
    static uint32_t words[3 * 4096];
    const size_t n = fread(words, 12, 4096, f);
    fclose(f);
    CHECK(aloha_vp_load_isram(vp, 256, words, (uint32_t)n));
    CHECK(aloha_vp_write(vp, ALOHA_SPACE_KSK, 0, ksk, 12 * N * 8));
    CHECK(aloha_vp_write(vp, ALOHA_SPACE_SPM, 0x000, ct, 4 * N * 8));
    CHECK(aloha_vp_run(vp, 256, 0x000, 0, 0x100, 0, 9));
    CHECK(aloha_vp_read(vp, ALOHA_SPACE_SPM, 0x100, out, 4 * N * 8));
    if (memcmp(out, exp, 4 * N * 8) != 0) { fprintf(stderr, "keyswitch.mem image differs from the golden rotate\n"); return 1; }
  }
  aloha_vp_destroy(vp);
  printf("ks_golden ok: dnum %u, %llu SPM rows, golden rotate reproduced%s\n", info.dnum, (unsigned long long)info.spm_rows,
         argc > 2 ? " (C host and reference image)" : " (C host)");
  free(ct); free(exp); free(ksk); free(out);
  return 0;
}
