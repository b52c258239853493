// tests/c/ks_plan_sanitize.cpp -- the C host's program generator (csrc/keyswitch_host.cpp) under AddressSanitizer and
// UBSan on the CPU: plans (no aloha_vp) for a spread of layouts, every phase fetched through aloha_ks_program, the
// on-demand programs (hoist of slot 0, rotations in one pass) generated twice.  The aloha_vp_* entry points the host
// calls when it RUNS are stubbed: a plan never reaches them.  Built and run by tests/test_c_host.py (g++, host only).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../aloha_amd/csrc/keyswitch_host.cpp"
#include "../../aloha_amd/csrc/numtheory.cpp"

extern "C" {
aloha_status aloha_vp_load_isram(aloha_vp *, uint32_t, const uint32_t *, uint32_t) { abort(); }
aloha_status aloha_vp_run(aloha_vp *, uint32_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t) { abort(); }
aloha_status aloha_vp_write(aloha_vp *, int, uint64_t, const void *, uint64_t) { abort(); }
aloha_status aloha_vp_read(aloha_vp *, int, uint64_t, void *, uint64_t) { abort(); }
aloha_status aloha_vp_sync(aloha_vp *) { abort(); }
void *aloha_vp_device_ptr(aloha_vp *, int, uint64_t) { abort(); }
void *aloha_vp_stream(aloha_vp *) { abort(); }
int aloha_vp_device(const aloha_vp *) { abort(); }
aloha_status aloha_vp_get_config(const aloha_vp *, aloha_vp_config *) { abort(); }
aloha_status aloha_vp_set_scratch(aloha_vp *, uint64_t, uint64_t) { abort(); }
const char *aloha_vp_last_error(const aloha_vp *) { return ""; }
// the two HIP runtime calls of aloha_ks_comm_init (never reached by a plan): no libamdhip64 in a sanitizer build
hipError_t hipGetDevice(int *) { abort(); }
hipError_t hipSetDevice(int) { abort(); }
}

int main() {
  struct Case { uint32_t logn, L, K, alpha, world, batch, flags; };
  const Case cases[] = {{13, 2, 1, 1, 1, 1, ALOHA_KS_REFERENCE_PHASE1}, {10, 5, 2, 2, 3, 2, 0}, {12, 6, 2, 4, 1, 3, 0},
                        {16, 24, 4, 4, 8, 1, 0}, {16, 32, 4, 4, 8, 2, 0}, {16, 24, 1, 1, 2, 1, 0}, {16, 32, 8, 8, 1, 4, 0}};
  const char *phases[] = {"rot1", "mul1", "mid", "terms", "fin_rot", "fin_mul", "hoist1", "modup", "inner_h", "fin_h", "rescale1", "rescale2",
                          "hoist1_1", "modup_1", "many", "many_terms", "many_fin"};
  size_t total = 0;
  for (const Case &c : cases) {
    std::vector<uint64_t> mods, sp;
    for (uint32_t i = 0; i < c.L; i++) mods.push_back(aloha::nt::nth_prime(c.logn, i));
    for (uint32_t k = 0; k < c.K; k++) sp.push_back(aloha::nt::nth_prime(c.logn, c.L + k));
    for (uint32_t rank = 0; rank < c.world; rank++) {
      aloha_ks_config cfg = {c.logn, c.L, c.K, c.alpha, c.world, rank, c.batch, c.flags, mods.data(), sp.data(), 5};
      aloha_ks *ks = nullptr;
      if (aloha_ks_create(nullptr, &cfg, &ks) != ALOHA_OK) { fprintf(stderr, "create failed\n"); return 1; }
      std::vector<uint64_t> steps;
      std::vector<uint32_t> regions;
      for (uint32_t b = 0; b < c.batch; b++) { steps.push_back(2 * b + 3); regions.push_back(c.batch - 1 - b); }
      for (int rep = 0; rep < 2; rep++) {
        if (aloha_ks_hoist(ks, 1) != ALOHA_OK || aloha_ks_rotate_hoisted_many(ks, steps.data(), regions.data(), c.batch) != ALOHA_OK) return 1;
        steps[0] += 2;                                           // a new step list: the programs are regenerated
      }
      for (const char *ph : phases)
        for (uint32_t idx = 0;; idx++) {
          uint32_t n = 0;
          if (aloha_ks_program(ks, ph, idx, nullptr, &n) != ALOHA_OK) return 1;
          if (!n) break;
          std::vector<uint32_t> w(3 * (size_t)n);
          uint32_t cap = n;
          if (aloha_ks_program(ks, ph, idx, w.data(), &cap) != ALOHA_OK || cap != n) return 1;
          total += n;
        }
      aloha_ks_info_t info;
      uint32_t ct[64], all[64];
      if (aloha_ks_info(ks, &info) != ALOHA_OK || aloha_ks_owned(ks, ct, all) != ALOHA_OK) return 1;
      if (aloha_ks_rotate(ks, 3) == ALOHA_OK) return 1;          // a plan must refuse to run
      aloha_ks_destroy(ks);
    }
  }
  printf("ks_plan_sanitize ok: %zu instructions generated\n", total);
  return 0;
}
