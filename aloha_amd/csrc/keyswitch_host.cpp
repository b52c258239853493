// keyswitch_host.cpp -- the L-limb key-switch of the VP as a C host (C ABI: aloha_ks_* in include/aloha_vp.h).
//
// The reference has ONE place where limbs interact: the rotate / key-switch micro-program
// (sim/vp/isram_file_generator/keyswitch.mem:1-122; digits :4-43, inner products :46-79, mod-down :80-121), written for
// L = 2 ciphertext primes and one special prime, driven by the test-bench task run_rotate
// (sim/top/top_noaxilite_tb.sv:522-532).  This file is that program for any L, K special primes and alpha primes per
// digit (alpha = 1, K = 1: the reference's algorithm step for step), generated as HE-ISA phase programs and run on an
// aloha_vp handle, with the RNS limbs sharded over the ranks of one node -- limb j (ciphertext and special primes alike)
// on rank j mod world -- and the two exchange steps issued straight on RCCL:
//     phase 1        X = vautn(c1) | d2,  ROT = vautn(c0) | d0;  Y[i] = iNTT(X[i]) * (Q_d/q_i)^-1        per owned limb
//     ncclAllGather  of the L coefficient-form digits Y                 (keyswitch.mem:4-5,:24-25 are what it spreads)
//     phase 2        ModUp sums + NTT under every owned modulus; inner products with the key     (keyswitch.mem:9-14,:46-79)
//     phase 3        owners of special primes: W = (iNTT(acc) + floor(P/2)) * (P/p_k)^-1          (keyswitch.mem:80-83)
//     ncclAllGather  of the 2K mod-down terms W
//     phase 4        t = NTT(sum_k W_k (P/p_k) - floor(P/2));  out = (acc - t) P^-1 + base        (keyswitch.mem:86-121)
// The programs, slot layout and scalars are those of aloha_amd/hybrid.py (HybridLayout, blocks_*): tests/ compares the
// two generators word for word, and the Python classes remain the driver of the oracle's ISA machine in the parity
// tests.  RCCL is loaded with dlopen at first use (librccl.so.1, or the one a hosting process has already mapped), so
// the library itself has no link-time dependency on it and a single-rank user never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/aloha_vp.h"
#include "numtheory.hpp"

namespace {

using aloha::nt::invmod;
using aloha::nt::mulmod;
typedef unsigned long long u64;

constexpr uint32_t kOpcode = 0x0B;
constexpr uint64_t kPtrAbs = 3, kPtrKsk = 15;     // pointer select in imm[63:48] (vp_top_full.sv:105-117; 3 = none = absolute)

struct Ins { uint32_t inst; uint64_t imm; };
typedef std::vector<Ins> Block;

// encodings: expander.v:65-81,:123-130 (unused fields as the shipped programs encode them, cf. aloha_amd/isa.py)
inline Ins enc(uint32_t f6, uint32_t vd, uint32_t vs1, uint32_t vs2, uint32_t f3, uint64_t imm) {
  return Ins{(f6 << 26) | (1u << 25) | (vs2 << 20) | (vs1 << 15) | (f3 << 12) | (vd << 7) | kOpcode, imm};
}
inline Ins i_setvl(uint64_t bits) { return enc(0x04, 0, 0, 0, 2, bits); }
inline Ins i_setq(uint64_t q) { return enc(0x08, 0, 0, 0, 2, q); }
inline Ins i_setiq(uint64_t mu) { return enc(0x0c, 0, 0, 0, 2, mu); }
inline Ins i_break() { return enc(0x10, 0, 0, 0, 2, 0); }
inline Ins i_vle(uint32_t vd, uint64_t ptr, uint64_t off) { return enc(0x03, vd, 0, 0, 1, (ptr << 48) | off); }
inline Ins i_vse(uint32_t vs1, uint64_t ptr, uint64_t off) { return enc(0x07, 0, vs1, 0, 1, (ptr << 48) | off); }
inline Ins i_vautn(uint32_t vd, uint32_t vs1, uint64_t imm) { return enc(0x1d, vd, vs1, 0, 1, imm); }
inline Ins i_vaut(uint32_t vd, uint32_t vs1, uint64_t imm) { return enc(0x15, vd, vs1, 0, 1, imm); }
inline Ins i_vautz(uint32_t vd, uint32_t vs1, uint64_t imm) { return enc(0x21, vd, vs1, 0, 1, imm); }
inline Ins i_vntt(uint32_t vd, uint32_t vs1) { return enc(0x02, vd, vs1, 0, 0, 0); }
inline Ins i_vintt(uint32_t vd, uint32_t vs1) { return enc(0x06, vd, vs1, 0, 0, 0); }
inline Ins i_mul_vv(uint32_t vd, uint32_t a, uint32_t b) { return enc(0x01, vd, a, b, 0, 0); }
inline Ins i_add_vv(uint32_t vd, uint32_t a, uint32_t b) { return enc(0x05, vd, a, b, 0, 0); }
inline Ins i_mul_vs(uint32_t vd, uint32_t a, uint64_t s) { return enc(0x01, vd, a, 0, 1, s); }
inline Ins i_add_vs(uint32_t vd, uint32_t a, uint64_t s) { return enc(0x05, vd, a, 0, 1, s); }
inline Ins i_sub_vs(uint32_t vd, uint32_t a, uint64_t s) { return enc(0x09, vd, a, 0, 1, s); }

void setq(Block &b, uint64_t q) { b.push_back(i_setq(q)); b.push_back(i_setiq(aloha::nt::barrett_mu(q))); }

struct Layout {                     // aloha_amd/hybrid.py: HybridLayout
  uint32_t n = 0, L = 0, K = 0, alpha = 0, world = 1, rank = 0, B = 1, dnum = 0, rows = 0;
  uint32_t rom = ALOHA_IRAM_DEPTH;   // instruction ROM words a program may fill (the reference's 4096, or the handle's deeper ROM)
  uint64_t base_row = 0;
  std::vector<uint64_t> moduli, specials, mods_all;
  std::vector<uint32_t> owned_ct, owned_all, owned_sp;
  uint32_t Lr = 0, Mr = 0, Yr = 0, Wr = 0;
  uint64_t ct_in = 0, ct_b = 0, x = 0, rot = 0, base1 = 0, y = 0, sd = 0, acc = 0, w = 0, out = 0, s1 = 0, s2 = 0, slots = 0;

  static uint32_t owner(uint32_t j, uint32_t world) { return j % world; }
  void init() {
    mods_all = moduli;
    mods_all.insert(mods_all.end(), specials.begin(), specials.end());
    dnum = (L + alpha - 1) / alpha;
    rows = n / 128;
    for (uint32_t j = 0; j < L + K; j++)
      if (owner(j, world) == rank) {
        owned_all.push_back(j);
        if (j < L) owned_ct.push_back(j); else owned_sp.push_back(j - L);
      }
    Lr = (uint32_t)owned_ct.size(); Mr = (uint32_t)owned_all.size();
    Yr = (L + world - 1) / world;
    uint32_t most = 0;
    for (uint32_t r = 0; r < world; r++) {
      uint32_t c = 0;
      for (uint32_t k = 0; k < K; k++) c += owner(L + k, world) == r;
      most = std::max(most, c);
    }
    Wr = 2 * std::max(1u, most);
    uint64_t s = 0;
    ct_in = s; s += 2ull * Lr * B;
    ct_b = s; s += 2ull * Lr * B;
    x = s; s += (uint64_t)Lr * B;
    rot = s; s += (uint64_t)Lr * B;
    base1 = s; s += (uint64_t)Lr * B;
    y = s; s += (uint64_t)world * Yr * B;
    sd = s; s += (uint64_t)Mr * dnum * B;
    acc = s; s += 2ull * Mr * B;
    w = s; s += (uint64_t)world * Wr * B;
    out = s; s += 2ull * Lr * B;
    s1 = s; s += 2ull * Lr * B;
    s2 = s; s += 2ull * Lr * B;
    slots = s;
  }
  std::vector<uint32_t> digit(uint32_t d) const {
    std::vector<uint32_t> v;
    for (uint32_t t = d * alpha; t < std::min((d + 1) * alpha, L); t++) v.push_back(t);
    return v;
  }
  bool in_digit(uint32_t j, uint32_t d) const { return j < L && j / alpha == d; }
  uint64_t row(uint64_t slot) const { return base_row + slot * rows; }
  uint64_t off(uint64_t slot) const { return row(slot) * 1024; }
  uint64_t s_in(uint32_t c, uint32_t i, uint32_t b) const { return ct_in + ((uint64_t)c * Lr + i) * B + b; }
  uint64_t s_inb(uint32_t c, uint32_t i, uint32_t b) const { return ct_b + ((uint64_t)c * Lr + i) * B + b; }
  uint64_t s_x(uint32_t i, uint32_t b) const { return x + (uint64_t)i * B + b; }
  uint64_t s_rot(uint32_t i, uint32_t b) const { return rot + (uint64_t)i * B + b; }
  uint64_t s_base1(uint32_t i, uint32_t b) const { return base1 + (uint64_t)i * B + b; }
  uint64_t s_out(uint32_t c, uint32_t i, uint32_t b) const { return out + ((uint64_t)c * Lr + i) * B + b; }
  uint64_t s_s1(uint32_t c, uint32_t i, uint32_t b) const { return s1 + ((uint64_t)c * Lr + i) * B + b; }
  uint64_t s_s2(uint32_t c, uint32_t i, uint32_t b) const { return s2 + ((uint64_t)c * Lr + i) * B + b; }
  uint64_t rescale_slot(uint32_t c, uint32_t b) const { return w + ((uint64_t)owner(L - 1, world) * Wr + c) * B + b; }
  uint64_t s_sd(uint32_t jl, uint32_t d, uint32_t b) const { return sd + ((uint64_t)jl * dnum + d) * B + b; }
  uint64_t s_acc(uint32_t jl, uint32_t c, uint32_t b) const { return acc + (2ull * jl + c) * B + b; }
  uint64_t y_slot(uint32_t i, uint32_t b) const { return y + ((uint64_t)owner(i, world) * Yr + i / world) * B + b; }
  uint64_t w_slot(uint32_t k, uint32_t c, uint32_t b) const {
    const uint32_t r = owner(L + k, world);
    uint32_t kl = 0;
    for (uint32_t k2 = 0; k2 < k; k2++) kl += owner(L + k2, world) == r;
    return w + ((uint64_t)r * Wr + 2 * kl + c) * B + b;
  }
  uint64_t spm_rows_needed() const { return base_row + slots * rows; }
  uint64_t ksk_limbs() const { return (uint64_t)Mr * dnum * 2; }
  uint32_t local_ct(uint32_t j) const { for (uint32_t i = 0; i < Lr; i++) if (owned_ct[i] == j) return i; return ~0u; }
  uint32_t local_all(uint32_t j) const { for (uint32_t i = 0; i < Mr; i++) if (owned_all[i] == j) return i; return ~0u; }
  // prod_{t in dig, t != skip} moduli[t]  mod m
  uint64_t prod_mod(const std::vector<uint32_t> &dig, uint32_t skip, uint64_t m) const {
    uint64_t p = 1 % m;
    for (uint32_t t : dig) if (t != skip) p = mulmod(p, moduli[t] % m, m);
    return p;
  }
  uint64_t P_mod(uint64_t m, uint32_t skip_k = ~0u) const {      // prod_k specials[k] (k != skip_k)  mod m
    uint64_t p = 1 % m;
    for (uint32_t k = 0; k < K; k++) if (k != skip_k) p = mulmod(p, specials[k] % m, m);
    return p;
  }
  uint64_t half_mod(uint64_t m) const {                          // floor(P/2) mod m = (P - 1) / 2 mod m  (P and m odd)
    const uint64_t pm1 = (P_mod(m) + m - 1) % m;
    return mulmod(pm1, (m + 1) / 2, m);
  }
};

// dst = sum_t mem[srcs[t]] * scalars[t] [- minus] [+ plus]: a chain the sequencer evaluates in one pass (hybrid.py: _scalar_sum)
void scalar_sum(Block &b, const std::vector<uint64_t> &srcs, const std::vector<uint64_t> &scal, uint64_t dst, bool has_minus = false,
                uint64_t minus = 0) {
  for (size_t t = 0; t < srcs.size(); t++) {
    b.push_back(i_vle(0, kPtrAbs, srcs[t]));
    if (t == 0) b.push_back(i_mul_vs(6, 0, scal[t]));
    else { b.push_back(i_mul_vs(3, 0, scal[t])); b.push_back(i_add_vv(6, 6, 3)); }
  }
  if (has_minus) b.push_back(i_sub_vs(6, 6, minus));
  b.push_back(i_vse(6, kPtrAbs, dst));
}

enum Mode { ROTATE, MULTIPLY, HOIST, ROTATE_REF };

// ciphertext slots a loop covers: all `B` of them, or the first `nb` only (hoist_one: slot 0 is the decomposed
// ciphertext, the other slots serve as rotation slots)
struct Slots { uint32_t nb; };

std::vector<Block> blocks_phase1(const Layout &lo, Mode mode, Slots sl) {
  std::vector<Block> bl;
  if (mode == ROTATE_REF) {
    // alpha = 1: phase 1 exactly as the reference writes it (keyswitch.mem:4-7,:24-27,:100-121): inverse transforms, vaut in
    // the coefficient domain, the rotated c0 (and the digit under its own modulus) back to NTT form -- a flipped zero then
    // reaches the basis extension as q_d, as in the RTL (vxu_lane.sv:598)
    for (uint32_t i = 0; i < lo.Lr; i++)
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        Block b;
        setq(b, lo.moduli[lo.owned_ct[i]]);
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(0, i, bi)))); b.push_back(i_vintt(2, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_s1(0, i, bi))));
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(1, i, bi)))); b.push_back(i_vintt(2, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_s1(1, i, bi))));
        bl.push_back(std::move(b));
      }
    for (uint32_t i = 0; i < lo.Lr; i++)
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        const uint32_t j = lo.owned_ct[i];
        Block b;
        setq(b, lo.moduli[j]);
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_s1(1, i, bi)))); b.push_back(i_vaut(2, 0, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.y_slot(j, bi))));
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_s1(0, i, bi)))); b.push_back(i_vaut(2, 0, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_s2(0, i, bi))));
        bl.push_back(std::move(b));
      }
    for (uint32_t i = 0; i < lo.Lr; i++)
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        const uint32_t j = lo.owned_ct[i];
        Block b;
        setq(b, lo.moduli[j]);
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_s2(0, i, bi)))); b.push_back(i_vntt(2, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_rot(i, bi))));
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.y_slot(j, bi)))); b.push_back(i_vntt(2, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_x(i, bi))));
        bl.push_back(std::move(b));
      }
    return bl;
  }
  if (mode != HOIST)
    for (uint32_t i = 0; i < lo.Lr; i++) {
      const uint32_t j = lo.owned_ct[i];
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        Block b;
        setq(b, lo.moduli[j]);
        if (mode == ROTATE) {                                     // NTT-domain automorphism of both components
          b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(0, i, bi)))); b.push_back(i_vautn(2, 0, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_rot(i, bi))));
          b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(1, i, bi)))); b.push_back(i_vautn(2, 0, 0)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_x(i, bi))));
        } else {                                                  // tensor product: d0 and d2
          b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(0, i, bi)))); b.push_back(i_vle(1, kPtrAbs, lo.off(lo.s_inb(0, i, bi))));
          b.push_back(i_mul_vv(2, 0, 1)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_rot(i, bi))));
          b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(1, i, bi)))); b.push_back(i_vle(1, kPtrAbs, lo.off(lo.s_inb(1, i, bi))));
          b.push_back(i_mul_vv(2, 0, 1)); b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_x(i, bi))));
        }
        bl.push_back(std::move(b));
      }
    }
  if (mode == MULTIPLY)
    for (uint32_t i = 0; i < lo.Lr; i++) {                        // d1 = a0*b1 + a1*b0
      const uint32_t j = lo.owned_ct[i];
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        Block b;
        setq(b, lo.moduli[j]);
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(0, i, bi)))); b.push_back(i_vle(1, kPtrAbs, lo.off(lo.s_inb(1, i, bi)))); b.push_back(i_mul_vv(6, 0, 1));
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(1, i, bi)))); b.push_back(i_vle(1, kPtrAbs, lo.off(lo.s_inb(0, i, bi)))); b.push_back(i_mul_vv(3, 0, 1));
        b.push_back(i_add_vv(6, 6, 3)); b.push_back(i_vse(6, kPtrAbs, lo.off(lo.s_base1(i, bi))));
        bl.push_back(std::move(b));
      }
    }
  for (uint32_t i = 0; i < lo.Lr; i++) {                          // coefficient form, times (Q_d/q_j)^-1 mod q_j
    const uint32_t j = lo.owned_ct[i];
    const uint64_t q = lo.moduli[j];
    const uint64_t c = invmod(lo.prod_mod(lo.digit(j / lo.alpha), j, q), q);
    Block b;
    setq(b, q);
    for (uint32_t bi = 0; bi < sl.nb; bi++) {
      const uint64_t x = mode == HOIST ? lo.s_in(1, i, bi) : lo.s_x(i, bi);     // hoisted: c1 itself, not rotated
      b.push_back(i_vle(0, kPtrAbs, lo.off(x))); b.push_back(i_vintt(2, 0)); b.push_back(i_mul_vs(4, 2, c));
      if (mode == ROTATE && lo.alpha == 1) {
        // one prime per digit = the reference's decomposition: the digit must be the reference's vaut(iNTT(c1)) word for
        // word, which stores a sign-flipped zero as q_d (vxu_lane.sv:598; keyswitch.mem:4-5,:24-25) -- vautz puts those back
        b.push_back(i_vautz(6, 4, 0));
        b.push_back(i_vse(6, kPtrAbs, lo.off(lo.y_slot(j, bi))));
      } else {
        b.push_back(i_vse(4, kPtrAbs, lo.off(lo.y_slot(j, bi))));
      }
    }
    bl.push_back(std::move(b));
  }
  return bl;
}

// [jl0, jl1): the owned moduli the blocks cover (a phase may be cut into programs by modulus, see mid_programs)
std::vector<Block> blocks_modup(const Layout &lo, Slots sl, uint32_t jl0 = 0, uint32_t jl1 = ~0u) {
  std::vector<Block> bl;
  jl1 = std::min(jl1, lo.Mr);
  for (uint32_t jl = jl0; jl < jl1; jl++) {                       // basis extension sums
    const uint32_t j = lo.owned_all[jl];
    const uint64_t m = lo.mods_all[j];
    for (uint32_t d = 0; d < lo.dnum; d++) {
      if (lo.in_digit(j, d)) continue;
      const std::vector<uint32_t> dig = lo.digit(d);
      std::vector<uint64_t> scal;
      for (uint32_t i : dig) scal.push_back(lo.prod_mod(dig, i, m));
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        Block b;
        setq(b, m);
        std::vector<uint64_t> srcs;
        for (uint32_t i : dig) srcs.push_back(lo.off(lo.y_slot(i, bi)));
        scalar_sum(b, srcs, scal, lo.off(lo.s_sd(jl, d, bi)));
        bl.push_back(std::move(b));
      }
    }
  }
  for (uint32_t jl = jl0; jl < jl1; jl++) {                       // to NTT form, in place
    const uint32_t j = lo.owned_all[jl];
    Block b;
    setq(b, lo.mods_all[j]);
    for (uint32_t d = 0; d < lo.dnum; d++) {
      if (lo.in_digit(j, d)) continue;
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        const uint64_t s = lo.off(lo.s_sd(jl, d, bi));
        b.push_back(i_vle(0, kPtrAbs, s)); b.push_back(i_vntt(2, 0)); b.push_back(i_vse(2, kPtrAbs, s));
      }
    }
    bl.push_back(std::move(b));
  }
  return bl;
}

// regions (hoisted rotations in one pass): slot b is rotation b of the ONE decomposed ciphertext in slot 0, with its key
// in key region regions[b]
std::vector<Block> blocks_inner(const Layout &lo, bool hoisted, const std::vector<uint32_t> *regions, uint32_t jl0 = 0, uint32_t jl1 = ~0u) {
  std::vector<Block> bl;
  const uint32_t nb = regions ? (uint32_t)regions->size() : lo.B;
  jl1 = std::min(jl1, lo.Mr);
  for (uint32_t jl = jl0; jl < jl1; jl++) {
    const uint32_t j = lo.owned_all[jl];
    for (uint32_t bi = 0; bi < nb; bi++) {
      Block b;
      setq(b, lo.mods_all[j]);
      const uint32_t sb = regions ? 0 : bi;                       // operand slot
      const uint64_t kb = regions ? (uint64_t)(*regions)[bi] * lo.ksk_limbs() : 0;
      for (uint32_t c = 0; c < 2; c++) {
        const uint32_t acc = 6 + 2 * c;
        for (uint32_t d = 0; d < lo.dnum; d++) {
          uint64_t src;
          if (lo.in_digit(j, d)) src = lo.off(hoisted ? lo.s_in(1, lo.local_ct(j), sb) : lo.s_x(lo.local_ct(j), sb));
          else src = lo.off(lo.s_sd(jl, d, sb));
          b.push_back(i_vle(0, kPtrAbs, src));
          b.push_back(i_vle(1, kPtrKsk, (kb + ((uint64_t)jl * lo.dnum + d) * 2 + c) * 8 * lo.n));
          if (d == 0) b.push_back(i_mul_vv(acc, 0, 1));
          else { b.push_back(i_mul_vv(3, 0, 1)); b.push_back(i_add_vv(acc, acc, 3)); }
        }
        b.push_back(i_vse(acc, kPtrAbs, lo.off(lo.s_acc(jl, c, bi))));
      }
      bl.push_back(std::move(b));
    }
  }
  return bl;
}

std::vector<Block> blocks_terms(const Layout &lo, Slots sl) {
  std::vector<Block> bl;
  for (uint32_t k : lo.owned_sp) {                                // one scaled inverse transform per accumulator
    const uint64_t p = lo.specials[k];
    const uint64_t cinv = invmod(lo.P_mod(p, k), p);
    const uint32_t jl = lo.local_all(lo.L + k);
    Block b;
    setq(b, p);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < sl.nb; bi++) {
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_acc(jl, c, bi)))); b.push_back(i_vintt(2, 0)); b.push_back(i_mul_vs(4, 2, cinv));
        b.push_back(i_add_vs(4, 4, mulmod(lo.half_mod(p), cinv, p)));
        b.push_back(i_vse(4, kPtrAbs, lo.off(lo.w_slot(k, c, bi))));
      }
    bl.push_back(std::move(b));
  }
  return bl;
}

// hoisted: base_0 is the unrotated c0 and the automorphism (vautn, CSR step) is applied to both results last;
// steps: slot b is rotation steps[b] of the ciphertext in slot 0 (the step rides in the instruction's immediate)
std::vector<Block> blocks_finish(const Layout &lo, bool add_base1, bool hoisted, const std::vector<uint64_t> *steps) {
  std::vector<Block> bl;
  const uint32_t nb = steps ? (uint32_t)steps->size() : lo.B;
  for (uint32_t i = 0; i < lo.Lr; i++) {                          // P-basis -> q_j, minus the rounding offset
    const uint64_t q = lo.moduli[lo.owned_ct[i]];
    std::vector<uint64_t> scal;
    for (uint32_t k = 0; k < lo.K; k++) scal.push_back(lo.P_mod(q, k));
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < nb; bi++) {
        Block b;
        setq(b, q);
        std::vector<uint64_t> srcs;
        for (uint32_t k = 0; k < lo.K; k++) srcs.push_back(lo.off(lo.w_slot(k, c, bi)));
        scalar_sum(b, srcs, scal, lo.off(lo.s_s2(c, i, bi)), true, lo.half_mod(q));
        bl.push_back(std::move(b));
      }
  }
  for (uint32_t i = 0; i < lo.Lr; i++) {                          // to NTT form, in place
    Block b;
    setq(b, lo.moduli[lo.owned_ct[i]]);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < nb; bi++) {
        const uint64_t s = lo.off(lo.s_s2(c, i, bi));
        b.push_back(i_vle(0, kPtrAbs, s)); b.push_back(i_vntt(2, 0)); b.push_back(i_vse(2, kPtrAbs, s));
      }
    bl.push_back(std::move(b));
  }
  for (uint32_t i = 0; i < lo.Lr; i++) {                          // (acc - t) * P^-1 + base, as a sum of scaled operands
    const uint32_t j = lo.owned_ct[i], jl = lo.local_all(j);
    const uint64_t q = lo.moduli[j], pinv = invmod(lo.P_mod(q), q);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < nb; bi++) {
        std::vector<uint64_t> srcs = {lo.off(lo.s_acc(jl, c, bi)), lo.off(lo.s_s2(c, i, bi))}, scal = {pinv, q - pinv};
        if (c == 0) { srcs.push_back(lo.off(hoisted ? lo.s_in(0, i, steps ? 0 : bi) : lo.s_rot(i, bi))); scal.push_back(1); }
        else if (add_base1) { srcs.push_back(lo.off(lo.s_base1(i, bi))); scal.push_back(1); }
        Block b;
        setq(b, q);
        scalar_sum(b, srcs, scal, lo.off(hoisted ? lo.s_s1(c, i, bi) : lo.s_out(c, i, bi)));
        bl.push_back(std::move(b));
      }
  }
  if (hoisted)
    for (uint32_t i = 0; i < lo.Lr; i++) {
      Block b;
      setq(b, lo.moduli[lo.owned_ct[i]]);
      for (uint32_t c = 0; c < 2; c++)
        for (uint32_t bi = 0; bi < nb; bi++) {
          b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_s1(c, i, bi)))); b.push_back(i_vautn(2, 0, steps ? (*steps)[bi] : 0));
          b.push_back(i_vse(2, kPtrAbs, lo.off(lo.s_out(c, i, bi))));
        }
      bl.push_back(std::move(b));
    }
  return bl;
}

// rescale by the last ciphertext prime (hybrid.py: blocks_rescale): [blocks of the owner of q_last, blocks of every rank]
void blocks_rescale(const Layout &lo, std::vector<Block> *first, std::vector<Block> *second) {
  const uint32_t L = lo.L;
  const uint64_t qL = lo.moduli[L - 1], half = qL / 2;
  const uint32_t iL = lo.local_ct(L - 1);
  if (iL != ~0u) {
    Block b;
    setq(b, qL);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < lo.B; bi++) {
        b.push_back(i_vle(0, kPtrAbs, lo.off(lo.s_in(c, iL, bi)))); b.push_back(i_vintt(2, 0)); b.push_back(i_mul_vs(4, 2, 1));
        b.push_back(i_add_vs(4, 4, half)); b.push_back(i_vse(4, kPtrAbs, lo.off(lo.rescale_slot(c, bi))));
      }
    first->push_back(std::move(b));
  }
  for (uint32_t i = 0; i < lo.Lr; i++) {
    const uint32_t j = lo.owned_ct[i];
    if (j == L - 1) continue;
    const uint64_t q = lo.moduli[j];
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < lo.B; bi++) {                    // t mod q_j, minus the rounding offset
        Block b;
        setq(b, q);
        scalar_sum(b, {lo.off(lo.rescale_slot(c, bi))}, {1}, lo.off(lo.s_s2(c, i, bi)), true, half % q);
        second->push_back(std::move(b));
      }
  }
  for (uint32_t i = 0; i < lo.Lr; i++) {
    const uint32_t j = lo.owned_ct[i];
    if (j == L - 1) continue;
    Block b;
    setq(b, lo.moduli[j]);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < lo.B; bi++) {
        const uint64_t s = lo.off(lo.s_s2(c, i, bi));
        b.push_back(i_vle(0, kPtrAbs, s)); b.push_back(i_vntt(2, 0)); b.push_back(i_vse(2, kPtrAbs, s));
      }
    second->push_back(std::move(b));
  }
  for (uint32_t i = 0; i < lo.Lr; i++) {
    const uint32_t j = lo.owned_ct[i];
    if (j == L - 1) continue;
    const uint64_t q = lo.moduli[j], inv = invmod(qL % q, q);
    for (uint32_t c = 0; c < 2; c++)
      for (uint32_t bi = 0; bi < lo.B; bi++) {
        Block b;
        setq(b, q);
        scalar_sum(b, {lo.off(lo.s_in(c, i, bi)), lo.off(lo.s_s2(c, i, bi))}, {inv, q - inv}, lo.off(lo.s_out(c, i, bi)));
        second->push_back(std::move(b));
      }
  }
}

// greedy packing of blocks into programs that fit the 4096-word ROM (hybrid.py: _pack); a program = 3 words per instruction
bool pack(const std::vector<Block> &blocks, uint32_t n, uint32_t kDepth, std::vector<std::vector<uint32_t>> *progs) {
  std::vector<Ins> cur = {i_setvl(64ull * n)};
  auto close = [&]() {
    cur.push_back(i_break());
    std::vector<uint32_t> w;
    for (const Ins &i : cur) { w.push_back(i.inst); w.push_back((uint32_t)(i.imm >> 32)); w.push_back((uint32_t)i.imm); }
    progs->push_back(std::move(w));
    cur = {i_setvl(64ull * n)};
  };
  for (const Block &b : blocks) {
    if (b.size() + 2 > kDepth) return false;
    if (cur.size() + b.size() + 1 > kDepth) close();
    cur.insert(cur.end(), b.begin(), b.end());
  }
  close();
  return true;
}

// Phase 2 ("mid": ModUp sums, transforms, inner products) as programs.  The three loops of one modulus form a chain --
// sums -> forward transforms -> inner products over the transformed limbs -- that the engine's sequencer runs as two
// fused kernels when it sees the whole chain in ONE run (vp_engine.cpp, "staged batches").  A phase that does not fit
// the 4096-word ROM is therefore cut BY MODULUS, every program holding all three loops of its moduli, not loop by loop.
bool mid_programs(const Layout &lo, std::vector<std::vector<uint32_t>> *progs) {
  const Slots all{lo.B};
  auto size_of = [&](uint32_t a, uint32_t b) {
    size_t n = 2;                                                 // vsetvl + break
    for (const Block &x : blocks_modup(lo, all, a, b)) n += x.size();
    for (const Block &x : blocks_inner(lo, false, nullptr, a, b)) n += x.size();
    return n;
  };
  if (!lo.Mr) return true;
  const uint32_t kDepth = lo.rom;
  const size_t total = size_of(0, lo.Mr);
  uint32_t nchunks = (uint32_t)((total + kDepth - 1) / kDepth);
  for (; nchunks <= lo.Mr; nchunks++) {                           // equal shares of the moduli; one more program if a share does not fit
    const uint32_t per = (lo.Mr + nchunks - 1) / nchunks;
    bool fits = true;
    for (uint32_t a = 0; a < lo.Mr && fits; a += per) fits = size_of(a, a + per) <= kDepth;
    if (!fits) continue;
    for (uint32_t a = 0; a < lo.Mr; a += per) {
      std::vector<Block> bl = blocks_modup(lo, all, a, a + per), inner = blocks_inner(lo, false, nullptr, a, a + per);
      bl.insert(bl.end(), inner.begin(), inner.end());
      const size_t before = progs->size();
      if (!pack(bl, lo.n, lo.rom, progs) || progs->size() != before + 1) return false;
    }
    return true;
  }
  return false;
}

// ---- RCCL, loaded on demand -------------------------------------------------------------------------------------
struct RcclUid { char b[128]; };     // ncclUniqueId (rccl/nccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, RcclUid, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int kNcclUint64 = 5;       // ncclDataType_t: ncclUint64 (rccl/nccl.h)

Rccl &rccl() {
  static Rccl r;
  if (r.lib) return r;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) return r;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
  return r;
}

}  // namespace

struct aloha_ks {
  aloha_vp *vp = nullptr;          // null: a plan only (program generation and inspection)
  Layout lo;
  std::map<std::string, std::vector<std::vector<uint32_t>>> progs;
  std::map<std::pair<std::string, uint32_t>, uint32_t> pc;       // resident programs
  bool resident = false;
  void *comm = nullptr;
  bool own_comm = false;
  bool p2p = false;                // exchanges as world-1 direct sends + receives per rank (one RCCL group) instead of ncclAllGather
  uint64_t bytes_sent = 0, bytes_received = 0;
  std::string many_key;            // (steps, regions) the "many*" programs were generated for
  std::string err;
};

namespace {

aloha_status ks_fail(aloha_ks *ks, aloha_status s, const char *fmt, ...) {
  if (ks) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ks->err = buf;
  }
  return s;
}

// Every modulus -- ciphertext and special primes alike -- must be what the engine's vsetq accepts (a prime in
// (2^59, 2^60): vp_engine.cpp modulus_ok), have a 2N-th root of unity, and differ from all the others: the layout's
// scalars are inverses and products modulo them (a zero modulus would divide by zero, a repeated one has no inverse).
aloha_status check_config(const aloha_ks_config *c) {
  if (!c || !c->moduli || !c->specials) return ALOHA_ERR_ARG;
  if (c->log_n < 10 || c->log_n > 16) return ALOHA_ERR_VL;
  if (!c->L || !c->K || !c->alpha || !c->world || c->rank >= c->world || !c->batch) return ALOHA_ERR_ARG;
  if (c->L + c->K > 4096 || c->batch > 64) return ALOHA_ERR_ARG;
  if (c->rom_depth && (c->rom_depth < ALOHA_IRAM_DEPTH || c->rom_depth > ALOHA_IRAM_DEPTH_MAX)) return ALOHA_ERR_ARG;
  std::vector<uint64_t> all(c->moduli, c->moduli + c->L);
  all.insert(all.end(), c->specials, c->specials + c->K);
  for (uint64_t m : all) {
    if (!(m > (1ull << 59) && m < (1ull << 60)) || !(m & 1) || !aloha::nt::is_prime(m)) return ALOHA_ERR_MODULUS;
    if ((m - 1) % (2ull << c->log_n)) return ALOHA_ERR_ROOT;
  }
  std::sort(all.begin(), all.end());
  for (size_t i = 1; i < all.size(); i++)
    if (all[i] == all[i - 1]) return ALOHA_ERR_MODULUS;
  return ALOHA_OK;
}

Layout layout_of(const aloha_ks_config *c) {
  Layout lo;
  lo.n = 1u << c->log_n; lo.L = c->L; lo.K = c->K; lo.alpha = c->alpha; lo.world = c->world; lo.rank = c->rank; lo.B = c->batch;
  lo.base_row = c->base_row;
  if (c->rom_depth) lo.rom = c->rom_depth;
  lo.moduli.assign(c->moduli, c->moduli + c->L);
  lo.specials.assign(c->specials, c->specials + c->K);
  lo.init();
  return lo;
}

// scratch_slot / scratch_slots: slots of the layout that only this phase's own instructions read back (the lifted digits
// between their transform and the inner product, the mod-down's t): the engine's scratch window for the run
// (aloha_vp_set_scratch), so that it need not write them to memory
aloha_status run_phase(aloha_ks *ks, const char *name, uint64_t step, uint64_t ksk_row = 0, uint64_t scratch_slot = 0, uint64_t scratch_slots = 0) {
  auto it = ks->progs.find(name);
  if (it == ks->progs.end()) return ALOHA_OK;
  struct Window {
    aloha_vp *vp;
    ~Window() { if (vp) (void)aloha_vp_set_scratch(vp, 0, 0); }
  } window{nullptr};
  if (scratch_slots && !it->second.empty()) {
    aloha_status s = aloha_vp_set_scratch(ks->vp, ks->lo.row(scratch_slot), scratch_slots * ks->lo.rows);
    if (s) return ks_fail(ks, s, "scratch window of phase %s: %s", name, aloha_vp_last_error(ks->vp));
    window.vp = ks->vp;
  }
  for (uint32_t idx = 0; idx < it->second.size(); idx++) {
    const std::vector<uint32_t> &w = it->second[idx];
    uint32_t pc = 0;
    if (ks->resident) pc = ks->pc[{name, idx}];
    else {
      aloha_status s = aloha_vp_load_isram(ks->vp, 0, w.data(), (uint32_t)(w.size() / 3));
      if (s) return ks_fail(ks, s, "loading phase %s[%u]: %s", name, idx, aloha_vp_last_error(ks->vp));
    }
    aloha_status s = aloha_vp_run(ks->vp, pc, 0, 0, 0, ksk_row, step);
    if (s) return ks_fail(ks, s, "phase %s[%u]: %s", name, idx, aloha_vp_last_error(ks->vp));
  }
  return ALOHA_OK;
}

// all-gather of `world` chunks of `chunk` words laid out back to back from SPM row `row0` (rank r owns chunk r), in place,
// on the engine's stream: the phase's kernels precede it and the next phase's kernels follow it in stream order
aloha_status all_gather(aloha_ks *ks, uint64_t row0, uint64_t chunk) {
  const Layout &lo = ks->lo;
  if (!ks->comm) return lo.world == 1 ? ALOHA_OK : ks_fail(ks, ALOHA_ERR_ARG, "world = %u but no communicator (aloha_ks_comm_init / aloha_ks_set_comm)", lo.world);
  Rccl &r = rccl();
  if (!r.ok) return ks_fail(ks, ALOHA_ERR_DEVICE, "librccl could not be loaded");
  char *base = (char *)aloha_vp_device_ptr(ks->vp, ALOHA_SPACE_SPM, row0);
  if (!base) return ks_fail(ks, ALOHA_ERR_ADDR, "gather buffer outside the SPM space");
  hipStream_t st = (hipStream_t)aloha_vp_stream(ks->vp);
  int rc = 0;
  if (ks->p2p && lo.world > 1 && r.Send && r.Recv && r.GroupStart && r.GroupEnd) {
    // every GPU pair of an MI355X node has its own xGMI link: world-1 direct sends per rank put 1/(world-1) of the traffic
    // on every link at once, where a ring all-gather moves each chunk over world-1 hops of one link each;
    // staggered so that no two ranks start on the same peer
    rc = r.GroupStart();
    for (uint32_t d = 1; d < lo.world && !rc; d++) {
      const uint32_t to = (lo.rank + d) % lo.world, from = (lo.rank + lo.world - d) % lo.world;
      rc = r.Send(base + (size_t)lo.rank * chunk * 8, (size_t)chunk, kNcclUint64, (int)to, ks->comm, st);
      if (!rc) rc = r.Recv(base + (size_t)from * chunk * 8, (size_t)chunk, kNcclUint64, (int)from, ks->comm, st);
    }
    const int rc2 = r.GroupEnd();
    if (!rc) rc = rc2;
  } else {
    rc = r.AllGather(base + (size_t)lo.rank * chunk * 8, base, (size_t)chunk, kNcclUint64, ks->comm, st);
  }
  if (rc) return ks_fail(ks, ALOHA_ERR_DEVICE, "%s: %s", ks->p2p ? "ncclSend/ncclRecv group" : "ncclAllGather", r.GetErrorString ? r.GetErrorString(rc) : "error");
  ks->bytes_sent += (uint64_t)(lo.world - 1) * chunk * 8;
  ks->bytes_received += (uint64_t)(lo.world - 1) * chunk * 8;
  return ALOHA_OK;
}

aloha_status key_switch(aloha_ks *ks, const char *first, const char *finish, uint64_t step) {
  const Layout &lo = ks->lo;
  aloha_status s = run_phase(ks, first, step);
  if (!s) s = all_gather(ks, lo.row(lo.y), (uint64_t)lo.Yr * lo.B * lo.n);
  if (!s) s = run_phase(ks, "mid", 0, 0, lo.sd, (uint64_t)lo.Mr * lo.dnum * lo.B);
  if (!s) s = run_phase(ks, "terms", 0);
  if (!s) s = all_gather(ks, lo.row(lo.w), (uint64_t)lo.Wr * lo.B * lo.n);
  if (!s) s = run_phase(ks, finish, 0, 0, lo.s2, 2ull * lo.Lr * lo.B);
  return s;
}

}  // namespace

extern "C" {

aloha_status aloha_ks_rows_needed(const aloha_ks_config *cfg, uint64_t *spm_rows, uint64_t *ksk_rows) {
  aloha_status s = check_config(cfg);
  if (s) return s;
  const Layout lo = layout_of(cfg);
  if (spm_rows) *spm_rows = lo.spm_rows_needed();
  if (ksk_rows) *ksk_rows = std::max<uint64_t>(1, lo.ksk_limbs() * lo.rows);
  return ALOHA_OK;
}

aloha_status aloha_ks_create(aloha_vp *vp, const aloha_ks_config *cfg, aloha_ks **out) {
  if (!out) return ALOHA_ERR_ARG;
  *out = nullptr;
  aloha_status s = check_config(cfg);
  if (s) return s;
  aloha_ks *ks = new aloha_ks();
  ks->vp = vp;
  ks->lo = layout_of(cfg);
  const Layout &lo = ks->lo;
  if (vp) {                                                       // the handle must cover this rank's share of the layout
    aloha_vp_config have;
    if (aloha_vp_get_config(vp, &have) != ALOHA_OK) { delete ks; return ALOHA_ERR_ARG; }
    if (have.max_log_n < cfg->log_n) { delete ks; return ALOHA_ERR_VL; }
    if (cfg->rom_depth > have.iram_depth) { delete ks; return ALOHA_ERR_ARG; }
    if (!cfg->rom_depth) ks->lo.rom = have.iram_depth;            // programs may use all of the handle's ROM
    if (have.spm_rows < lo.spm_rows_needed() || have.ksk_rows < lo.ksk_limbs() * lo.rows) { delete ks; return ALOHA_ERR_ADDR; }
  }
  bool ok = true;
  auto add = [&](const char *name, bool cond, const std::vector<Block> &bl) {
    std::vector<std::vector<uint32_t>> p;
    if (cond && !pack(bl, lo.n, lo.rom, &p)) ok = false;
    ks->progs[name] = std::move(p);
  };
  const Slots all{lo.B};
  const bool ref1 = (cfg->flags & ALOHA_KS_REFERENCE_PHASE1) != 0;
  if (ref1 && lo.alpha != 1) { delete ks; return ALOHA_ERR_ARG; }
  add("rot1", lo.Lr, blocks_phase1(lo, ref1 ? ROTATE_REF : ROTATE, all));
  add("mul1", lo.Lr, blocks_phase1(lo, MULTIPLY, all));
  {
    std::vector<std::vector<uint32_t>> p;
    if (lo.Mr && !mid_programs(lo, &p)) ok = false;
    ks->progs["mid"] = std::move(p);
  }
  add("terms", !lo.owned_sp.empty(), blocks_terms(lo, all));
  add("fin_rot", lo.Lr, blocks_finish(lo, false, false, nullptr));
  add("fin_mul", lo.Lr, blocks_finish(lo, true, false, nullptr));
  // hoisted rotations (one decomposition, many steps) and the rescale of leveled CKKS
  add("hoist1", lo.Lr, blocks_phase1(lo, HOIST, all));
  add("modup", lo.Mr, blocks_modup(lo, all));
  add("inner_h", lo.Mr, blocks_inner(lo, true, nullptr));
  add("fin_h", lo.Lr, blocks_finish(lo, false, true, nullptr));
  {
    std::vector<Block> r1, r2;
    blocks_rescale(lo, &r1, &r2);
    add("rescale1", !r1.empty(), r1);
    add("rescale2", !r2.empty(), r2);
  }
  if (!ok) { delete ks; return ALOHA_ERR_ARG; }
  size_t total = 0;
  for (auto &kv : ks->progs) for (auto &w : kv.second) total += w.size() / 3;
  ks->resident = total <= lo.rom;
  if (vp && ks->resident) {
    uint32_t pc = 0;
    for (auto &kv : ks->progs)
      for (uint32_t idx = 0; idx < kv.second.size(); idx++) {
        s = aloha_vp_load_isram(vp, pc, kv.second[idx].data(), (uint32_t)(kv.second[idx].size() / 3));
        if (s) { delete ks; return s; }
        ks->pc[{kv.first, idx}] = pc;
        pc += (uint32_t)(kv.second[idx].size() / 3);
      }
  }
  *out = ks;
  return ALOHA_OK;
}

void aloha_ks_destroy(aloha_ks *ks) {
  if (!ks) return;
  if (ks->comm && ks->own_comm && rccl().ok) {
    if (ks->vp) (void)aloha_vp_sync(ks->vp);
    (void)rccl().CommDestroy(ks->comm);
  }
  delete ks;
}

const char *aloha_ks_last_error(const aloha_ks *ks) { return ks ? ks->err.c_str() : "null handle"; }

aloha_status aloha_ks_nccl_unique_id(void *id128) {
  if (!id128) return ALOHA_ERR_ARG;
  Rccl &r = rccl();
  if (!r.ok) return ALOHA_ERR_DEVICE;
  return r.GetUniqueId(id128) ? ALOHA_ERR_DEVICE : ALOHA_OK;
}

aloha_status aloha_ks_comm_init(aloha_ks *ks, const void *id128) {
  if (!ks || !ks->vp || !id128) return ALOHA_ERR_ARG;
  Rccl &r = rccl();
  if (!r.ok) return ks_fail(ks, ALOHA_ERR_DEVICE, "librccl could not be loaded");
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(aloha_vp_device(ks->vp));                   // the communicator binds to the current device
  RcclUid uid;
  memcpy(uid.b, id128, sizeof uid.b);
  void *comm = nullptr;
  const int rc = r.CommInitRank(&comm, (int)ks->lo.world, uid, (int)ks->lo.rank);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc) return ks_fail(ks, ALOHA_ERR_DEVICE, "ncclCommInitRank: %s", r.GetErrorString ? r.GetErrorString(rc) : "error");
  ks->comm = comm; ks->own_comm = true;
  return ALOHA_OK;
}

aloha_status aloha_ks_set_pairwise(aloha_ks *ks, int on) {
  if (!ks) return ALOHA_ERR_ARG;
  ks->p2p = on != 0;
  return ALOHA_OK;
}

aloha_status aloha_ks_set_comm(aloha_ks *ks, void *nccl_comm) {
  if (!ks) return ALOHA_ERR_ARG;
  ks->comm = nccl_comm; ks->own_comm = false;
  return ALOHA_OK;
}

aloha_status aloha_ks_info(const aloha_ks *ks, aloha_ks_info_t *out) {
  if (!ks || !out) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  memset(out, 0, sizeof *out);
  out->dnum = lo.dnum; out->owned_ct = lo.Lr; out->owned_all = lo.Mr; out->rows_per_limb = lo.rows;
  out->spm_rows = lo.spm_rows_needed(); out->ksk_rows = std::max<uint64_t>(1, lo.ksk_limbs() * lo.rows);
  out->resident = ks->resident ? 1 : 0;
  out->key_rows = lo.ksk_limbs() * lo.rows;
  out->bytes_sent = ks->bytes_sent; out->bytes_received = ks->bytes_received;
  return ALOHA_OK;
}

aloha_status aloha_ks_owned(const aloha_ks *ks, uint32_t *ct_indices, uint32_t *all_indices) {
  if (!ks) return ALOHA_ERR_ARG;
  if (ct_indices) for (uint32_t i = 0; i < ks->lo.Lr; i++) ct_indices[i] = ks->lo.owned_ct[i];
  if (all_indices) for (uint32_t i = 0; i < ks->lo.Mr; i++) all_indices[i] = ks->lo.owned_all[i];
  return ALOHA_OK;
}

aloha_status aloha_ks_program(const aloha_ks *ks, const char *phase, uint32_t index, uint32_t *words, uint32_t *ninst) {
  if (!ks || !phase || !ninst) return ALOHA_ERR_ARG;
  auto it = ks->progs.find(phase);
  if (it == ks->progs.end()) return ALOHA_ERR_ARG;
  if (index >= it->second.size()) { *ninst = 0; return ALOHA_OK; }        // no such program: zero instructions
  const std::vector<uint32_t> &w = it->second[index];
  const uint32_t have = (uint32_t)(w.size() / 3), cap = *ninst;
  *ninst = have;
  if (words) {
    if (cap < have) return ALOHA_ERR_ARG;
    memcpy(words, w.data(), w.size() * 4);
  }
  return ALOHA_OK;
}

aloha_status aloha_ks_load_key(aloha_ks *ks, const uint64_t *key_full) {
  if (!ks || !ks->vp || !key_full) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  const uint64_t per = (uint64_t)lo.dnum * 2 * lo.n;             // words per modulus: [digit][component][N]
  for (uint32_t jl = 0; jl < lo.Mr; jl++) {
    aloha_status s = aloha_vp_write(ks->vp, ALOHA_SPACE_KSK, (uint64_t)jl * lo.dnum * 2 * lo.rows, key_full + lo.owned_all[jl] * per, per * 8);
    if (s) return ks_fail(ks, s, "load_key: %s", aloha_vp_last_error(ks->vp));
  }
  return ALOHA_OK;
}

aloha_status aloha_ks_load_ct(aloha_ks *ks, int which, uint32_t slot, const uint64_t *ct_full) {
  if (!ks || !ks->vp || !ct_full || slot >= ks->lo.B || (which != 0 && which != 1)) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  for (uint32_t c = 0; c < 2; c++)
    for (uint32_t i = 0; i < lo.Lr; i++) {
      const uint64_t s_ = which ? lo.s_inb(c, i, slot) : lo.s_in(c, i, slot);
      aloha_status s = aloha_vp_write(ks->vp, ALOHA_SPACE_SPM, lo.row(s_), ct_full + ((uint64_t)c * lo.L + lo.owned_ct[i]) * lo.n, (uint64_t)lo.n * 8);
      if (s) return ks_fail(ks, s, "load_ct: %s", aloha_vp_last_error(ks->vp));
    }
  return ALOHA_OK;
}

aloha_status aloha_ks_read_out(aloha_ks *ks, uint32_t slot, uint64_t *out) {
  if (!ks || !ks->vp || !out || slot >= ks->lo.B) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  for (uint32_t c = 0; c < 2; c++)
    for (uint32_t i = 0; i < lo.Lr; i++) {
      aloha_status s = aloha_vp_read(ks->vp, ALOHA_SPACE_SPM, lo.row(lo.s_out(c, i, slot)), out + ((uint64_t)c * lo.Lr + i) * lo.n, (uint64_t)lo.n * 8);
      if (s) return ks_fail(ks, s, "read_out: %s", aloha_vp_last_error(ks->vp));
    }
  return ALOHA_OK;
}

aloha_status aloha_ks_rotate(aloha_ks *ks, uint64_t k) {
  if (!ks || !ks->vp) return ALOHA_ERR_ARG;
  return key_switch(ks, "rot1", "fin_rot", k);
}

aloha_status aloha_ks_multiply(aloha_ks *ks) {
  if (!ks || !ks->vp) return ALOHA_ERR_ARG;
  return key_switch(ks, "mul1", "fin_mul", 0);
}

/* For hosts that carry the two exchanges themselves (MPI, sockets, a test harness with gloo): the key-switch stage by
 * stage.  stage 0 = phase 1; [all-gather of buffer 0]; stage 1 = ModUp, inner products, mod-down terms; [all-gather of
 * buffer 1]; stage 2 = finish.  A gather buffer is world chunks of chunk_words words back to back from SPM row row0; rank
 * r owns chunk r and must receive the others before the next stage. */
aloha_status aloha_ks_stage(aloha_ks *ks, int op, int stage, uint64_t k) {
  if (!ks || !ks->vp || (op != 0 && op != 1) || stage < 0 || stage > 2) return ALOHA_ERR_ARG;
  if (stage == 0) return run_phase(ks, op ? "mul1" : "rot1", op ? 0 : k);
  const Layout &lo = ks->lo;
  if (stage == 2) return run_phase(ks, op ? "fin_mul" : "fin_rot", 0, 0, lo.s2, 2ull * lo.Lr * lo.B);
  aloha_status s = run_phase(ks, "mid", 0, 0, lo.sd, (uint64_t)lo.Mr * lo.dnum * lo.B);
  if (!s) s = run_phase(ks, "terms", 0);
  return s;
}

aloha_status aloha_ks_gather_buffer(const aloha_ks *ks, int which, uint64_t *row0, uint64_t *chunk_words) {
  if (!ks || (which != 0 && which != 1) || !row0 || !chunk_words) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  *row0 = lo.row(which ? lo.w : lo.y);
  *chunk_words = (uint64_t)(which ? lo.Wr : lo.Yr) * lo.B * lo.n;
  return ALOHA_OK;
}

/* ---- hoisted rotations: one decomposition of the ciphertext, many rotation steps (hybrid.py: hoist / rotate_hoisted /
 * rotate_hoisted_many; BASELINE config 5 is a BATCH of rotations) -------------------------------------------------- */
aloha_status aloha_ks_load_key_hoisted(aloha_ks *ks, const uint64_t *key_full, uint64_t k, uint32_t index) {
  if (!ks || !ks->vp || !key_full || !(k & 1)) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  // stored as sigma_k^-1(KEY): in NTT form a pure permutation of positions, position j <- brv(((2 brv(j) + 1) k^-1 mod 2N - 1) / 2)
  const uint32_t n = lo.n, bits = 31 - (uint32_t)__builtin_clz(n);
  uint64_t kinv = 1;                                              // inverse of k mod 2N (Newton: doubles the correct bits)
  for (int it = 0; it < 6; it++) kinv = kinv * (2 - k * kinv);
  kinv &= 2ull * n - 1;
  std::vector<uint32_t> src(n);
  for (uint32_t j = 0; j < n; j++) {
    const uint64_t e = ((2ull * aloha::nt::bitrev(j, bits) + 1) * kinv) & (2ull * n - 1);
    src[j] = aloha::nt::bitrev((uint32_t)((e - 1) >> 1), bits);
  }
  const uint64_t per = (uint64_t)lo.dnum * 2 * n, key_rows = lo.ksk_limbs() * lo.rows;
  std::vector<uint64_t> buf(per);
  for (uint32_t jl = 0; jl < lo.Mr; jl++) {
    const uint64_t *from = key_full + lo.owned_all[jl] * per;
    for (uint64_t limb = 0; limb < (uint64_t)lo.dnum * 2; limb++)
      for (uint32_t j = 0; j < n; j++) buf[limb * n + j] = from[limb * n + src[j]];
    aloha_status s = aloha_vp_write(ks->vp, ALOHA_SPACE_KSK, index * key_rows + (uint64_t)jl * lo.dnum * 2 * lo.rows, buf.data(), per * 8);
    if (s) return ks_fail(ks, s, "load_key_hoisted: %s", aloha_vp_last_error(ks->vp));
  }
  return ALOHA_OK;
}

// programs generated after create (hoist of slot 0 only; rotations in one pass) are not resident: from then on every
// program is loaded at address 0 before it runs, as in hybrid.py
static bool set_dynamic(aloha_ks *ks, const char *name, bool cond, const std::vector<Block> &bl) {
  std::vector<std::vector<uint32_t>> p;
  if (cond && !pack(bl, ks->lo.n, ks->lo.rom, &p)) return false;
  ks->progs[name] = std::move(p);
  ks->resident = false;
  return true;
}

aloha_status aloha_ks_hoist(aloha_ks *ks, int first_slot_only) {
  if (!ks) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  if (first_slot_only && !ks->progs.count("hoist1_1")) {
    if (!set_dynamic(ks, "hoist1_1", lo.Lr, blocks_phase1(lo, HOIST, Slots{1})) || !set_dynamic(ks, "modup_1", lo.Mr, blocks_modup(lo, Slots{1})))
      return ks_fail(ks, ALOHA_ERR_ARG, "a block exceeds the instruction ROM");
  }
  if (!ks->vp) return ALOHA_OK;                                   // a plan: programs only
  aloha_status s = run_phase(ks, first_slot_only ? "hoist1_1" : "hoist1", 0);
  if (!s) s = all_gather(ks, lo.row(lo.y), (uint64_t)lo.Yr * lo.B * lo.n);
  if (!s) s = run_phase(ks, first_slot_only ? "modup_1" : "modup", 0);
  return s;
}

aloha_status aloha_ks_rotate_hoisted(aloha_ks *ks, uint64_t k, uint32_t index) {
  if (!ks || !ks->vp) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  aloha_status s = run_phase(ks, "inner_h", 0, index * lo.ksk_limbs() * lo.rows);
  if (!s) s = run_phase(ks, "terms", 0);
  if (!s) s = all_gather(ks, lo.row(lo.w), (uint64_t)lo.Wr * lo.B * lo.n);
  if (!s) s = run_phase(ks, "fin_h", k, 0, lo.s2, 2ull * lo.Lr * lo.B);
  return s;
}

aloha_status aloha_ks_rotate_hoisted_many(aloha_ks *ks, const uint64_t *steps, const uint32_t *regions, uint32_t count) {
  if (!ks || !steps || !count || count > ks->lo.B) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  std::vector<uint64_t> st(steps, steps + count);
  std::vector<uint32_t> rg(count);
  for (uint32_t i = 0; i < count; i++) rg[i] = regions ? regions[i] : i;
  std::string key = "many";
  for (uint32_t i = 0; i < count; i++) key += ":" + std::to_string(st[i]) + "@" + std::to_string(rg[i]);
  if (ks->many_key != key) {
    if (!set_dynamic(ks, "many", lo.Mr, blocks_inner(lo, true, &rg)) || !set_dynamic(ks, "many_terms", !lo.owned_sp.empty(), blocks_terms(lo, Slots{count})) ||
        !set_dynamic(ks, "many_fin", lo.Lr, blocks_finish(lo, false, true, &st)))
      return ks_fail(ks, ALOHA_ERR_ARG, "a block exceeds the instruction ROM");
    ks->many_key = key;
  }
  if (!ks->vp) return ALOHA_OK;
  aloha_status s = run_phase(ks, "many", 0);
  if (!s) s = run_phase(ks, "many_terms", 0);
  if (!s) s = all_gather(ks, lo.row(lo.w), (uint64_t)lo.Wr * lo.B * lo.n);
  if (!s) s = run_phase(ks, "many_fin", 0, 0, lo.s2, 2ull * lo.Lr * lo.B);
  return s;
}

/* CT (slot-wise, L limbs) -> OUT (limbs 0 .. L-2 valid): rounded division by the last ciphertext prime -- the step after
 * a product in leveled CKKS; one all-gather of the coefficient-form last limb (2 x batch limbs) */
aloha_status aloha_ks_rescale(aloha_ks *ks) {
  if (!ks || !ks->vp) return ALOHA_ERR_ARG;
  const Layout &lo = ks->lo;
  aloha_status s = run_phase(ks, "rescale1", 0);
  if (!s) s = all_gather(ks, lo.row(lo.w), (uint64_t)lo.Wr * lo.B * lo.n);
  if (!s) s = run_phase(ks, "rescale2", 0, 0, lo.s2, 2ull * lo.Lr * lo.B);
  return s;
}

}  // extern "C"
