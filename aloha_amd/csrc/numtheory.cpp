// numtheory.cpp -- see numtheory.hpp.
#include "numtheory.hpp"

namespace aloha {
namespace nt {

uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

// deterministic Miller-Rabin, valid for all 64-bit inputs with the first twelve prime bases
bool is_prime(uint64_t n) {
  static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return false;
  for (uint64_t b : bases) {
    if (n == b) return true;
    if (n % b == 0) return false;
  }
  uint64_t d = n - 1;
  unsigned s = 0;
  while (!(d & 1)) { d >>= 1; s++; }
  for (uint64_t b : bases) {
    uint64_t x = powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool composite = true;
    for (unsigned r = 1; r < s && composite; r++) {
      x = mulmod(x, x, n);
      if (x == n - 1) composite = false;
    }
    if (composite) return false;
  }
  return true;
}

bool find_psi(uint64_t q, unsigned logn, uint64_t *psi) {
  if (logn < 1 || logn > 16) return false;
  const uint64_t n = 1ull << logn, two_n = n << 1;
  if ((q - 1) % two_n) return false;
  uint64_t g = 0;
  for (uint64_t x = 2; x < 4096 && !g; x++) {
    uint64_t r = powmod(x, (q - 1) / two_n, q);
    if (powmod(r, n, q) == q - 1) g = r;          // order exactly 2N
  }
  if (!g) return false;
  // the primitive 2N-th roots are the odd powers of g; keep the smallest
  const uint64_t g2 = mulmod(g, g, q);
  uint64_t cur = g, best = g;
  for (uint64_t i = 1; i < n; i++) {
    cur = mulmod(cur, g2, q);
    if (cur < best) best = cur;
  }
  *psi = best;
  return true;
}

uint64_t nth_prime(unsigned logn, unsigned idx) {
  const uint64_t two_n = 2ull << logn;
  uint64_t p = ((1ull << 60) - 1) / two_n * two_n + 1;
  while (p >= (1ull << 60)) p -= two_n;
  for (;; p -= two_n) {
    if (is_prime(p)) {
      if (idx == 0) return p;
      idx--;
    }
  }
}

bool make_twiddles(uint64_t q, unsigned logn, TwiddleTables *out) {
  uint64_t psi;
  if (!find_psi(q, logn, &psi)) return false;
  const uint32_t n = 1u << logn;
  const uint64_t ipsi = invmod(psi, q);
  out->psi = psi;
  out->fwd.assign(2 * (size_t)n, 0);
  out->inv.assign(2 * (size_t)n, 0);
  uint64_t p = 1, ip = 1;
  for (uint32_t e = 0; e < n; e++) {
    const uint32_t k = bitrev(e, logn);
    out->fwd[2 * k] = p;
    out->fwd[2 * k + 1] = shoup(p, q);
    out->inv[2 * k] = ip;
    out->inv[2 * k + 1] = shoup(ip, q);
    p = mulmod(p, psi, q);
    ip = mulmod(ip, ipsi, q);
  }
  out->n_inv = invmod(n % q, q);
  out->n_inv_shoup = shoup(out->n_inv, q);
  out->w1n_inv = mulmod(out->inv[2], out->n_inv, q);
  out->w1n_inv_shoup = shoup(out->w1n_inv, q);
  return true;
}

}  // namespace nt
}  // namespace aloha
