// numtheory.hpp -- host-side parameter generation for the VP engine (primes, roots, twiddles).
//
// The reference bakes these into ROM images at elaboration time
// (sim/vp/tf_rom_generator/tf_rom_generator.sv:28-63,:75-77,:105-111,:147-148) for three
// fixed primes and N = 8192.  Here they are generated on demand for any prime
// 2^59 < q < 2^60 with q == 1 (mod 2N), N = 2^10 .. 2^16, and cached per (q, N).
#pragma once
#include <cstdint>
#include <vector>

namespace aloha {
namespace nt {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
uint64_t powmod(uint64_t a, uint64_t e, uint64_t q);
inline uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }
bool is_prime(uint64_t n);

// floor(w * 2^64 / q): Shoup companion of a constant multiplier w < q
inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }
// floor(2^121 / q): single-word Barrett ratio for 60-bit moduli (cf. vsetiq, modalu_tb.sv:113-119)
inline uint64_t barrett_mu(uint64_t q) { return (uint64_t)((((u128)1) << 121) / q); }

inline uint32_t bitrev(uint32_t x, unsigned len) {
  uint32_t r = 0;
  for (unsigned i = 0; i < len; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

// Numerically smallest primitive 2N-th root of unity mod q.  Returns false when q != 1 mod 2N.
// The reference's three hard-coded psi values (tf_rom_generator.sv:75) follow this rule.
bool find_psi(uint64_t q, unsigned logn, uint64_t *psi);

// idx-th largest prime p < 2^60 with p == 1 (mod 2^(logn+1))
uint64_t nth_prime(unsigned logn, unsigned idx);

// Twiddle tables in the reference's indexing: fwd[k] = psi^bitrev(k, logn),
// inv[k] = psi^-bitrev(k, logn); each entry is the pair (w, floor(w*2^64/q)) so that one
// 16-byte load feeds a Shoup multiplication.
struct TwiddleTables {
  std::vector<uint64_t> fwd;   // 2*N words: w0, w0', w1, w1', ...
  std::vector<uint64_t> inv;   // 2*N words
  uint64_t psi = 0;
  uint64_t n_inv = 0, n_inv_shoup = 0;      // N^-1 mod q (for fused-scaling inverse kernels)
  uint64_t w1n_inv = 0, w1n_inv_shoup = 0;  // inv[1] * N^-1 mod q: the last inverse stage's one twiddle with the scaling folded in
};
bool make_twiddles(uint64_t q, unsigned logn, TwiddleTables *out);

}  // namespace nt
}  // namespace aloha
