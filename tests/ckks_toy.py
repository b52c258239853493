"""A toy RNS-CKKS key generator / encryptor / decryptor in exact integer arithmetic -- test
infrastructure for checking what the key-switch programs MEAN (decryption), independently of
how they are executed.  Key format = the reference's (keyswitch.mem:46-77): one
digit per ciphertext prime, one special prime P, limbs ordered [modulus][digit][component],
component 0 adds to c0 (the "b" part), component 1 to c1; decryption is c0 + c1*s.

Polynomials mod q are multiplied through the pinned oracle transforms; everything else is
Python integers."""
from __future__ import annotations

import numpy as np

import oracle


class ToyCKKS:
    def __init__(self, n: int, moduli, special: int, seed: int = 1):
        self.n, self.moduli, self.special = n, [int(q) for q in moduli], int(special)
        self.L = len(self.moduli)
        self.rng = np.random.default_rng(seed)
        self.s = [int(x) for x in self.rng.integers(-1, 2, n)]             # ternary secret
        self.Q = 1
        for q in self.moduli:
            self.Q *= q

    # ---- helpers ---------------------------------------------------------------------------
    def limb(self, coeffs, q):
        return np.array([int(c) % q for c in coeffs], dtype=np.uint64)

    def ntt(self, coeffs, q):
        return oracle.ntt(self.limb(coeffs, q), q)

    def small_error(self):
        return [int(x) for x in np.rint(self.rng.normal(0, 3.2, self.n))]

    def uniform(self, q):
        return self.rng.integers(0, q, self.n, dtype=np.uint64)

    def mul_ntt(self, a, b, q):
        return np.array([(int(x) * int(y)) % q for x, y in zip(a, b)], dtype=np.uint64)

    def negacyclic(self, a, b):
        """Exact product of integer polynomials mod X^n + 1."""
        n = self.n
        out = [0] * n
        for i, x in enumerate(a):
            if x == 0:
                continue
            for j, y in enumerate(b):
                if y == 0:
                    continue
                k = i + j
                if k < n:
                    out[k] += x * y
                else:
                    out[k - n] -= x * y
        return out

    def automorph(self, a, k):
        n = self.n
        out = [0] * n
        for i, x in enumerate(a):
            t = (i * k) % (2 * n)
            if t < n:
                out[t] = x
            else:
                out[t - n] = -x
        return out

    # ---- scheme ------------------------------------------------------------------------------
    def encrypt(self, m):
        """m: integer coefficients (already scaled).  Returns (2, L, n) NTT-form limbs."""
        e = self.small_error()
        ct = np.empty((2, self.L, self.n), dtype=np.uint64)
        for j, q in enumerate(self.moduli):
            a = self.uniform(q)                                            # NTT form of a uniform polynomial
            s_hat = self.ntt(self.s, q)
            me = self.ntt([x + y for x, y in zip(m, e)], q)
            ct[1, j] = a
            ct[0, j] = np.array([(int(v) - int(u) * int(w)) % q for u, v, w in zip(a, me, s_hat)], dtype=np.uint64)
        return ct

    def decrypt(self, ct):
        """(2, L, n) NTT-form limbs -> centred integer coefficients mod Q."""
        res = []
        for j, q in enumerate(self.moduli):
            s_hat = self.ntt(self.s, q)
            v = np.array([(int(a) + int(b) * int(c)) % q for a, b, c in zip(ct[0, j], ct[1, j], s_hat)], dtype=np.uint64)
            res.append(oracle.intt(v, q))
        out = []
        for i in range(self.n):
            x = 0
            for j, q in enumerate(self.moduli):                            # CRT
                Qj = self.Q // q
                x += int(res[j][i]) * Qj * pow(Qj % q, q - 2, q)
            x %= self.Q
            out.append(x - self.Q if x > self.Q // 2 else x)
        return out

    def switch_key(self, s_from):
        """Key switching s_from -> s: ((L+1)*L*2, n) limbs [modulus][digit][component], NTT form."""
        L, P = self.L, self.special
        mods = self.moduli + [P]
        key = np.empty((L + 1, L, 2, self.n), dtype=np.uint64)
        for d in range(L):
            e = self.small_error()
            for j, m in enumerate(mods):
                a = self.uniform(m)
                s_hat = self.ntt(self.s, m)
                payload = [(P % m) * x for x in s_from] if j == d else [0] * self.n
                pe = self.ntt([x + y for x, y in zip(payload, e)], m)
                key[j, d, 1] = a
                key[j, d, 0] = np.array([(int(v) - int(u) * int(w)) % m for u, v, w in zip(a, pe, s_hat)], dtype=np.uint64)
        return key.reshape(-1, self.n)

    def switch_key_hybrid(self, s_from, specials, alpha):
        """Hybrid key (dnum = ceil(L/alpha) digits, P = prod specials): ((L+K)*dnum*2, n) limbs
        [modulus][digit][component]; digit d carries P*s_from under the primes of D_d, 0 elsewhere."""
        L = self.L
        specials = [int(p) for p in specials]
        mods = self.moduli + specials
        P = 1
        for p in specials:
            P *= p
        dnum = -(-L // alpha)
        key = np.empty((len(mods), dnum, 2, self.n), dtype=np.uint64)
        for d in range(dnum):
            e = self.small_error()
            members = range(d * alpha, min((d + 1) * alpha, L))
            for j, m in enumerate(mods):
                a = self.uniform(m)
                s_hat = self.ntt(self.s, m)
                payload = [(P % m) * x for x in s_from] if j in members else [0] * self.n
                pe = self.ntt([x + y for x, y in zip(payload, e)], m)
                key[j, d, 1] = a
                key[j, d, 0] = np.array([(int(v) - int(u) * int(w)) % m for u, v, w in zip(a, pe, s_hat)], dtype=np.uint64)
        return key.reshape(-1, self.n)
