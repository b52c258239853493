"""Hybrid key-switch with several primes per digit and NTT-domain automorphism,
limb-sharded over ranks like ``keyswitch.py``.

The reference's program (``keyswitch.mem:1-122``) uses one digit per ciphertext prime and one
special prime: L*(L+1) digit transforms and a key of 2*L*(L+1) limbs per switch.  Grouping
``alpha`` primes into a digit (dnum = ceil(L/alpha) digits, K special primes, P = prod p_k) cuts
both to dnum*(L+K); applying the automorphism to the NTT-form limbs (the ``vautn`` extension)
removes the iNTT -> vaut -> NTT detour of the rotated c0.  With alpha = 1, K = 1 the arithmetic
below is the reference's own, step for step, and reproduces its golden rotate bit-exactly
(tests) -- which is what pins this generalisation; for alpha > 1 no reference vectors exist and
parity is against the oracle's ISA machine plus decryption semantics (tests/ckks_toy.py).

    digits       D_d = {q_i : i in [d*alpha, (d+1)*alpha)},   Q_d = prod D_d
    phase 1      X[i]  = the NTT-form limb to switch (vautn(c1[i]) or d2[i]);  ROT[i] = vautn(c0[i]) or d0[i]
                 Y[i]  = iNTT(X[i]) * (Q_d/q_i)^-1  mod q_i                          (coefficient form)
    exchange     ALL-GATHER of Y[0..L-1]
    phase 2a     ModUp: for every owned modulus m_j (q's and p's) and digit d with m_j not in D_d
                     SD[j][d] = NTT_j( sum_{i in D_d} Y[i] * (Q_d/q_i mod m_j)  mod m_j )
                 (for m_j in D_d the digit's limb under m_j is X[j] itself)
    phase 2b     acc_c[j] = sum_d SD[j][d] * KEY[j][d][c]
    phase 3      owners of p_k:  W[k][c] = (iNTT(acc_c[p_k]) + floor(P/2)) * (P/p_k)^-1  mod p_k   (a one-term sum + constant)
    exchange     ALL-GATHER of W (2K limbs)
    phase 4      t_c[j] = NTT_j( sum_k W[k][c] * (P/p_k mod q_j) - floor(P/2) );
                 r_c[j] = (acc_c[j] - t_c[j]) * P^-1 mod q_j;   out = (ROT + r_0, [BASE1 +] r_1)

THE PROGRAMS COME FROM ONE GENERATOR: the C host (csrc/keyswitch_host.cpp, ``aloha_ks_program``).  This module holds the
slot layout (``HybridLayout``, what a Python caller needs to place and fetch data) and a driver that fetches the phase
programs from a C plan and runs them on any engine with the VP call surface -- the GPU engine, or the oracle's ISA
machine in the parity tests, which therefore execute word for word what the C host executes.

Everything is HE-ISA: the ModUp / ModDown sums are ``vfqmul.vs`` / ``vfqadd.vv`` chains over memory
operands, which the engine's fusing sequencer evaluates as one pass per output limb and one
launch per program (scalar lazy terms), exactly like the digit-by-key products.  The basis
conversions are the usual fast (approximate) ones: the lifted digit may exceed the true value by
a small multiple of Q_d, which the key's structure cancels mod Q.

Key layout: limb ((jl*dnum + d)*2 + c) for owned modulus jl, digit d, component c -- the
reference's [modulus][digit][component] order (keyswitch.mem:46-77) with dnum digits.

``batch = B`` switches B ciphertexts per call with the same key (B products relinearised with the
one relinearisation key, or B ciphertexts rotated by the same step): every loop of every phase
then covers B times as many limbs, which is what fills the chip when the limbs are spread over 8
ranks (3-4 moduli each) -- the launch count per call does not change.

``hoist()`` + ``rotate_hoisted(k, index)`` rotate one ciphertext by many steps off a single decomposition
(phases 1-2a once; per step phases 2b-4 with the key stored under sigma_k^-1 and the automorphism applied to
the two results): see ``load_key_hoisted``.

``rescale()`` is the leveled-CKKS step after a product, in the same style: the owner of q_{L-1}
publishes v_c = iNTT(c_c[L-1]) + floor(q_{L-1}/2) (one all-gather of 2B limbs), every rank then
forms out_c[j] = (c_c[j] - NTT_j(v_c - floor(q_{L-1}/2) mod q_j)) * q_{L-1}^-1 mod q_j for its
j < L-1 -- the rounded division by the last prime, which decrypts to round(m / q_{L-1}) (tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import isa
from .keyswitch import Exchange, LANES, SPM, KSK, owner


def _prod(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p


def _bitrev(x: np.ndarray, bits: int) -> np.ndarray:
    r = np.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def ntt_automorphism_gather(n: int, k: int) -> np.ndarray:
    """``src`` with  sigma_k(x) in NTT form [j] == x in NTT form [src[j]]  for odd k in [0, 2n): position j of a
    vntt result holds x(psi^(2*brv(j)+1)) and sigma_k(x)(psi^e) = x(psi^(e*k)) (the map of the ``vautn``
    instruction, without its reduction of k to log2 n bits)."""
    assert k & 1
    bits = n.bit_length() - 1
    j = np.arange(n, dtype=np.int64)
    e = ((2 * _bitrev(j, bits) + 1) * int(k)) % (2 * n)
    return _bitrev((e - 1) >> 1, bits)


class HybridLayout:
    """Slots of N/128 rows from ``base_row``; every group holds ``B`` ciphertexts, index*B + b."""

    def __init__(self, n: int, moduli: Sequence[int], specials: Sequence[int], alpha: int, world: int = 1,
                 rank: int = 0, base_row: int = 0, batch: int = 1):
        self.n, self.moduli, self.specials, self.alpha = n, [int(q) for q in moduli], [int(p) for p in specials], alpha
        self.world, self.rank, self.base_row, self.B = world, rank, base_row, batch
        self.L, self.K = len(self.moduli), len(self.specials)
        self.mods_all = self.moduli + self.specials
        self.dnum = -(-self.L // alpha)
        self.rows = n // LANES
        self.owned_ct = [j for j in range(self.L) if owner(j, world) == rank]
        self.owned_all = [j for j in range(self.L + self.K) if owner(j, world) == rank]
        self.owned_sp = [k for k in range(self.K) if owner(self.L + k, world) == rank]
        self.Lr, self.Mr = len(self.owned_ct), len(self.owned_all)
        self.Yr = -(-self.L // world)
        self.Wr = 2 * max(1, max(sum(1 for k in range(self.K) if owner(self.L + k, world) == r) for r in range(world)))
        B, s = batch, 0
        self.ct_in = s; s += 2 * self.Lr * B
        self.ct_b = s; s += 2 * self.Lr * B
        self.x = s; s += self.Lr * B
        self.rot = s; s += self.Lr * B
        self.base1 = s; s += self.Lr * B
        self.y = s; s += world * self.Yr * B
        self.sd = s; s += self.Mr * self.dnum * B
        self.acc = s; s += 2 * self.Mr * B
        self.w = s; s += world * self.Wr * B
        self.out = s; s += 2 * self.Lr * B
        self.s1 = s; s += 2 * self.Lr * B
        self.s2 = s; s += 2 * self.Lr * B
        self.slots = s
        self.P = _prod(self.specials)
        self.half = self.P // 2

    def digit(self, d: int) -> List[int]:
        return list(range(d * self.alpha, min((d + 1) * self.alpha, self.L)))

    def row(self, slot: int) -> int:
        return self.base_row + slot * self.rows

    def off(self, slot: int) -> int:
        return self.row(slot) * 1024

    # slot of (group, index, ciphertext b)
    def s_in(self, c, i, b=0): return self.ct_in + (c * self.Lr + i) * self.B + b
    def s_inb(self, c, i, b=0): return self.ct_b + (c * self.Lr + i) * self.B + b
    def s_x(self, i, b=0): return self.x + i * self.B + b
    def s_rot(self, i, b=0): return self.rot + i * self.B + b
    def s_base1(self, i, b=0): return self.base1 + i * self.B + b
    def s_out(self, c, i, b=0): return self.out + (c * self.Lr + i) * self.B + b
    def s_s1(self, c, i, b=0): return self.s1 + (c * self.Lr + i) * self.B + b
    def s_s2(self, c, i, b=0): return self.s2 + (c * self.Lr + i) * self.B + b
    def s_sd(self, jl, d, b=0): return self.sd + (jl * self.dnum + d) * self.B + b
    def s_acc(self, jl, c, b=0): return self.acc + (2 * jl + c) * self.B + b

    def y_slot(self, i: int, b: int = 0) -> int:                 # gather buffer: rank r owns [r*Yr*B, (r+1)*Yr*B)
        return self.y + (owner(i, self.world) * self.Yr + i // self.world) * self.B + b

    def w_slot(self, k: int, c: int, b: int = 0) -> int:
        r = owner(self.L + k, self.world)
        kl = sum(1 for k2 in range(k) if owner(self.L + k2, self.world) == r)
        return self.w + (r * self.Wr + 2 * kl + c) * self.B + b

    def rescale_slot(self, c: int, b: int = 0) -> int:
        """Coefficient-form last limb (+ rounding offset) in the W gather buffer: chunk of the owner of q_{L-1}."""
        return self.w + (owner(self.L - 1, self.world) * self.Wr + c) * self.B + b

    @property
    def spm_rows_needed(self) -> int:
        return self.base_row + self.slots * self.rows

    @property
    def ksk_limbs(self) -> int:
        return self.Mr * self.dnum * 2


class HybridKeySwitch:
    """Rotate / multiply with the hybrid key-switch on one rank of ``world``; ``batch`` ciphertexts per call."""

    def __init__(self, vp, n: int, moduli: Sequence[int], specials: Sequence[int], alpha: int, world: int = 1, rank: int = 0,
                 exchange: Optional[Exchange] = None, base_row: int = 0, batch: int = 1, reference_phase1: bool = False):
        """``reference_phase1`` (alpha = 1 only): ``rotate`` runs the reference's own phase 1 (iNTT, ``vaut``, NTT) instead of
        the NTT-domain automorphism: identical words except where iNTT(c1) has a zero at a sign-flipped position, which the
        reference lifts as q_d (see ``keyswitch.program_digits``)."""
        from . import ks as cks
        self.vp = vp
        self.lo = lo = HybridLayout(n, moduli, specials, alpha, world, rank, base_row, batch)
        self.ex = exchange or Exchange()
        # a PLAN of the C host (no engine attached): the one generator of the phase programs
        self._rom = int(getattr(vp, "iram_depth", isa.IRAM_DEPTH))  # the engine's instruction ROM: 4096 words unless it was created deeper
        self._plan = cks.KeySwitch(None, n.bit_length() - 1, moduli, specials, alpha, world, rank, batch, base_row,
                                   flags=cks.REFERENCE_PHASE1 if reference_phase1 else 0, rom_depth=self._rom)
        assert (self._plan.owned_ct, self._plan.owned_all, self._plan.dnum) == (lo.owned_ct, lo.owned_all, lo.dnum)
        self._p = {name: self._fetch(name) for name in cks.PHASES}
        total = sum(len(w) for ps in self._p.values() for w in ps)
        self._resident = total <= self._rom
        self._pc = {}
        if self._resident:
            pc = 0
            for name, ps in self._p.items():
                for idx, w in enumerate(ps):
                    vp.load_isram(pc, w)
                    self._pc[(name, idx)] = pc
                    pc += len(w)

    def _fetch(self, name: str):
        progs = []
        while True:
            w = self._plan.program(name, len(progs))
            if not len(w):
                return progs
            progs.append(w)

    # -- data placement -------------------------------------------------------------------------
    def load_key(self, key_full: np.ndarray) -> None:
        """``key_full``: (L+K) * dnum * 2 limbs ordered [modulus][digit][component]."""
        lo = self.lo
        k = np.ascontiguousarray(key_full, dtype=np.uint64).reshape(lo.L + lo.K, lo.dnum * 2 * lo.n)
        for jl, j in enumerate(lo.owned_all):
            self.vp.write(KSK, jl * lo.dnum * 2 * lo.rows, k[j])

    def _load(self, slot_of, ct_full: np.ndarray, b: int) -> None:
        lo = self.lo
        ct = np.ascontiguousarray(ct_full, dtype=np.uint64).reshape(2, lo.L, lo.n)
        for c in range(2):
            for i, j in enumerate(lo.owned_ct):
                self.vp.write(SPM, lo.row(slot_of(c, i, b)), ct[c, j])

    def load_ct(self, ct_full: np.ndarray, b: int = 0) -> None:
        self._load(self.lo.s_in, ct_full, b)

    def load_ct_b(self, ct_full: np.ndarray, b: int = 0) -> None:
        self._load(self.lo.s_inb, ct_full, b)

    def read_out(self, b: int = 0) -> np.ndarray:
        """Owned result limbs of ciphertext ``b``, shape (2, Lr, n)."""
        lo = self.lo
        self.vp.sync()
        out = np.empty((2, lo.Lr, lo.n), dtype=np.uint64)
        for c in range(2):
            for i in range(lo.Lr):
                out[c, i] = self.vp.read(SPM, lo.row(lo.s_out(c, i, b)), lo.n)
        return out

    # -- the ops ----------------------------------------------------------------------------------
    def _run(self, name: str, step: int = 0, ksk: int = 0, scratch=None) -> None:
        """``scratch`` = (first slot, slots): slots of the layout that only this phase's own instructions read back (the
        lifted digits between their transform and the inner product, the mod-down's t): declared to the engine as its
        scratch window for the run (``aloha_vp_set_scratch``), so that it need not write them to memory."""
        set_scratch = getattr(self.vp, "set_scratch", None) if scratch else None
        if set_scratch:
            set_scratch(self.lo.row(scratch[0]), scratch[1] * self.lo.rows)
        try:
            for idx, words in enumerate(self._p[name]):
                if self._resident:
                    pc = self._pc[(name, idx)]
                else:
                    pc = 0
                    self.vp.load_isram(0, words)
                rc = self.vp.run(pc, src0=0, src1=0, rslt=0, ksk=ksk, step=step)
                if rc:
                    raise RuntimeError(f"hybrid key-switch phase {name}[{idx}] failed with status {rc}: "
                                       f"{getattr(self.vp, 'last_error', '')}")
        finally:
            if set_scratch:
                set_scratch(0, 0)

    @property
    def _sd_slots(self):
        return (self.lo.sd, self.lo.Mr * self.lo.dnum * self.lo.B)

    @property
    def _s2_slots(self):
        return (self.lo.s2, 2 * self.lo.Lr * self.lo.B)

    def _switch(self, finish: str) -> None:
        lo = self.lo
        self.ex.all_gather(self.vp, lo.row(lo.y), lo.Yr * lo.B * lo.n, lo.world, lo.rank)
        self._run("mid", scratch=self._sd_slots)
        self._run("terms")
        self.ex.all_gather(self.vp, lo.row(lo.w), lo.Wr * lo.B * lo.n, lo.world, lo.rank)
        self._run(finish, scratch=self._s2_slots)

    def rotate(self, k: int) -> None:
        self._run("rot1", step=k)
        self._switch("fin_rot")

    def multiply(self) -> None:
        self._run("mul1")
        self._switch("fin_mul")

    # -- hoisted rotations: one decomposition, many steps -------------------------------------------------
    def load_key_hoisted(self, key_full: np.ndarray, k: int, index: int) -> None:
        """Key for the rotation by ``k`` (same layout as ``load_key``) into key region ``index``, stored as
        sigma_k^-1(KEY): then  sum_d sigma_k(SD_d) (.) KEY_d == sigma_k( sum_d SD_d (.) sigma_k^-1(KEY_d) )  and the
        automorphism moves from the dnum*(L+K) digit limbs to the two results.  In NTT form sigma is a pure
        permutation of positions, so this is an identity on the stored words."""
        lo = self.lo
        src = ntt_automorphism_gather(lo.n, pow(int(k), -1, 2 * lo.n))
        kf = np.ascontiguousarray(key_full, dtype=np.uint64).reshape(lo.L + lo.K, lo.dnum * 2, lo.n)
        for jl, j in enumerate(lo.owned_all):
            self.vp.write(KSK, index * self.key_rows + jl * lo.dnum * 2 * lo.rows, kf[j][:, src].reshape(-1))

    @property
    def key_rows(self) -> int:
        return self.lo.ksk_limbs * self.lo.rows

    def hoist(self) -> None:
        """Decompose c1 of CT_IN once: digits to coefficient form, all-gather, ModUp, back to NTT form (SD).
        Every ``rotate_hoisted`` afterwards reuses SD: dnum*(L+K) - ... transforms are paid once per ciphertext
        instead of once per rotation."""
        lo = self.lo
        self._run("hoist1")
        self.ex.all_gather(self.vp, lo.row(lo.y), lo.Yr * lo.B * lo.n, lo.world, lo.rank)
        self._run("modup")

    def rotate_hoisted(self, k: int, index: int) -> None:
        """OUT = rotation of CT_IN by ``k`` with the key in region ``index`` (``load_key_hoisted``), after ``hoist()``:
        inner products on the unrotated digits, mod-down, automorphism of the two result polynomials."""
        lo = self.lo
        self._run("inner_h", ksk=index * self.key_rows)
        self._run("terms")
        self.ex.all_gather(self.vp, lo.row(lo.w), lo.Wr * lo.B * lo.n, lo.world, lo.rank)
        self._run("fin_h", step=k, scratch=self._s2_slots)

    def hoist_one(self) -> None:
        """``hoist()`` for the ciphertext in slot 0 only: the other ``batch`` slots then serve as ROTATION slots of
        ``rotate_hoisted_many``."""
        lo = self.lo
        if "hoist1_1" not in self._p:
            self._plan.hoist(first_slot_only=True)                # (a plan: generates the two programs, runs nothing)
            self._p["hoist1_1"], self._p["modup_1"] = self._fetch("hoist1_1"), self._fetch("modup_1")
            self._resident = False
        self._run("hoist1_1")
        self.ex.all_gather(self.vp, lo.row(lo.y), lo.Yr * lo.B * lo.n, lo.world, lo.rank)
        self._run("modup_1")

    def rotate_hoisted_many(self, steps: Sequence[int], regions: Optional[Sequence[int]] = None) -> None:
        """R = len(steps) <= batch rotations of the ciphertext in slot 0 (after ``hoist_one()``) in ONE pass of the
        phases: OUT slot b = rotation by ``steps[b]`` with the key of region ``regions[b]`` (default b).  Word for word
        what R calls of ``rotate_hoisted`` produce, with the launches of a single call: every loop covers R times the
        limbs (the latency-bound small transforms of the mod-down are shared by the R rotations)."""
        lo = self.lo
        steps = [int(k) for k in steps]
        regions = list(range(len(steps))) if regions is None else [int(r) for r in regions]
        assert 0 < len(steps) <= lo.B and len(regions) == len(steps)
        key = ("many", tuple(steps), tuple(regions))
        if key not in self._p:
            self._plan.rotate_hoisted_many(steps, regions)        # (a plan: generates the three programs, runs nothing)
            self._p[key] = self._fetch("many")
            self._p[key + ("terms",)] = self._fetch("many_terms")
            self._p[key + ("fin",)] = self._fetch("many_fin")
            self._resident = False
        self._run(key)
        self._run(key + ("terms",))
        self.ex.all_gather(self.vp, lo.row(lo.w), lo.Wr * lo.B * lo.n, lo.world, lo.rank)
        self._run(key + ("fin",), scratch=self._s2_slots)

    def rescale(self) -> None:
        """CT_IN (L limbs) -> OUT (limbs 0 .. L-2 valid): divide by the last ciphertext prime, rounding.
        The product of two ciphertexts at scale Delta carries Delta^2; this brings it back (no key involved)."""
        lo = self.lo
        self._run("rescale1")
        self.ex.all_gather(self.vp, lo.row(lo.w), lo.Wr * lo.B * lo.n, lo.world, lo.rank)
        self._run("rescale2", scratch=self._s2_slots)
