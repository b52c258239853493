"""Micro-program generators for the HE-ISA (own code; the reference ships four fixed images).

The reference's instruction ROM holds four hand-written programs for N = 8192, two
ciphertext primes and one special prime (``sim/vp/isram_file_generator/*.mem``, entry points
0/64/160/256).  The engine executes those images unchanged (``VP.load_isram``); this module
generates the *same algorithms* for any ring size N = 2^10..2^16 and any number of RNS limbs,
in the same ISA and the same operand layout:

* ciphertext  = 2 components x L limbs, limb ``c*L + j`` at byte offset ``(c*L + j) * 8N``
  (``tv/README.md:4-9``; offsets 0x0/0x10000/0x20000/0x30000 at N = 8192, L = 2,
  ``mul_plain.mem:4-19``);
* plaintext   = L limbs;
* key-switch key = ``(L+K) * L * 2`` limbs ordered ``[modulus][digit][component]``
  (``keyswitch.mem:46-77``), one digit per ciphertext prime, K = 1 special prime.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import isa

# The three primes of the shipped design (src/top/h2_top_no_axilite.sv:31-32, keyswitch.mem:11)
Q0 = 0x0800001100000001
Q1 = 0x0800002800000001
P_SPECIAL = 0x0800020000404001


def barrett_mu(q: int) -> int:
    """``vsetiq`` operand: floor(2^121 / q) (modalu_tb.sv:113-119).  Only the RTL's Barrett
    unit consumes it; the GPU engine derives its own constants from q."""
    return (1 << 121) // q


def _setq(q: int) -> List[str]:
    return [f"vsetq {q:#x}", f"vsetiq {barrett_mu(q):#x}"]


def _lb(n: int) -> int:
    return 8 * n


def encode_post(n: int, moduli: Sequence[int]) -> str:
    """``vle -> vntt -> vse`` per limb (algorithm of encode_post.mem:1-12)."""
    out = [f"vsetvl {64 * n:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q)
        out += [f"vle v0, src0+{j * _lb(n):#x}", "vntt v2, v0", f"vse v2, rslt+{j * _lb(n):#x}"]
    return "\n".join(out + ["break"])


def intt_rows(n: int, moduli: Sequence[int]) -> str:
    """Inverse transform per limb (no shipped counterpart; mirror of ``encode_post``)."""
    out = [f"vsetvl {64 * n:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q)
        out += [f"vle v0, src0+{j * _lb(n):#x}", "vintt v2, v0", f"vse v2, rslt+{j * _lb(n):#x}"]
    return "\n".join(out + ["break"])


def _ct_pt(n: int, moduli: Sequence[int], op: str, comps: int, pt_is_ct: bool) -> str:
    L = len(moduli)
    out = [f"vsetvl {64 * n:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q)
        for c in range(comps):
            off = (c * L + j) * _lb(n)
            out.append(f"vle v0, src0+{off:#x}")
            if pt_is_ct or c == 0:
                out.append(f"vle v1, src1+{(off if pt_is_ct else j * _lb(n)):#x}")
            out += [f"{op}.vv v2, v0, v1", f"vse v2, rslt+{off:#x}"]
    return "\n".join(out + ["break"])


def mul_plain(n: int, moduli: Sequence[int]) -> str:
    """ct x pt, limb-wise (algorithm of mul_plain.mem:1-20)."""
    return _ct_pt(n, moduli, "vfqmul", 2, pt_is_ct=False)


def hom_add(n: int, moduli: Sequence[int]) -> str:
    """ct + ct, limb-wise (algorithm of hom_add.mem:1-22)."""
    return _ct_pt(n, moduli, "vfqadd", 2, pt_is_ct=True)


def poly_mul(n: int, moduli: Sequence[int]) -> str:
    """c = iNTT(NTT(a) . NTT(b)) per limb (BASELINE config 3): a at src0, b at src1."""
    out = [f"vsetvl {64 * n:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q)
        off = j * _lb(n)
        out += [f"vle v0, src0+{off:#x}", "vntt v2, v0", f"vle v4, src1+{off:#x}", "vntt v1, v4",
                "vfqmul.vv v0, v2, v1", "vintt v2, v0", f"vse v2, rslt+{off:#x}"]
    return "\n".join(out + ["break"])


def poly_mul_batched(n: int, moduli: Sequence[int], scratch_row: int) -> str:
    """Same product as ``poly_mul`` written loop by loop (all NTT(a), all NTT(b), all pointwise
    products, all inverse transforms) so a fusing sequencer runs each loop as one batched launch.
    NTT(b) goes to ``len(moduli)`` scratch limbs at absolute row ``scratch_row``."""
    out = [f"vsetvl {64 * n:#x}"]
    lb = _lb(n)
    for j, q in enumerate(moduli):
        out += _setq(q) + [f"vle v0, src0+{j * lb:#x}", "vntt v2, v0", f"vse v2, rslt+{j * lb:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q) + [f"vle v0, src1+{j * lb:#x}", "vntt v2, v0", f"vse v2, abs+{scratch_row * 1024 + j * lb:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q) + [f"vle v0, rslt+{j * lb:#x}", f"vle v1, abs+{scratch_row * 1024 + j * lb:#x}",
                           "vfqmul.vv v2, v0, v1", f"vse v2, rslt+{j * lb:#x}"]
    for j, q in enumerate(moduli):
        out += _setq(q) + [f"vle v0, rslt+{j * lb:#x}", "vintt v2, v0", f"vse v2, rslt+{j * lb:#x}"]
    return "\n".join(out + ["break"])


def keyswitch(n: int, moduli: Sequence[int], special: int) -> str:
    """Rotate (automorphism k = csr_step) + hybrid key-switch, one digit per prime, one
    special prime, rounding mod-down: the algorithm of keyswitch.mem:1-122 for
    L = len(moduli) limbs, all intermediates in vector registers (L <= 4).

    Register plan (own allocation): even bank temporaries v0 (transform result) / v2, odd bank
    v1 (key limb) / v3 (product) / v5; digits and accumulators in the even bank, rotated c0
    limbs in the odd bank, so every ``.vv`` pairs one even with one odd register
    (expander.v:183-200: one read port per bank)."""
    L = len(moduli)
    lb = _lb(n)
    even = iter(range(4, 32, 2))
    odd = iter(range(7, 32, 2))
    try:
        D = [next(even) for _ in range(L)]                     # digits, coefficient form
        R = [next(odd) for _ in range(L)]                      # rotated c0, NTT form
        ACC = [[next(even) for _ in range(2)] for _ in range(L + 1)]
        T = [next(even) for _ in range(2)]                     # mod-down terms (coefficient form)
    except StopIteration:
        raise ValueError("keyswitch(): too many limbs for the 32-entry register file; "
                         "use keyswitch_streamed()")
    allq = list(moduli) + [special]
    half = special >> 1
    out = [f"vsetvl {64 * n:#x}"]
    for d, q in enumerate(moduli):
        out += _setq(q)
        out += [f"vle v0, src0+{(L + d) * lb:#x}", "vintt v2, v0", f"vaut v{D[d]}, v2, 0x0"]
        out += [f"vle v0, src0+{d * lb:#x}", "vintt v2, v0", "vaut v0, v2, 0x0", f"vntt v{R[d]}, v0"]
    for j, q in enumerate(allq):
        out += _setq(q)
        for d in range(L):
            out += [f"vfqmod v2, v{D[d]}", "vntt v0, v2"]      # digit d under modulus j, NTT form
            for c in range(2):
                out.append(f"vle v1, ksk+{((j * L + d) * 2 + c) * lb:#x}")
                if d == 0:
                    out.append(f"vfqmul.vv v{ACC[j][c]}, v0, v1")
                else:
                    out += ["vfqmul.vv v3, v0, v1", f"vfqadd.vv v{ACC[j][c]}, v{ACC[j][c]}, v3"]
    out += _setq(special)
    for c in range(2):
        out += [f"vintt v{T[c]}, v{ACC[L][c]}", f"vfqadd.vs v{T[c]}, v{T[c]}, {half:#x}"]
    for j, q in enumerate(moduli):
        pinv = pow(special % q, q - 2, q)
        out += _setq(q)
        for c in range(2):
            out += [f"vfqsub.vs v2, v{T[c]}, {half:#x}", "vntt v5, v2",
                    f"vfqsub.vv v0, v{ACC[j][c]}, v5", f"vfqmul.vs v0, v0, {pinv:#x}"]
            if c == 0:
                out.append(f"vfqadd.vv v0, v0, v{R[j]}")
            out.append(f"vse v0, rslt+{(c * L + j) * lb:#x}")
    return "\n".join(out + ["break"])


def assemble(text: str) -> np.ndarray:
    return isa.assemble(text)
