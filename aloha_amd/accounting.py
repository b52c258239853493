"""ALGORITHMIC bytes of the BASELINE configs: what `bench.py` divides by the measured time to get a
roofline fraction.  Everything is counted in limbs of N words (8N bytes); a unit reads its inputs and writes its outputs
ONCE, twiddles excluded:

    transform (vntt / vintt)                      16 N      read N + write N
    element-wise .vv                              24 N      .vs / vcp / vfqmod / vaut / vroli: 16 N
    fused polynomial product (config 3)           3 * 8N    per limb (a, b in, c out); ISA-materialised 9 * 8N
    key-switch (L limbs, K special, dnum digits)  see `keyswitch_bytes`

The key-switch formula is SURVEY 8d's -- "transforms [L + L(L+K) + 2(K+L)] * 16N + inner product 2 L (L+K) * 8N (KSK read
once) + small terms", written there for the reference's decomposition (one digit per prime: dnum = L; keyswitch.mem:1-122)
-- with dnum digits in place of the L:

    survey_transforms   L + dnum (L + K) + 2 (K + L)
                        ^digits to coefficient form (:4-5,:24-25)
                            ^every digit under every modulus (:9-17,:29-37)
                                             ^mod-down terms to coefficient form (:80-83) and back under every q_i (:94-99)
    key                 2 dnum (L + K) limbs, read once (:46-79)

`survey_total` = survey_transforms * 16N + key * 8N is the figure `frac` is quoted on.  A LEANER count is printed beside it
(`lean_total`): a digit needs no transform under the moduli of its own digit when the automorphism is applied in NTT form
(dnum (L + K) - L transforms there), but the ciphertext in and out (4 L limbs, 6 L for a product) are traffic too.
`keyswitch_bytes` returns the itemised sums so that the harness prints exactly what it used."""
from __future__ import annotations

from typing import Dict


def ntt_bytes(n: int, limbs: int = 1) -> int:
    return 16 * n * limbs


def polymul_bytes(n: int, limbs: int) -> Dict[str, int]:
    """Config 3: c = iNTT(NTT(a) .* NTT(b)) per limb."""
    return {"fused_minimum": 3 * 8 * n * limbs, "isa_materialised": 9 * 8 * n * limbs}


def keyswitch_bytes(n: int, L: int, K: int, dnum: int, op: str = "rotate") -> Dict[str, int]:
    """Itemised algorithmic bytes of ONE rotate / multiply + key-switch."""
    assert op in ("rotate", "multiply")
    survey_transforms = L + dnum * (L + K) + 2 * (K + L)
    lean_transforms = L + (dnum * (L + K) - L) + 2 * (K + L)
    key_limbs = 2 * dnum * (L + K)
    io_limbs = 4 * L + (2 * L if op == "multiply" else 0)        # c0, c1 in, two polynomials out (+ the second factor)
    items = {
        "survey_transforms": survey_transforms, "lean_transforms": lean_transforms, "key_limbs": key_limbs, "io_limbs": io_limbs,
        "survey_total": survey_transforms * 16 * n + key_limbs * 8 * n,
        "lean_total": lean_transforms * 16 * n + key_limbs * 8 * n + io_limbs * 8 * n,
    }
    return items
