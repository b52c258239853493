#!/usr/bin/env python3
"""Soak of the fused key-switch phases on the GPU box over RANDOM shapes (tests/test_ks_fused.py runs four fixed ones):

    python tools/soak_ks.py <first seed> <count>

Per seed: L in [2, 13] ciphertext primes, alpha in [1, 4] primes per digit, K = alpha special primes (sometimes more), one or two
ciphertexts per call, N = 2^16 -- so that a modulus of the inner-product kernel meets every mix of transformed and plain digits
(4 nquad + tr transformed, 0-1 plain: tiles shared by both kinds, all three row shapes as the launch picks them) -- and each of
multiply / rotate / rescale must equal the oracle's ISA machine running the same programs one instruction at a time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (the checker)
import test_ks_fused as tk  # noqa: E402
from aloha_amd import hybrid  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad, shapes = [], set()
t0 = time.time()
for i, seed in enumerate(range(first, first + count)):
    rng = np.random.default_rng(seed)
    alpha = int(rng.integers(1, 5))
    L = int(rng.integers(max(2, alpha), 14))
    K = alpha + int(rng.integers(0, 2)) * int(rng.integers(0, 2))
    batch = 1 + int(rng.integers(0, 4) == 0)
    mods, specials, ct = tk._case(L, K, seed)
    lo = hybrid.HybridLayout(tk.N, mods, specials, alpha, batch=batch)
    sizes = dict(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows)
    shapes.add((L, K, alpha, batch))
    for op in ("multiply", "rotate", "rescale"):
        try:
            o = oracle.OracleVP(**sizes)
            exp, _ = tk._run(o, mods, specials, alpha, ct, op, batch)
            o.close()
            with avp.VP(max_log_n=16, **sizes) as vp:
                got, _ = tk._run(vp, mods, specials, alpha, ct, op, batch)
                fused = vp.stats()["ks_fused_launches"]
            if not np.array_equal(got, exp):
                bad.append((seed, L, K, alpha, batch, op, "differs from the oracle"))
            elif fused < 2:
                bad.append((seed, L, K, alpha, batch, op, f"only {fused} fused launches"))
        except Exception as e:                                   # noqa: BLE001
            bad.append((seed, L, K, alpha, batch, op, repr(e)[:200]))
    print(f"{i + 1} seeds ({L=}, {K=}, {alpha=}, {batch=}), {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print("shapes:", len(shapes), "failures:", bad)
sys.exit(1 if bad else 0)
