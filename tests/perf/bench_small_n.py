"""Batched forward / inverse transforms at one small ring size (single-pass kernels, csrc/ntt_single.hip), first limb checked
against the oracle:  python tests/perf/bench_small_n.py <log_n> [limbs] [reps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = 1 << logn
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 28) // (8 * n)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
with avp.VP(spm_rows=B * n // 128, ksk_rows=1, max_log_n=logn) as vp:
    P = int(os.environ.get("PRIMES", "16"))                       # limbs per prime = B / P (bench.py quotes N = 2^13 on one prime)
    mods = [avp.nth_prime(logn, i) for i in range(P) for _ in range(B // P)]
    vp.fill_synthetic(SPM, 0, logn, mods, seed_base=0xA10A + 900000)
    a = vp.read(SPM, 0, n)
    vp.ntt_rows(SPM, 0, 0, logn, mods)
    assert np.array_equal(vp.read(SPM, 0, n), oracle.ntt(a, mods[0]))
    vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=True)
    assert np.array_equal(vp.read(SPM, 0, n), a)
    res = {}
    for name, inv in (("fwd", False), ("inv", True)):
        vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=inv); vp.sync()
        t_settle = time.perf_counter()                            # settled clocks (an idle GPU's first ~100 ms of launches run slow)
        while time.perf_counter() - t_settle < 0.15:
            for _ in range(10):
                vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=inv)
            vp.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=inv)
        vp.sync()
        t = (time.perf_counter() - t0) / reps
        res[name] = {"ntt_per_s": B / t, "algorithmic_GBps": 16 * n * B / t / 1e9, "frac_of_8TBps": 16 * n * B / t / 8e12}
print(json.dumps({"log_n": logn, "limbs": B, **res}))
