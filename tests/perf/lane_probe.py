#!/usr/bin/env python3
"""How much do the key-switch's REAL kernels overlap when they run on two HIP streams?  (round 4, before the engine's second lane)

Two engines on two streams of one process, each with its own memories and the same HMult + relinearisation (N = 2^16, L = 24,
alpha = K = 4) through the C host, stage by stage:
    A = stage 1 (ModUp + column pass, row pass + inner products, inverse transforms of the special accumulators)
    B = stage 0 + stage 2 (tensor product, inverse transforms of the digits | mod-down column pass, row pass with the tail)
timed alone, back to back on ONE stream, and side by side on two.  No dependency crosses the streams here: this is the ceiling of
what concurrency can buy, before any fork / join cost.  Prints one JSON line."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from aloha_amd import ks as cks  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402


def main() -> int:
    logn, L, alpha = 16, int(os.environ.get("PROBE_L", "24")), 4
    n = 1 << logn
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    specials = [avp.nth_prime(logn, L + k) for k in range(alpha)]
    spm, ksk = cks.rows_needed(logn, mods, specials, alpha)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    vps = [avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=logn, stream=s.cuda_stream, iram_depth=16384) for s in streams]
    kss = [cks.KeySwitch(v, logn, mods, specials, alpha) for v in vps]
    all_mods = mods + specials
    for k in kss:
        key = np.concatenate([np.random.default_rng(1000 + j).integers(0, all_mods[j], k.dnum * 2 * n, dtype=np.uint64) for j in range(L + alpha)])
        k.load_key(key)
        ct = np.stack([np.random.default_rng(77 + j).integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
        k.load_ct(ct); k.load_ct_b(ct[::-1].copy())
        k.multiply()
    for v in vps:
        v.sync()

    def A(k): k.stage("multiply", 1)
    def B(k): k.stage("multiply", 0); k.stage("multiply", 2)
    def full(k): k.multiply()

    def timed(body, reps=30):
        for v in vps:
            v.sync()
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                body()
            for v in vps:
                v.sync()
            best.append((time.perf_counter() - t0) / reps * 1e6)
        return sorted(best)[len(best) // 2]

    out = {"config": f"HMult + relin N=2^16 L={L} alpha=K=4, two engines on two streams"}
    out["A_alone_us"] = timed(lambda: A(kss[0]))
    out["B_alone_us"] = timed(lambda: B(kss[1]))
    out["A_then_B_one_stream_us"] = timed(lambda: (A(kss[0]), B(kss[0])))
    out["A_beside_B_two_streams_us"] = timed(lambda: (A(kss[0]), B(kss[1])))
    out["full_one_stream_us"] = timed(lambda: full(kss[0]))
    out["two_full_one_stream_us"] = timed(lambda: (full(kss[0]), full(kss[0])))
    out["two_full_two_streams_us"] = timed(lambda: (full(kss[0]), full(kss[1])))
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
