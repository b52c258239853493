#!/usr/bin/env python3
"""Per-kernel roofline table on ONE GPU: every kernel class of the library driven
through the C ABI at N = 2^16 on a batch large enough to stream from HBM, timed over repeated
calls; algorithmic bytes per SURVEY 8d (element-wise .vv 24 N, .vs / vcp / vfqmod / vaut 16 N,
transforms 16 N, sums (2T+1) 8 N) against the 8 TB/s HBM peak.

    python tests/perf/bench_kernels.py > gpurun_out/kernels.json

Every workload is checked against the oracle on one limb before it is timed."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402  (checker only)
from aloha_amd import isa  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM, KSK = 0, 1
LOGN, N = 16, 1 << 16
ROWS = N // 128
PEAK = 8000.0


def timeit(fn, sync, reps):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def entry(t, alg_bytes, **extra):
    g = alg_bytes / t / 1e9
    return dict(us=t * 1e6, algorithmic_GBps=g, frac_of_hbm_peak=g / PEAK, **extra)


def fill(vp, nlimbs, q, seed):
    """nlimbs limbs of uniform residues < q from row 0 (host-generated once, replicated on the device side by writes)."""
    rng = np.random.default_rng(seed)
    blk = rng.integers(0, q, 16 * N, dtype=np.uint64)
    for i in range(0, nlimbs, 16):
        vp.write(SPM, i * ROWS, blk[: min(16, nlimbs - i) * N])
    return blk


def main() -> int:
    B, C = 256, 4                                                # a program covers B limbs; C chunks are cycled through, so
    q = avp.nth_prime(LOGN, 0)                                   # each operand spans C*B limbs = 512 MiB (> the 256 MiB Infinity Cache)
    res = {}
    with avp.VP(spm_rows=3 * C * B * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
        blk = fill(vp, 2 * C * B, q, 1)
        a0, b0 = blk[:N], blk[:N]                                # limb 0 of operand A, limb B (= same pattern)
        off = lambda limb: limb * 8 * N

        def program(body_of_limb):
            lines = [f"vsetvl {64 * N:#x}", f"vsetq {q:#x}"]
            for i in range(B):
                lines += body_of_limb(i)
            return isa.assemble("\n".join(lines + ["break"]))

        A0, B0, R0 = 0, C * B * ROWS, 2 * C * B * ROWS               # operand A, operand B, results

        def run_prog(words, check=None, step=0):
            vp.load_isram(0, words)
            def once():
                for c in range(C):
                    assert vp.run(0, src0=A0 + c * B * ROWS, src1=B0 + c * B * ROWS, rslt=R0 + c * B * ROWS, step=step) == 0
            once(); vp.sync()
            if check is not None:
                assert np.array_equal(vp.read(SPM, R0, N), check), "kernel result differs from the oracle"
            return timeit(once, vp.sync, 5) / C

        mulmod = lambda x, y: np.array([(int(u) * int(v)) % q for u, v in zip(x[:4096], y[:4096])], dtype=np.uint64)
        # ---- element-wise .vv (a2): 24 N bytes per limb
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", f"vle v1, src1+{off(i):#x}", "vfqmul.vv v2, v0, v1", f"vse v2, rslt+{off(i):#x}"])
        t = run_prog(w)
        assert np.array_equal(vp.read(SPM, R0, 4096), mulmod(a0, b0))
        res["vfqmul.vv"] = entry(t, 24 * N * B, limbs=B)
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", f"vle v1, src1+{off(i):#x}", "vfqadd.vv v2, v0, v1", f"vse v2, rslt+{off(i):#x}"])
        res["vfqadd.vv"] = entry(run_prog(w, (a0 + b0) % np.uint64(q)), 24 * N * B, limbs=B)
        # ---- .vs (a2): 16 N
        s = 0x123456789abcdef
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", f"vfqmul.vs v2, v0, {s:#x}", f"vse v2, rslt+{off(i):#x}"])
        t = run_prog(w)
        assert np.array_equal(vp.read(SPM, R0, 4096), mulmod(a0, np.full(4096, s, dtype=np.uint64)))
        res["vfqmul.vs"] = entry(t, 16 * N * B, limbs=B)
        # ---- vaut (a5), two coalesced passes; vautn (extension), gather
        k = pow(3, 7, 2 * N)
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", "vaut v2, v0, 0x0", f"vse v2, rslt+{off(i):#x}"])
        res["vaut"] = entry(run_prog(w, oracle.vaut(a0, q, k), step=k), 16 * N * B, limbs=B)
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", "vautn v2, v0, 0x0", f"vse v2, rslt+{off(i):#x}"])
        res["vautn"] = entry(run_prog(w, oracle.vautn(a0, k), step=k), 16 * N * B, limbs=B)
        w = program(lambda i: [f"vle v0, src0+{off(i):#x}", "vroli v2, v0, 0x0", f"vse v2, rslt+{off(i):#x}"])
        res["vroli"] = entry(run_prog(w, oracle.vroli(a0, 12345), step=12345), 16 * N * B, limbs=B)
        # ---- inner products (a9): T terms per accumulator, B/T accumulators -> (2T+1) 8 N bytes each
        T = 16
        def dot_body(i):
            if i >= B // T:
                return []
            out = []
            for d in range(T):
                out += [f"vle v0, src0+{off(i * T + d):#x}", f"vle v1, src1+{off(i * T + d):#x}"]
                out += ["vfqmul.vv v6, v0, v1"] if d == 0 else ["vfqmul.vv v3, v0, v1", "vfqadd.vv v6, v6, v3"]
            return out + [f"vse v6, rslt+{off(i):#x}"]
        w = program(dot_body)
        t = run_prog(w)
        exp = np.zeros(4096, dtype=object)
        for d in range(T):
            exp = (exp + blk[d * N:d * N + 4096].astype(object) * blk[d * N:d * N + 4096].astype(object)) % q
        assert np.array_equal(vp.read(SPM, R0, 4096), np.array(exp, dtype=np.uint64))
        res["inner_product_T16"] = entry(t, (2 * T + 1) * 8 * N * (B // T), accumulators=B // T, terms=T)
        # ---- transforms (a3, a4) on the 256-limb batch (the headline uses 4096 limbs: bench.py)
        BT = C * B
        mods = [avp.nth_prime(LOGN, i) for i in range(16) for _ in range(BT // 16)]
        vp.ntt_rows(SPM, R0, A0, LOGN, mods); vp.sync()
        assert np.array_equal(vp.read(SPM, R0, N), oracle.ntt(a0 % np.uint64(mods[0]), mods[0]))
        res["vntt_1024_limbs"] = entry(timeit(lambda: vp.ntt_rows(SPM, R0, A0, LOGN, mods), vp.sync, 10), 16 * N * BT, limbs=BT)
        res["vintt_1024_limbs"] = entry(timeit(lambda: vp.ntt_rows(SPM, R0, A0, LOGN, mods, inverse=True), vp.sync, 10),
                                        16 * N * BT, limbs=BT)
        # ---- encoder block (f4): 64 cleartexts of N doubles -> 2 limbs each
        z = np.random.default_rng(3).uniform(-1, 1, (64, N))
        vp.encode(SPM, 0, z, LOGN, mods[:2]); vp.sync()
        t = timeit(lambda: vp.encode(SPM, 0, z, LOGN, mods[:2]), vp.sync, 5)
        res["encode_64_cleartexts"] = dict(us=t * 1e6, cleartexts_per_s=64 / t,
                                           note="includes the host-to-device copy of the cleartexts (borrowed host buffer)")
    print(json.dumps(res, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
