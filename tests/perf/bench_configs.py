#!/usr/bin/env python3
"""Timings of BASELINE.json configs 2-5 on ONE GPU (the headline config is bench.py).

    python tests/perf/bench_configs.py > gpurun_out/configs.json

Every workload is checked against the oracle on a sample before it is timed.  Numbers go to
DESIGN.md section 6; they are not the headline metric.
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402  (checker only)
from aloha_amd import isa, keyswitch, programs  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM, KSK = 0, 1


def timeit(fn, sync, reps):
    # median of up to five blocks of back-to-back calls (rare 20-80 ms pauses of the host between two launches would
    # otherwise be the whole figure of a short operation: profiles/r03_z_launch_gaps.txt)
    fn(); sync()
    nblk = 5 if reps >= 50 else 3 if reps >= 15 else 1
    per = max(1, reps // nblk)
    ts = []
    for _ in range(nblk):
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        sync()
        ts.append((time.perf_counter() - t0) / per)
    return sorted(ts)[len(ts) // 2]


def config2():
    logn, n = 12, 1 << 12
    q = avp.nth_prime(logn, 0)
    rng = np.random.default_rng(2)
    out = {}
    with avp.VP(spm_rows=4096 * n // 128, ksk_rows=1, max_log_n=12) as vp:
        a = rng.integers(0, q, n, dtype=np.uint64)
        vp.write(SPM, 0, a)
        vp.ntt_rows(SPM, n // 128, 0, logn, [q])
        assert np.array_equal(vp.read(SPM, n // 128, n), oracle.ntt(a, q))
        t1 = timeit(lambda: (vp.ntt_rows(SPM, 0, 0, logn, [q]), vp.sync()), vp.sync, 200)
        out["single_ntt_latency_us"] = t1 * 1e6
        B = 4096
        mods = [q] * B
        tb = timeit(lambda: vp.ntt_rows(SPM, 0, 0, logn, mods), vp.sync, 50)
        out["batched_ntt_per_s"] = B / tb
        out["batched_algorithmic_GBps"] = 16 * n * B / tb / 1e9
    return out


def small_n_batched():
    """Batched transform rates at N = 2^12..2^15 (the reference's own size is 2^13), both directions."""
    out = {}
    for logn in (12, 13, 14, 15):
        n = 1 << logn
        B = (1 << 28) // (8 * n)                                 # 256 MiB of limbs
        with avp.VP(spm_rows=B * n // 128, ksk_rows=1, max_log_n=logn) as vp:
            mods = [avp.nth_prime(logn, i) for i in range(16) for _ in range(B // 16)]
            rng = np.random.default_rng(logn)
            a = rng.integers(0, mods[0], n, dtype=np.uint64)
            vp.write(SPM, 0, a)
            vp.ntt_rows(SPM, 0, 0, logn, mods)
            assert np.array_equal(vp.read(SPM, 0, n), oracle.ntt(a, mods[0]))
            vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=True)
            assert np.array_equal(vp.read(SPM, 0, n), a)
            tf = timeit(lambda: vp.ntt_rows(SPM, 0, 0, logn, mods), vp.sync, 30)
            ti = timeit(lambda: vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=True), vp.sync, 30)
            out[f"n{n}"] = {"limbs": B, "fwd_ntt_per_s": B / tf, "fwd_algorithmic_GBps": 16 * n * B / tf / 1e9,
                            "inv_ntt_per_s": B / ti, "inv_algorithmic_GBps": 16 * n * B / ti / 1e9}
    return out


def config3():
    logn, n, L = 14, 1 << 14, 8
    rows = n // 128
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    rng = np.random.default_rng(3)
    a = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods])
    b = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods])
    with avp.VP(spm_rows=4 * L * rows, ksk_rows=1, max_log_n=14) as vp:
        vp.write(SPM, 0, a.reshape(-1)); vp.write(SPM, L * rows, b.reshape(-1))
        # the product as the ISA writes it, limb by limb: one launch (csrc/ntt_product.hip); and written loop by loop with stores
        # in between (the form of rounds 1-2: one batched launch per loop and pass)
        vp.load_isram(0, programs.assemble(programs.poly_mul(n, mods)))
        vp.load_isram(2048, programs.assemble(programs.poly_mul_batched(n, mods, scratch_row=3 * L * rows)))
        fa, fb = oracle.ntt(a[0], mods[0]), oracle.ntt(b[0], mods[0])
        prod = np.array([(int(x) * int(y)) % mods[0] for x, y in zip(fa, fb)], dtype=np.uint64)
        want = oracle.intt(prod, mods[0])
        res = {}
        for name, pc in (("poly_mul_8_limbs_us", 0), ("loop_by_loop_us", 2048)):
            run = lambda: vp.run(pc, src0=0, src1=L * rows, rslt=2 * L * rows)
            assert run() == 0
            assert np.array_equal(vp.read(SPM, 2 * L * rows, n), want)
            l0 = vp.stats()["kernel_launches"]
            res[name] = timeit(run, vp.sync, 250) * 1e6
            res[name.replace("_us", "_launches")] = (vp.stats()["kernel_launches"] - l0) / 251
    t = res["poly_mul_8_limbs_us"] * 1e-6
    res.update({"fused_min_GBps": 3 * 8 * n * L / t / 1e9, "isa_materialised_GBps": 9 * 8 * n * L / t / 1e9})
    return res


def config45(L):
    logn, n = 16, 1 << 16
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    special = avp.nth_prime(logn, L)
    lo = keyswitch.Layout(n, mods, special)
    rng = np.random.default_rng(4)
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
        for jl in range(lo.Mr):                                  # random key, limb by limb (no 600 MB host array)
            q = lo.modulus(jl)
            vp.write(KSK, jl * L * 2 * lo.rows, rng.integers(0, q, L * 2 * n, dtype=np.uint64))
        ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
        ks.load_ct(ct)
        ks.load_ct_b(ct[::-1].copy())
        k = pow(3, 5, 2 * n)
        ks.rotate(k); vp.sync()
        l0 = vp.stats()["kernel_launches"]
        t = timeit(lambda: ks.rotate(k), vp.sync, 5)
        launches_rot = (vp.stats()["kernel_launches"] - l0) / 6
        ks.multiply(); vp.sync()
        l0 = vp.stats()["kernel_launches"]
        tm = timeit(lambda: ks.multiply(), vp.sync, 5)
        launches_mul = (vp.stats()["kernel_launches"] - l0) / 6
        # phase 1 with the automorphism in the NTT domain (keyswitch.py: reference_shape=False): L transforms instead of 3L
        ks2 = keyswitch.ShardedKeySwitch(vp, n, mods, special, reference_shape=False)
        ks2.rotate(k); vp.sync()
        t_nd = timeit(lambda: ks2.rotate(k), vp.sync, 5)
        transforms = 2 * L + L * (L + 1) + L + 2 * (1 + L)          # rotate: iNTT c0,c1; digits x moduli; NTT rotated c0; mod-down
        transforms_mul = L + L * (L + 1) + 2 * (1 + L)              # product: iNTT d2; digits x moduli; mod-down
        # algorithmic bytes of the product: transforms*16N + key 2*L*(L+1)*8N + tensor (4 in, 3 out)*8N*L
        alg = (transforms_mul * 16 + 2 * L * (L + 1) * 8 + 7 * 8 * L) * n
    return {"L": L, "rotate_keyswitch_ms": t * 1e3, "rotate_keyswitch_ms_ntt_domain_phase1": t_nd * 1e3, "transforms_per_rotate": transforms,
            "transform_equiv_per_s": transforms / t, "kernel_launches_per_rotate": launches_rot,
            "hmult_relin_ms": tm * 1e3, "transforms_per_hmult": transforms_mul, "kernel_launches_per_hmult": launches_mul,
            "hmult_algorithmic_GBps": alg / tm / 1e9}


def hybrid45(L, alpha, batch=1):
    """Configs 4/5 with the hybrid key-switch (alpha primes per digit, K = alpha special primes, vautn);
    ``batch`` ciphertexts per call under one key (times are per call)."""
    from aloha_amd import hybrid
    logn, n, K = 16, 1 << 16, alpha
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    specials = [avp.nth_prime(logn, L + k) for k in range(K)]
    lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=batch)
    rng = np.random.default_rng(6)
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, batch=batch)
        for jl in range(lo.Mr):
            vp.write(KSK, jl * lo.dnum * 2 * lo.rows, rng.integers(0, lo.mods_all[jl], lo.dnum * 2 * n, dtype=np.uint64))
        ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
        for b in range(batch):
            hs.load_ct(ct, b); hs.load_ct_b(ct[::-1].copy(), b)
        k = pow(3, 5, 2 * n)
        hs.rotate(k); vp.sync()
        l0 = vp.stats()["kernel_launches"]
        t = timeit(lambda: hs.rotate(k), vp.sync, 5)
        launches_rot = (vp.stats()["kernel_launches"] - l0) / 6
        hs.multiply(); vp.sync()
        tm = timeit(lambda: hs.multiply(), vp.sync, 5)
        transforms = L + (lo.dnum * (L + K) - L) + 2 * K + 2 * L
    return {"L": L, "alpha": alpha, "K": K, "dnum": lo.dnum, "batch": batch, "rotate_keyswitch_ms": t * 1e3, "hmult_relin_ms": tm * 1e3,
            "hmult_relin_ms_per_ciphertext": tm * 1e3 / batch,
            "transforms_per_switch": transforms, "key_limbs": 2 * lo.dnum * (L + K), "kernel_launches_per_rotate": launches_rot}


def aut_batch():
    logn, n, B = 16, 1 << 16, 64
    q = avp.nth_prime(logn, 0)
    rows = n // 128
    lines = [f"vsetvl {64 * n:#x}", f"vsetq {q:#x}"]
    for i in range(B):
        lines += [f"vle v0, src0+{i * 8 * n:#x}", "vaut v2, v0, 0x0", f"vse v2, rslt+{i * 8 * n:#x}"]
    words = isa.assemble("\n".join(lines + ["break"]))
    rng = np.random.default_rng(5)
    a = rng.integers(0, q, B * n, dtype=np.uint64)
    with avp.VP(spm_rows=2 * B * rows, ksk_rows=1, max_log_n=16) as vp:
        vp.write(SPM, 0, a)
        vp.load_isram(0, words)
        run = lambda: vp.run(0, src0=0, rslt=B * rows, step=pow(3, 7, 2 * n))
        assert run() == 0
        assert np.array_equal(vp.read(SPM, B * rows, n), oracle.vaut(a[:n], q, pow(3, 7, 2 * n)))
        t = timeit(run, vp.sync, 20)
    return {"limbs": B, "us_per_limb": t / B * 1e6, "algorithmic_GBps": 16 * n * B / t / 1e9}


if __name__ == "__main__":
    todo = {"config2_ntt_n4096": config2, "small_n_batched": small_n_batched, "config3_polymul_n16384_8limbs": config3,
            "config4_keyswitch_n65536_L24": lambda: config45(24), "config5_keyswitch_n65536_L32": lambda: config45(32),
            "config5_automorphism_batch": aut_batch,
            "hybrid_config4_L24_alpha4": lambda: hybrid45(24, 4), "hybrid_config5_L32_alpha4": lambda: hybrid45(32, 4),
            "hybrid_config4_L24_alpha8": lambda: hybrid45(24, 8),
            "hybrid_config4_L24_alpha4_batch4": lambda: hybrid45(24, 4, 4), "hybrid_config5_L32_alpha4_batch4": lambda: hybrid45(32, 4, 4)}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]           # e.g. `config4` (prefix match), for profiling runs
    res = {k: f() for k, f in todo.items() if not only or any(k.startswith(o) for o in only)}
    print(json.dumps(res, indent=1))
