#!/usr/bin/env python3
"""CPU baseline for EVERY BASELINE.json config: the oracle -- a plain-C port of the reference's
arithmetic and ISA machine (oracle/vp_oracle.c; the reference itself is RTL and no Verilog simulator exists) -- timed
on THIS host, so that each GPU number of tests/perf/bench_configs.py has a CPU number beside it.

    python tests/perf/bench_configs_cpu.py [--threads T] [prefix ...] > gpurun_out/configs_cpu.json

Two figures per config, both `kind: "port"`:
  * one thread: one operation on one oracle machine (the machine retires one instruction at a time, like the
    reference's sequencer, seq_top.v:343-413);
  * all cores: T independent operations on T oracle machines in T threads (the C code runs outside the GIL): the
    throughput a host reaches on a stream of ciphertexts -- limbs of ONE operation cannot be spread over cores
    without changing the machine.
The same programs, layouts and input rules as the GPU script; config 1 is the reference's own CPU-runnable case (the
8-op chain of tv/case0_4_4 through the test-bench flow).  This file lives under tests/: it is measurement
infrastructure and the only thing it runs is the oracle.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle  # noqa: E402
from aloha_amd import harness, hybrid, isa, keyswitch, programs  # noqa: E402

SPM, KSK = 0, 1


def host_cores() -> int:
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def one_and_many(make, op, threads, reps=1):
    """make() -> state (an oracle machine, loaded); op(state) runs ONE operation.  Returns seconds per operation on one
    thread and the all-cores throughput (operations per second with `threads` machines side by side)."""
    st = make()
    op(st)                                                   # warm: twiddle tables, page-in
    t0 = time.perf_counter()
    for _ in range(reps):
        op(st)
    single = (time.perf_counter() - t0) / reps
    states = [st] + [make() for _ in range(threads - 1)]
    for s in states[1:]:
        op(s)
    done = [0] * threads
    budget = max(2.0, 2 * single)

    def worker(i):
        t_end = time.perf_counter() + budget
        while True:
            op(states[i])
            done[i] += 1
            if time.perf_counter() >= t_end:
                break

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    for s in states:
        close = getattr(s, "close", None) or getattr(getattr(s, "vp", None), "close", None)
        if close:
            close()
    return {"one_thread_ms": single * 1e3, "all_cores_ops_per_s": sum(done) / wall, "all_cores_ms_per_op": wall / sum(done) * 1e3,
            "cores": threads, "kind": "port"}


def config1(threads):
    """tv/case0_4_4: load, rotate + key-switch, 2 encodes, 2 mul_plain, hom_add, store (the RTL's 8-op chain, N = 8192)
    through the test-bench flow on the oracle machine -- the reference's own CPU-runnable case."""
    from test_oracle_golden import check_case0, run_case0
    from util import reference_images

    def make():
        vp = oracle.OracleVP()
        tb = harness.TestBench(vp)
        tb.load_programs(reference_images())
        return (vp, tb)

    def op(s):
        check_case0(run_case0(*s))

    r = one_and_many(make, op, threads, reps=3)
    r["what"] = "8-op RTL chain of tv/case0_4_4 incl. the check of all 7 dumps"
    return r


def config2(threads):
    logn, n = 12, 1 << 12
    q = oracle.nth_prime(logn, 0)
    a = np.random.default_rng(2).integers(0, q, (4096, n), dtype=np.uint64)
    oracle.set_threads(1)
    oracle.ntt(a[:64], q)
    t0 = time.perf_counter(); oracle.ntt(a[:1], q); lat = time.perf_counter() - t0
    t0 = time.perf_counter(); oracle.ntt(a, q); one = 4096 / (time.perf_counter() - t0)
    oracle.set_threads(threads)
    oracle.ntt(a, q)
    t0 = time.perf_counter(); oracle.ntt(a, q); many = 4096 / (time.perf_counter() - t0)
    oracle.set_threads(1)
    return {"single_ntt_latency_us": lat * 1e6, "one_thread_ntt_per_s": one, "all_cores_ntt_per_s": many, "cores": threads, "kind": "port"}


def config3(threads):
    logn, n, L = 14, 1 << 14, 8
    rows = n // 128
    mods = [oracle.nth_prime(logn, i) for i in range(L)]
    rng = np.random.default_rng(3)
    a = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods])
    b = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods])
    words = programs.assemble(programs.poly_mul_batched(n, mods, scratch_row=3 * L * rows))

    def make():
        vp = oracle.OracleVP(spm_rows=4 * L * rows, ksk_rows=1)
        vp.write(SPM, 0, a.reshape(-1)); vp.write(SPM, L * rows, b.reshape(-1))
        vp.load_isram(0, words)
        return vp

    r = one_and_many(make, lambda vp: vp.run(0, src0=0, src1=L * rows, rslt=2 * L * rows), threads, reps=5)
    r["what"] = "8-limb polynomial product N=2^14 (NTT, pointwise, iNTT) as an HE-ISA program"
    return r


def reference_ks(L, op_name, threads):
    """Configs 4 / 5 with the reference's decomposition (one digit per prime, one special prime: keyswitch.py)."""
    n = 1 << 16
    mods = [oracle.nth_prime(16, i) for i in range(L)]
    special = oracle.nth_prime(16, L)
    lo = keyswitch.Layout(n, mods, special)
    rng = np.random.default_rng(4)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    k = pow(3, 5, 2 * n)

    def make():
        vp = oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows)
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
        for jl in range(lo.Mr):
            vp.write(KSK, jl * L * 2 * lo.rows, np.random.default_rng(900 + jl).integers(0, lo.modulus(jl), L * 2 * n, dtype=np.uint64))
        ks.load_ct(ct); ks.load_ct_b(ct[::-1].copy())
        return ks

    r = one_and_many(make, (lambda ks: ks.multiply()) if op_name == "multiply" else (lambda ks: ks.rotate(k)), threads)
    r["what"] = f"{op_name} + key-switch, N=2^16, L={L}+1, one digit per prime (the reference's algorithm)"
    return r


def hybrid_ks(L, alpha, op_name, threads):
    n, K = 1 << 16, alpha
    mods = [oracle.nth_prime(16, i) for i in range(L)]
    specials = [oracle.nth_prime(16, L + k) for k in range(K)]
    lo = hybrid.HybridLayout(n, mods, specials, alpha)
    rng = np.random.default_rng(6)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    k = pow(3, 5, 2 * n)

    def make():
        vp = oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows)
        hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha)
        for jl in range(lo.Mr):
            vp.write(KSK, jl * lo.dnum * 2 * lo.rows, np.random.default_rng(600 + jl).integers(0, lo.mods_all[jl], lo.dnum * 2 * n, dtype=np.uint64))
        hs.load_ct(ct); hs.load_ct_b(ct[::-1].copy())
        return hs

    r = one_and_many(make, (lambda hs: hs.multiply()) if op_name == "multiply" else (lambda hs: hs.rotate(k)), threads)
    r["what"] = f"{op_name} + hybrid key-switch, N=2^16, L={L}+{K}, alpha={alpha}"
    return r


def aut_batch(threads):
    logn, n, B = 16, 1 << 16, 64
    q = oracle.nth_prime(logn, 0)
    rows = n // 128
    lines = [f"vsetvl {64 * n:#x}", f"vsetq {q:#x}"]
    for i in range(B):
        lines += [f"vle v0, src0+{i * 8 * n:#x}", "vaut v2, v0, 0x0", f"vse v2, rslt+{i * 8 * n:#x}"]
    words = isa.assemble("\n".join(lines + ["break"]))
    a = np.random.default_rng(5).integers(0, q, B * n, dtype=np.uint64)

    def make():
        vp = oracle.OracleVP(spm_rows=2 * B * rows, ksk_rows=1)
        vp.write(SPM, 0, a)
        vp.load_isram(0, words)
        return vp

    r = one_and_many(make, lambda vp: vp.run(0, src0=0, rslt=B * rows, step=pow(3, 7, 2 * n)), threads, reps=3)
    r["what"] = "automorphism of 64 limbs, N=2^16 (vle; vaut; vse per limb)"
    r["one_thread_algorithmic_GBps"] = 16 * n * B / (r["one_thread_ms"] * 1e-3) / 1e9
    return r


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("only", nargs="*")
    args = ap.parse_args()
    T = args.threads or host_cores()
    todo = {"config1_tv_case0_chain": config1, "config2_ntt_n4096": config2, "config3_polymul_n16384_8limbs": config3,
            "config4_keyswitch_n65536_L24": lambda t: reference_ks(24, "multiply", t),
            "config5_keyswitch_n65536_L32": lambda t: reference_ks(32, "rotate", t),
            "config5_automorphism_batch": aut_batch,
            "hybrid_config4_L24_alpha4": lambda t: hybrid_ks(24, 4, "multiply", t),
            "hybrid_config5_L32_alpha4": lambda t: hybrid_ks(32, 4, "rotate", t)}
    res = {"host_cores_used": T}
    for k, f in todo.items():
        if args.only and not any(k.startswith(o) for o in args.only):
            continue
        res[k] = f(T)
        print(f"{k}: {json.dumps(res[k])}", file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))
