#!/usr/bin/env python3
"""bench.py -- headline benchmark: forward negacyclic NTT/s at N = 2^16, 60-bit primes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--primes L] [--polys P]

With --gpus N > 1 and no launcher (WORLD_SIZE unset) the script starts its N ranks itself, one per GPU, before it
touches the GPU; under torch.distributed.run it uses the ranks it is given.  After W warm-up steps an untimed settle
loop runs until the step time is stable (`settle_ms` in the output: the clocks of a freshly loaded GPU need ~100 ms),
then exactly K steps are timed.  For N > 1 a second object `keyswitch` reports BASELINE config 4 sharded over the ranks.

One "step" = one pass of the hot path (vntt, fused over the batch: aloha_vp_ntt_rows) over a
batch of L x P limbs of N = 2^16 coefficients resident in HBM, in place.  The default batch
(32 primes x 128 limbs = 4096 limbs = 2 GiB) is eight times the 256 MiB Infinity Cache, so every
step streams from HBM, and long enough (1.6 ms) that the launch tails of the two kernels are
under 2 % (1024 limbs: 2.40 M NTT/s, 4096: 2.53 M; SURVEY 8d asks for "B >= 256, report B").  For N > 1 the driver launches one rank per GPU (torch.distributed,
backend nccl = RCCL); limbs are independent so ranks share nothing on the data path (weak
scaling: every rank transforms its own batch), only the barrier and the max-over-ranks of the
elapsed time go through the collective.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes (16*N per transform) / measured device time of
                  the transform's kernels, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (oracle/vp_oracle.c, a scalar port of the reference's
                  arithmetic; the reference itself is RTL and cannot run here) timed on this
                  host for a bounded sample of the same transforms.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

LOGN = 16
N = 1 << LOGN
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_SUMMARY = "r04_pmc_summary.json"     # written by tools/collect_pmc.sh; its "commit" field says for which kernels


def pmc_summary():
    """The committed rocprofv3 PMC summary of this same command (separate --pmc passes, tools/collect_pmc.sh), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_SUMMARY)) as f:
            return json.load(f)
    except Exception:
        return {}


def measured_traffic_per_step(limbs: int):
    """HBM bytes per step from the PMC summary (FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE), scaled
    to the batch size; None when no summary is present."""
    try:
        d = pmc_summary()["derived"]
        return d["hbm_bytes_per_step_corrected"] * limbs / (d["algorithmic_bytes_per_step"] / (16 * N))
    except Exception:
        return None


def valu_insts_per_ntt():
    """VALU lane-instructions per transform from the PMC summary: SQ_INSTS_VALU counts wave64 instructions per launch
    of the 4096-limb batch, both kernels; None when no summary is present."""
    try:
        d = pmc_summary()
        waves = d["ntt_col_kernel.SQ_INSTS_VALU"] + d["ntt_row_kernel.SQ_INSTS_VALU"]
        return waves * 64.0 / (d["derived"]["algorithmic_bytes_per_step"] / (16 * N))
    except Exception:
        return None


def self_launch(argv, gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this same script, one per GPU, BEFORE anything in
    this process touches the GPU (the parent only waits), and return the worst exit code.  The driver's own launcher
    (torch.distributed.run) sets WORLD_SIZE, in which case this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # The ranks are polled together: the first one that fails -- a non-zero exit or a death by signal (negative code: a GPU
    # fault ends a rank with SIGABRT / SIGSEGV) -- ends the run; its siblings, which would otherwise sit in the rendezvous
    # or in a collective until a timeout, are terminated, and the failure is what this process returns.
    worst, failed_at = 0, None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0 and worst == 0:
                worst = rc if rc > 0 else 128 - rc            # 128 + signal number, as a shell reports it
                failed_at = time.monotonic()
        if not procs:
            break
        time.sleep(0.05)
        if failed_at is not None and time.monotonic() - failed_at > 3.0:   # (siblings that fail for the same reason say so themselves first)
            for o in procs:
                o.terminate()
            deadline = time.monotonic() + 10.0                  # a rank that ignores SIGTERM for 10 s is killed
            while procs and time.monotonic() < deadline:
                procs = [o for o in procs if o.poll() is None]
                time.sleep(0.05)
            for o in procs:
                o.kill()
                o.wait()
            procs = []
    return worst



def cpu_model() -> str:
    """The host CPU's model name (SURVEY 8d / BASELINE.md 3 ask for core count AND model beside the CPU figures)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(seconds: float = 8.0, check=None) -> dict:
    """The CPU path beside the GPU number: the oracle -- a scalar C port of the reference's
    arithmetic; the reference itself is RTL and no Verilog simulator exists (probed below) -- timed on
    this host on a bounded sample of the same transforms, on ONE thread and on ALL host cores (OpenMP
    over independent limbs).  `value` is the all-cores rate and `cores` the threads it used.
    ``check = (input limb, prime, the GPU's output)``: the same CPU transform also verifies the first limb
    of the GPU batch word for word (this leg is the only place bench.py touches oracle/)."""
    import shutil
    import oracle
    agrees = None
    if check is not None:
        agrees = bool(np.array_equal(oracle.ntt(check[0], check[1]), check[2]))
    q = oracle.nth_prime(LOGN, 0)
    rng = np.random.default_rng(1)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                     # a container's CPU quota (cgroup v2), when it is tighter
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    a = rng.integers(0, q, size=(max(16, 2 * cores), N), dtype=np.uint64)

    def rate(threads, batch, budget):
        oracle.set_threads(threads)
        oracle.ntt(a[:batch], q)                             # warm (library load, page-in, thread pool)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            oracle.ntt(a[:batch], q)
            done += batch
        return done / (time.perf_counter() - t0), done

    single, n1 = rate(1, 16, seconds / 2)
    multi, nm = rate(cores, a.shape[0], seconds / 2)
    oracle.set_threads(1)
    sims = [t for t in ("xsim", "vivado", "iverilog", "verilator", "vcs", "xrun") if shutil.which(t)]
    return {"value": multi, "unit": "NTT/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "single_thread": single, "gpu_first_limb_equals_cpu": agrees,
            "sample": f"{nm} forward NTTs on {cores} threads + {n1} on one thread, N=2^16, one 60-bit prime, "
                      f"{seconds:.0f} s in all; Verilog simulators on this host: {sims or 'none'} "
                      "(the reference's own test bench cannot be run)",
            "reference_rtl_derived": {"note": "DERIVED, not measured: the RTL's cycle model at the test bench's 100 MHz "
                                              "clock, N=8192 only (ntt_fsm.sv:68, vp_defines.vh:79-81, top_noaxilite_tb.sv:232-235)",
                                      "cycles_per_ntt_n8192": 850, "ntt_per_s_n8192": 1e8 / 850}}


# Design ceilings of the headline transform, each MEASURED on this chip (frac of the 16*N HBM roofline):
CEILINGS = {
    "two_pass_memory": {"frac": 0.375, "source": "profiles/r01_c_ubench_mem.txt",
                        "what": "read+write streams with the two passes' access patterns (6.05-6.1 TB/s each): 357 us per 1024 limbs -- a 512 KiB limb "
                                "does not fit a CU's 160 KiB LDS, so the transform is two HBM round trips whatever the arithmetic costs"},
    "single_round_trip_skeleton": {"frac": 0.447, "source": "profiles/r02_c_ubench_icfuse.txt",
                                   "what": "memory skeleton of ONE launch handing the intermediate over through the Infinity Cache: 1200 us per 4096 "
                                           "limbs; as real kernels 1.88-2.13 ms (variant build, csrc/experiments)"},
}


def configs_leg(device: int, budget_s: float = 85.0) -> dict:
    """BASELINE.json configs 2-5 on this GPU, bounded (the default bench run stays within minutes): per config the time, the
    ALGORITHMIC bytes by SURVEY.md 8d's formulas (aloha_amd/accounting.py; the itemised sum is printed), the fraction of the
    8 TB/s roofline, `verified` (every output word against the CPU oracle's ISA machine, in the untimed part) and the CPU
    port's time for the same operation (that verification run, one thread).  Config 1 is the reference's CPU-runnable case
    (parity tests only).  Key-switch = the hybrid form the product recommends (alpha = K = 4)."""
    import oracle
    from aloha_amd import accounting as acc
    from aloha_amd import hybrid, isa, programs
    from aloha_amd import vp as avp
    t_start = time.perf_counter()
    out = {"note": "one GPU, inputs resident in HBM; every time after an untimed >= 100 ms of the same operation back to back (settled clocks, as the "
                   "headline); frac = algorithmic bytes / time / 8 TB/s; cpu_port_ms = the oracle (kind: port), one thread, same operation"}
    SPM, KSK = 0, 1

    def timeit(fn, sync, reps):
        # median of up to five blocks of back-to-back calls: a process that overdraws the container's CPU quota is paused by the kernel
        # for 20-80 ms (seen as GAPS between launches in rocprofv3 traces, never as long kernels; the cause found in round 3 -- the
        # oracle's idle OpenMP workers spinning -- is fixed, profiles/r03_z_launch_gaps.txt), and one such pause inside a single
        # long loop is the whole figure for a 15 us operation
        # ... and every figure is taken at SETTLED clocks, like the headline: an MI355X that idled while the host verified the previous
        # config runs its first ~100 ms of launches up to 9 % slow (profiles/r04_configs_clock_settle.txt: HMult L = 24 0.270 ms over 30
        # calls from idle, 0.250 over 400 and over 1500), and none of these loops is 100 ms long
        settle(fn, sync, min_ms=100.0, max_ms=400.0)
        nblk = 5 if reps >= 50 else 3 if reps >= 15 else 1          # (one synchronisation per block: ~20 us spread over its calls)
        per = max(1, reps // nblk)
        ts = []
        for _ in range(nblk):
            t0 = time.perf_counter()
            for _ in range(per):
                fn()
            sync()
            ts.append((time.perf_counter() - t0) / per)
        return sorted(ts)[len(ts) // 2]

    # -- config 2: one forward NTT at N = 2^12 (latency), and batched -----------------------------------------------------
    logn, n = 12, 1 << 12
    q = avp.nth_prime(logn, 0)
    rng = np.random.default_rng(2)
    with avp.VP(spm_rows=4096 * n // 128, ksk_rows=1, max_log_n=12, device=device) as vp:
        a = rng.integers(0, q, n, dtype=np.uint64)
        vp.write(SPM, 0, a)
        vp.ntt_rows(SPM, n // 128, 0, logn, [q])
        t0 = time.perf_counter(); exp = oracle.ntt(a, q); cpu_ms = (time.perf_counter() - t0) * 1e3
        ok = bool(np.array_equal(vp.read(SPM, n // 128, n), exp))
        t1 = timeit(lambda: (vp.ntt_rows(SPM, 0, 0, logn, [q]), vp.sync()), vp.sync, 200)
        B = 4096
        tb = timeit(lambda: vp.ntt_rows(SPM, 0, 0, logn, [q] * B), vp.sync, 250)   # (blocks of ~5 ms: shorter ones sit in the clock ramp)
        out["config2_ntt_n4096"] = {"single_call_us": t1 * 1e6, "batched_ntt_per_s": B / tb, "algorithmic_bytes": acc.ntt_bytes(n, B),
                                    "batched_GBps": acc.ntt_bytes(n, B) / tb / 1e9, "frac": acc.ntt_bytes(n, B) / tb / 1e9 / HBM_PEAK_GBS,
                                    "verified": ok, "cpu_port_ms": cpu_ms}
    # ... and the reference's own ring size, N = 2^13 (every tv/ vector, every shipped program): batched forward transforms
    logn, n, B = 13, 1 << 13, 4096
    q = avp.nth_prime(logn, 0)
    with avp.VP(spm_rows=B * n // 128, ksk_rows=1, max_log_n=13, device=device) as vp:
        vp.fill_synthetic(SPM, 0, logn, [q] * 2, seed_base=0xA10A + 200000)
        x = vp.read(SPM, 0, n)
        vp.ntt_rows(SPM, n // 128, 0, logn, [q])
        ok = bool(np.array_equal(vp.read(SPM, n // 128, n), oracle.ntt(x, q)))
        tb = timeit(lambda: vp.ntt_rows(SPM, 0, 0, logn, [q] * B), vp.sync, 150)
        out["reference_size_ntt_n8192"] = {"batched_ntt_per_s": B / tb, "algorithmic_bytes": acc.ntt_bytes(n, B), "batched_GBps": acc.ntt_bytes(n, B) / tb / 1e9,
                                            "frac": acc.ntt_bytes(n, B) / tb / 1e9 / HBM_PEAK_GBS, "verified": ok,
                                            "reference_rtl_derived_ntt_per_s": 1e8 / 850,
                                            "note": "the RTL's figure is DERIVED, not measured: 832 + 18 cycles per transform (ntt_fsm.sv:68, vp_defines.vh:79-81) at the test bench's 100 MHz"}
    # -- config 3: 8-limb polynomial product at N = 2^14 ------------------------------------------------------------------
    logn, n, L = 14, 1 << 14, 8
    rows = n // 128
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    rng = np.random.default_rng(3)
    a = np.stack([rng.integers(0, m, n, dtype=np.uint64) for m in mods])
    b = np.stack([rng.integers(0, m, n, dtype=np.uint64) for m in mods])
    # the product as the ISA writes it (mul_plain.mem:4-19's shape with the transforms added): the sequencer hands every limb's
    # `vle; vntt; vle; vntt; vfqmul.vv; vintt; vse` to ONE kernel (csrc/ntt_product.hip); `loop_by_loop_us`: the same product
    # written as four loops with stores in between (one batched launch per loop and pass, the form of rounds 1-2)
    words = programs.assemble(programs.poly_mul(n, mods))
    words_loops = programs.assemble(programs.poly_mul_batched(n, mods, scratch_row=3 * L * rows))
    with avp.VP(spm_rows=4 * L * rows, ksk_rows=1, max_log_n=14, device=device) as vp:
        o = oracle.OracleVP(spm_rows=4 * L * rows, ksk_rows=1)
        for e in (vp, o):
            e.write(SPM, 0, a.reshape(-1)); e.write(SPM, L * rows, b.reshape(-1)); e.load_isram(0, words)
        vp.load_isram(2048, words_loops)
        run = lambda: vp.run(0, src0=0, src1=L * rows, rslt=2 * L * rows)
        assert run() == 0
        t0 = time.perf_counter(); assert o.run(0, src0=0, src1=L * rows, rslt=2 * L * rows) == 0; cpu_ms = (time.perf_counter() - t0) * 1e3
        want = o.read(SPM, 2 * L * rows, L * n)
        ok = bool(np.array_equal(vp.read(SPM, 2 * L * rows, L * n), want))
        l0 = vp.stats()["kernel_launches"]
        assert run() == 0
        launches = float(vp.stats()["kernel_launches"] - l0)
        t = timeit(run, vp.sync, 250)
        run_loops = lambda: vp.run(2048, src0=0, src1=L * rows, rslt=2 * L * rows)
        assert run_loops() == 0
        ok = ok and bool(np.array_equal(vp.read(SPM, 2 * L * rows, L * n), want))
        t_loops = timeit(run_loops, vp.sync, 250)
        by = acc.polymul_bytes(n, L)
        out["config3_polymul_n16384_8limbs"] = {"us": t * 1e6, "launches": launches, "algorithmic_bytes": by,
                                                "fused_min_GBps": by["fused_minimum"] / t / 1e9, "frac": by["fused_minimum"] / t / 1e9 / HBM_PEAK_GBS,
                                                "loop_by_loop_us": t_loops * 1e6, "verified": ok, "cpu_port_ms": cpu_ms}
        o.close()
    # ... and 64 limbs (8 such products) per run: the same kernel with the chip half full
    Lb = 64
    mods_b = [avp.nth_prime(logn, i % L) for i in range(Lb)]
    with avp.VP(spm_rows=3 * Lb * rows, ksk_rows=1, max_log_n=14, device=device) as vp:
        vp.fill_synthetic(SPM, 0, logn, mods_b + mods_b, seed_base=0xA10A + 300000)
        vp.load_isram(0, programs.assemble(programs.poly_mul(n, mods_b)))
        run = lambda: vp.run(0, src0=0, src1=Lb * rows, rslt=2 * Lb * rows)
        assert run() == 0
        # all 64 limbs against the oracle's ISA machine running the same program (limb 0 alone says nothing about the
        # hand-offs of the later limb slots or the XCD interleave at full grid)
        o = oracle.OracleVP(spm_rows=3 * Lb * rows, ksk_rows=1)
        o.write(SPM, 0, vp.read(SPM, 0, 2 * Lb * n)); o.load_isram(0, programs.assemble(programs.poly_mul(n, mods_b)))
        assert o.run(0, src0=0, src1=Lb * rows, rslt=2 * Lb * rows) == 0
        okb = bool(np.array_equal(vp.read(SPM, 2 * Lb * rows, Lb * n), o.read(SPM, 2 * Lb * rows, Lb * n)))
        o.close()
        tb = timeit(run, vp.sync, 250)
        byb = acc.polymul_bytes(n, Lb)
        out["config3_polymul_n16384_8limbs"]["batch_of_8_products"] = {
            "us_per_product": tb * 1e6 / 8, "limbs": Lb, "fused_min_GBps": byb["fused_minimum"] / tb / 1e9,
            "frac": byb["fused_minimum"] / tb / 1e9 / HBM_PEAK_GBS, "verified": okb}

    # -- configs 4 / 5: HMult + relinearisation at L = 24, rotate + key-switch at L = 32 (N = 2^16, hybrid alpha = K = 4) -------
    def keyswitch(L, op, alpha=4):
        logn, n = 16, 1 << 16
        mods = [avp.nth_prime(logn, i) for i in range(L)]
        specials = [avp.nth_prime(logn, L + k) for k in range(alpha)]
        lo = hybrid.HybridLayout(n, mods, specials, alpha)
        # (instruction ROM of 16384 words on both machines: the mid phase of L = 32 needs ~6700, the reference's 4096 would cut it in two)
        sizes = dict(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, iram_depth=16384)
        k = pow(3, 5, 2 * n)
        with avp.VP(max_log_n=16, device=device, **sizes) as vp:
            hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha)
            # synthetic key and ciphertexts straight into HBM (SURVEY 8d's splitmix64 rule), copied out once for the oracle
            vp.fill_synthetic(KSK, 0, logn, [m for m in lo.mods_all for _ in range(lo.dnum * 2)], seed_base=0xA10A + 7000)
            for base, seed in ((lo.ct_in, 1), (lo.ct_b, 2)):
                vp.fill_synthetic(SPM, lo.row(base), logn, [mods[j] for _ in range(2) for j in range(L)], seed_base=0xA10A + 100000 * seed)
            res = {}
            verify = time.perf_counter() - t_start < budget_s - (12.0 if L > 24 else 8.0)
            if verify:
                o = oracle.OracleVP(**sizes)
                ho = hybrid.HybridKeySwitch(o, n, mods, specials, alpha)
                o.write(KSK, 0, vp.read(KSK, 0, lo.ksk_limbs * n))
                for base in (lo.ct_in, lo.ct_b):
                    o.write(SPM, lo.row(base), vp.read(SPM, lo.row(base), 2 * L * n))
                t0 = time.perf_counter()
                ho.multiply() if op == "multiply" else ho.rotate(k)
                res["cpu_port_ms"] = (time.perf_counter() - t0) * 1e3
            run = (lambda: hs.multiply()) if op == "multiply" else (lambda: hs.rotate(k))
            run(); vp.sync()
            if verify:
                res["verified"] = bool(np.array_equal(hs.read_out(), ho.read_out()))
                o.close()
            else:
                res["verified"] = None
            l0 = vp.stats()["kernel_launches"]
            run()
            launches = float(vp.stats()["kernel_launches"] - l0)
            t = timeit(run, vp.sync, 45)
            by = acc.keyswitch_bytes(n, L, alpha, lo.dnum, op)
            res.update({"ms": t * 1e3, "launches": launches, "L": L, "K": alpha, "alpha": alpha, "dnum": lo.dnum,
                        "iram_depth": 16384,
                        "algorithmic_bytes": by, "GBps": by["survey_total"] / t / 1e9, "frac": by["survey_total"] / t / 1e9 / HBM_PEAK_GBS,
                        "frac_lean": by["lean_total"] / t / 1e9 / HBM_PEAK_GBS})
            return res

    out["config4_hmult_relin_n65536_L24"] = keyswitch(24, "multiply")
    out["config5_rotate_keyswitch_n65536_L32"] = keyswitch(32, "rotate")
    # the digit size is the implementation's choice (BASELINE names the limbs, not the decomposition): alpha = K = 8 -- 3 / 4 digits --
    # halves the lifted limbs and the key against alpha = 4 at the price of twice the special primes
    out["config4_hmult_relin_n65536_L24_alpha8"] = keyswitch(24, "multiply", 8)
    out["config5_rotate_keyswitch_n65536_L32_alpha8"] = keyswitch(32, "rotate", 8)

    # -- configs 4 / 5 on the decomposition SURVEY 8d DEFINES them on: one digit per ciphertext prime and ONE special prime -- the
    #    reference's own algorithm (keyswitch.mem:4-43 digits, :46-79 inner products, :80-121 mod-down), generated for L limbs by
    #    aloha_amd/keyswitch.py in the reference's 4096-word ROM; dnum = L, K = 1: L (L + 1) digit transforms, 2 L (L + 1) key limbs
    def keyswitch_reference(L, op):
        from aloha_amd import keyswitch as refks
        logn, n = 16, 1 << 16
        mods = [avp.nth_prime(logn, i) for i in range(L)]
        special = avp.nth_prime(logn, L)
        lo = refks.Layout(n, mods, special)
        sizes = dict(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows)
        k = pow(3, 5, 2 * n)
        with avp.VP(max_log_n=16, device=device, **sizes) as vp:
            ks = refks.ShardedKeySwitch(vp, n, mods, special)
            vp.fill_synthetic(KSK, 0, logn, [lo.modulus(j) for j in range(L + 1) for _ in range(2 * L)], seed_base=0xA10A + 17000)
            for base, seed in ((lo.ct_in, 3), (lo.ct_b, 4)):
                vp.fill_synthetic(SPM, lo.row(base), logn, [mods[j] for _ in range(2) for j in range(L)], seed_base=0xA10A + 100000 * seed)
            res = {}
            # the CPU port needs ~4 s (L = 24) / ~8 s (L = 32) for this operation: verified here while the leg's budget lasts, and at full
            # size in tests/test_keyswitch_sharded.py (GPU == the oracle's ISA machine) either way
            verify = time.perf_counter() - t_start < budget_s - (14.0 if L > 24 else 9.0)
            if verify:
                o = oracle.OracleVP(**sizes)
                ko = refks.ShardedKeySwitch(o, n, mods, special)
                for jl in range(lo.Mr):                              # key limbs of one modulus at a time (the whole key is 0.6-1 GiB)
                    o.write(KSK, jl * L * 2 * lo.rows, vp.read(KSK, jl * L * 2 * lo.rows, L * 2 * n))
                for base in (lo.ct_in, lo.ct_b):
                    o.write(SPM, lo.row(base), vp.read(SPM, lo.row(base), 2 * L * n))
                t0 = time.perf_counter()
                ko.multiply() if op == "multiply" else ko.rotate(k)
                res["cpu_port_ms"] = (time.perf_counter() - t0) * 1e3
            run = (lambda: ks.multiply()) if op == "multiply" else (lambda: ks.rotate(k))
            run(); vp.sync()
            if verify:
                res["verified"] = bool(np.array_equal(ks.read_out(), ko.read_out()))
                o.close()
            else:
                res["verified"] = None
            l0 = vp.stats()["kernel_launches"]
            run()
            launches = float(vp.stats()["kernel_launches"] - l0)
            t = timeit(run, vp.sync, 15)
            by = acc.keyswitch_bytes(n, L, 1, L, op)
            res.update({"ms": t * 1e3, "launches": launches, "L": L, "K": 1, "alpha": 1, "dnum": L, "iram_depth": 4096,
                        "decomposition": "the reference's: one digit per prime, one special prime (keyswitch.mem:4-121 for L limbs)",
                        "algorithmic_bytes": by, "GBps": by["survey_total"] / t / 1e9, "frac": by["survey_total"] / t / 1e9 / HBM_PEAK_GBS,
                        "frac_lean": by["lean_total"] / t / 1e9 / HBM_PEAK_GBS})
            return res

    out["config4_hmult_relin_n65536_L24_reference_decomposition"] = keyswitch_reference(24, "multiply")
    out["config5_rotate_keyswitch_n65536_L32_reference_decomposition"] = keyswitch_reference(32, "rotate")

    # -- config 5 as a BATCH of rotations: R rotations of one ciphertext share its decomposition (hoisting) and run in ONE pass of
    #    the phases (hybrid.rotate_hoisted_many: the batch slots of the layout serve as rotation slots, one key region each) ----------
    def rotation_batch(L, R):
        logn, n, alpha = 16, 1 << 16, 4
        mods = [avp.nth_prime(logn, i) for i in range(L)]
        specials = [avp.nth_prime(logn, L + k) for k in range(alpha)]
        lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=R)
        steps = [pow(3, r + 1, 2 * n) for r in range(R)]
        with avp.VP(max_log_n=16, device=device, spm_rows=lo.spm_rows_needed, ksk_rows=R * lo.ksk_limbs * lo.rows, iram_depth=16384) as vp:
            hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, batch=R)
            # (any words are a valid key for timing and parity: the regions hold what load_key_hoisted would have stored)
            vp.fill_synthetic(KSK, 0, logn, [m for _ in range(R) for m in lo.mods_all for _ in range(lo.dnum * 2)], seed_base=0xA10A + 9000)
            for c in range(2):
                for i in range(L):
                    vp.fill_synthetic(SPM, lo.row(lo.s_in(c, i, 0)), logn, [mods[i]], seed_base=0xA10A + 800000 + c * L + i)
            run = lambda: (hs.hoist_one(), hs.rotate_hoisted_many(steps))
            run(); vp.sync()
            res = {}
            verify = time.perf_counter() - t_start < budget_s - 14.0
            if verify:                                           # rotation slot 0 against the oracle machine: the same decomposition, one step
                sizes = dict(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, iram_depth=16384)
                o = oracle.OracleVP(**sizes)
                ho = hybrid.HybridKeySwitch(o, n, mods, specials, alpha, batch=R)
                o.write(KSK, 0, vp.read(KSK, 0, lo.ksk_limbs * n))
                for c in range(2):
                    for i in range(L):
                        o.write(SPM, lo.row(lo.s_in(c, i, 0)), vp.read(SPM, lo.row(lo.s_in(c, i, 0)), n))
                t0 = time.perf_counter()
                ho.hoist_one(); ho.rotate_hoisted_many(steps[:1])
                res["cpu_port_ms_one_rotation"] = (time.perf_counter() - t0) * 1e3
                res["verified"] = bool(np.array_equal(hs.read_out(0), ho.read_out(0)))
                o.close()
            else:
                res["verified"] = None
            l0 = vp.stats()["kernel_launches"]
            run()
            launches = float(vp.stats()["kernel_launches"] - l0)
            t = timeit(run, vp.sync, 15)
            res.update({"rotations": R, "ms_per_rotation": t * 1e3 / R, "ms": t * 1e3, "launches": launches, "L": L, "alpha": alpha, "dnum": lo.dnum})
            return res

    out["config5_rotation_batch_hoisted_n65536_L32"] = rotation_batch(32, 8)
    # -- config 5: the automorphism alone (vaut, 64 limbs) ---------------------------------------------------------------------
    logn, n, B = 16, 1 << 16, 64
    q = avp.nth_prime(logn, 0)
    rows = n // 128
    lines = [f"vsetvl {64 * n:#x}", f"vsetq {q:#x}"]
    for i in range(B):
        lines += [f"vle v0, src0+{i * 8 * n:#x}", "vaut v2, v0, 0x0", f"vse v2, rslt+{i * 8 * n:#x}"]
    with avp.VP(spm_rows=2 * B * rows, ksk_rows=1, max_log_n=16, device=device) as vp:
        vp.fill_synthetic(SPM, 0, logn, [q] * B, seed_base=0xA10A + 500000)
        vp.load_isram(0, isa.assemble("\n".join(lines + ["break"])))
        kk = pow(3, 7, 2 * n)
        run = lambda: vp.run(0, src0=0, rslt=B * rows, step=kk)
        assert run() == 0
        t0 = time.perf_counter(); exp = oracle.vaut(vp.read(SPM, 0, n), q, kk); cpu_ms = (time.perf_counter() - t0) * 1e3
        ok = bool(np.array_equal(vp.read(SPM, B * rows, n), exp))
        t = timeit(run, vp.sync, 150)
        out["config5_automorphism_64limbs"] = {"us_per_limb": t / B * 1e6, "algorithmic_bytes": 16 * n * B, "GBps": 16 * n * B / t / 1e9,
                                               "frac": 16 * n * B / t / 1e9 / HBM_PEAK_GBS, "verified": ok, "cpu_port_ms_per_limb": cpu_ms}
    out["reference_programs_n8192"] = reference_programs(device, timeit)
    out["seconds"] = time.perf_counter() - t_start
    return out


def reference_programs(device: int, timeit) -> dict:
    """The reference's OWN workload: its four instruction-ROM images (sim/vp/isram_file_generator/{encode_post,mul_plain,hom_add,
    keyswitch}.mem at PC 0 / 64 / 160 / 256, loaded unchanged), N = 8192, two ciphertext primes + one special prime, on the golden
    vectors of tv/case0_4_4.  The 8-op chain runs once and every dump is compared with the reference's RTL dumps (tests/golden,
    fixtures made from tv/ by tests/golden/make_fixtures.py) -- golden data, not the oracle; then each program is timed by
    itself, back to back, writing to a scratch ciphertext.  Beside each: the RTL's time DERIVED from its cycle model at the test
    bench's 100 MHz (not a measurement) and the CPU port (the oracle's ISA machine, one thread)."""
    import json as _json
    import oracle
    from aloha_amd import harness
    from aloha_amd import vp as avp
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    man = _json.load(open(os.path.join(gdir, "manifest.json")))
    ld = lambda e: np.fromfile(os.path.join(gdir, e["file"]), dtype="<u8")
    images = {k: np.fromfile(os.path.join(gdir, man["isram"][k]["file"]), dtype="<u4").reshape(-1, 3) for k in ("encode_post", "mul_plain", "hom_add", "keyswitch")}
    m = man["case0"]
    H = harness
    ops = [H.Op(H.OP_LOAD, 0x000, dram_addr=0), H.Op(H.OP_ROTATE, 0x100, src1=0x000, step=2),
           H.Op(H.OP_ENCODE, 0x200, dram_addr=0x0), H.Op(H.OP_ENCODE, 0x280, dram_addr=0x40000),
           H.Op(H.OP_MUL_PLAIN, 0x300, src1=0x000, src2=0x200), H.Op(H.OP_MUL_PLAIN, 0x400, src1=0x100, src2=0x280),
           H.Op(H.OP_HOM_ADD, 0x500, src1=0x300, src2=0x400), H.Op(H.OP_STORE, 0x500, dram_addr=0x1000)]
    n, kstep = 8192, pow(3, 2, 2 * 8192)
    # (program, PC, src0, src1, step): the chain's own operands; every timed run writes the scratch ciphertext at row 0x600
    progs = [("encode_post", 0, 0x200, 0, 0), ("mul_plain", 64, 0x000, 0x200, 0), ("hom_add", 160, 0x300, 0x400, 0), ("keyswitch_rotate", 256, 0x000, 0, kstep)]
    rtl_cycles = {"encode_post": 2100, "mul_plain": 1200, "hom_add": 1300, "keyswitch_rotate": 20600}
    res = {"note": "us_per_op: back-to-back runs of ONE program (aloha_vp_run at the reference's entry point), one GPU, operands resident; "
                   "rtl_derived_us: DERIVED from the RTL's cycle model at the test bench's 100 MHz clock (BASELINE.md section 2), not measured"}
    def chain(e):
        tb = H.TestBench(e)
        tb.load_programs({"encode_post": images["encode_post"], "mul_plain": images["mul_plain"], "hom_add": images["hom_add"], "keyswitch": images["keyswitch"]})
        tb.load_ksk([ld(m["ksk_step2"])])
        tb.dram_ct[0] = ld(m["inst_0_out"])
        tb.dram_encoder[0x0] = ld(m["inst_2_0_out"]); tb.dram_encoder[0x40000] = ld(m["inst_3_0_out"])
        dumps = {}
        tb.dump = lambda k, sub, data: dumps.__setitem__(f"{k}{sub}", data)
        tb.run(ops)
        ok = True
        for k in range(1, 8):
            exp = ld(m[f"inst_{k}_out"])
            ok = ok and bool(np.array_equal(dumps[str(k)][:exp.size], exp))
        return ok
    with avp.VP(spm_rows=0x800, ksk_rows=1024, max_log_n=13, device=device) as vp:
        res["chain_case0_equals_rtl_dumps"] = chain(vp)
        o = oracle.OracleVP(spm_rows=0x800, ksk_rows=1024)
        chain(o)
        for name, pc, s0, s1, step in progs:
            run = lambda: vp.run(pc, src0=s0, src1=s1, rslt=0x600, ksk=0, step=step)
            assert run() == 0
            t0 = time.perf_counter(); assert o.run(pc, src0=s0, src1=s1, rslt=0x600, ksk=0, step=step) == 0; cpu_us = (time.perf_counter() - t0) * 1e6
            same = bool(np.array_equal(vp.read(0, 0x600, 4 * n), o.read(0, 0x600, 4 * n)))     # (the scratch destination: GPU == CPU port, word for word)
            l0 = vp.stats()["kernel_launches"]
            assert run() == 0
            launches = float(vp.stats()["kernel_launches"] - l0)
            t = timeit(run, vp.sync, 250)
            t1 = timeit(lambda: (run(), vp.sync()), vp.sync, 100)
            res[name] = {"pc": pc, "us_per_op": t * 1e6, "single_call_with_sync_us": t1 * 1e6, "launches": launches, "rtl_derived_us": rtl_cycles[name] / 100.0,
                         "cpu_port_us": cpu_us, "scratch_run_equals_cpu_port": same}
        o.close()
    return res


def settle(step, sync, min_ms: float = 120.0, max_ms: float = 1500.0, block_ms: float = 20.0, tol: float = 0.01):
    """Untimed steps until the step time is stable: the clocks of a freshly loaded MI355X dip ~25 % for the first
    milliseconds of sustained launches and take ~100 ms to settle, and the driver's short runs
    (--steps 20) would otherwise time that transient.  Runs blocks of ~block_ms, stops once at least min_ms have
    passed and two consecutive blocks agree within tol (or at max_ms).  Returns (ms spent, steps run)."""
    t_start = time.perf_counter()
    step(); sync()
    t0 = time.perf_counter(); step(); sync()
    one = max(1e-5, time.perf_counter() - t0)
    k = max(1, int(block_ms / 1e3 / one))
    prev, total = None, 2
    while True:
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        sync()
        cur = (time.perf_counter() - t0) / k
        total += k
        spent = (time.perf_counter() - t_start) * 1e3
        if (spent >= min_ms and prev is not None and abs(cur - prev) <= tol * prev) or spent >= max_ms:
            return spent, total
        prev = cur


def keyswitch_leg(dist, backend, rank, world, local_rank, batch=1):
    """BASELINE config 4 on the ranks of this run: HMult + relinearisation at N = 2^16, L = 24, hybrid key-switch
    with alpha = K = 4 (aloha_amd/hybrid.py), RNS limbs sharded limb j -> rank j mod world, the digit all-gather and
    the mod-down all-gather over torch.distributed (keyswitch.mem:4-43,:80-83 are the steps they parallelise).  Timed
    with the collective and with world-1 direct sends per rank; every rank checks its owned output limbs against a
    single-rank engine running the same operation on the same inputs."""
    from aloha_amd import hybrid, keyswitch
    from aloha_amd import vp as avp
    L, alpha, reps = 24, 4, 10
    warm = 200 // batch                                          # untimed calls before the timed ones (clock settle; the same count on every rank)
    mods = [avp.nth_prime(LOGN, i) for i in range(L)]
    specials = [avp.nth_prime(LOGN, L + k) for k in range(alpha)]
    on_gpu = backend == "nccl"

    def load(ks, lo):
        for jl, j in enumerate(lo.owned_all):                   # key and ciphertext words depend on the limb only
            kl = np.random.default_rng(7000 + j).integers(0, (mods + specials)[j], lo.dnum * 2 * N, dtype=np.uint64)
            ks.vp.write(1, (jl * lo.dnum) * 2 * lo.rows, kl)
        ct = np.stack([np.random.default_rng(539 + j).integers(0, mods[j % L], N, dtype=np.uint64) for j in range(2 * L)])
        for b in range(batch):                                   # ciphertext b = the same words rolled by b (cheap, distinct)
            ks.load_ct(np.roll(ct, b, axis=1), b); ks.load_ct_b(np.roll(ct[::-1], b, axis=1).copy(), b)

    res = {"config": f"HMult + relinearisation, N=2^16, L={L} (+{alpha} special), hybrid key-switch alpha={alpha}, limbs sharded j -> rank j mod {world}",
           "n_gpus": world, "reps": reps, "backend": backend, "batch": batch}
    t_of = lambda *a: torch.tensor(list(a), dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    out = None
    for mode in ("collective", "p2p"):
        stream = torch.cuda.Stream()
        lo = hybrid.HybridLayout(N, mods, specials, alpha, world, rank, batch=batch)
        ex = keyswitch.Exchange(tensor_of=keyswitch.device_tensor if on_gpu else None, stream=stream if on_gpu else None,
                                p2p=(mode == "p2p"))
        with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=max(1, lo.ksk_limbs * lo.rows), max_log_n=LOGN, device=local_rank,
                    stream=stream.cuda_stream) as vp:
            ks = hybrid.HybridKeySwitch(vp, N, mods, specials, alpha, world=world, rank=rank, exchange=ex, batch=batch)
            load(ks, lo)
            ks.multiply(); vp.sync()
            for _ in range(warm):                                # untimed, the SAME count on every rank (collectives inside): settled clocks
                ks.multiply()
            vp.sync()
            dist.barrier(); torch.cuda.synchronize()
            sent0, recv0 = ex.bytes_sent, ex.bytes_received
            t0 = time.perf_counter()
            for _ in range(reps):
                ks.multiply()
            vp.sync()
            el = time.perf_counter() - t0
            t = t_of(el); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[f"ms_{mode}"] = float(t[0]) / reps * 1e3
            res[f"bytes_sent_per_rank_{mode}"] = (ex.bytes_sent - sent0) // reps
            res[f"bytes_received_per_rank_{mode}"] = (ex.bytes_received - recv0) // reps
            got = np.stack([ks.read_out(b) for b in range(batch)])
            if out is None:
                out = got
                owned = list(lo.owned_ct)
            else:                                                # both exchange forms must give the same words
                same = bool(np.array_equal(out, got))
                t = t_of(1.0 if same else 0.0); dist.all_reduce(t, op=dist.ReduceOp.MIN)
                res["p2p_equals_collective"] = bool(t[0] > 0.5)
    lo1 = hybrid.HybridLayout(N, mods, specials, alpha, 1, 0, batch=batch)
    with avp.VP(spm_rows=lo1.spm_rows_needed, ksk_rows=lo1.ksk_limbs * lo1.rows, max_log_n=LOGN, device=local_rank) as vp1:
        ks1 = hybrid.HybridKeySwitch(vp1, N, mods, specials, alpha, world=1, rank=0, exchange=keyswitch.Exchange(), batch=batch)
        load(ks1, lo1)
        ks1.multiply(); vp1.sync()
        for _ in range(warm):
            ks1.multiply()
        vp1.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ks1.multiply()
        vp1.sync()
        res["ms_single_rank"] = (time.perf_counter() - t0) / reps * 1e3      # the same operation on ONE GPU, for the speed-up
        ref = np.stack([ks1.read_out(b) for b in range(batch)])
    ok = all(np.array_equal(out[:, :, i], ref[:, :, j]) for i, j in enumerate(owned))
    t = t_of(1.0 if ok else 0.0); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    res["verified"] = bool(t[0] > 0.5)
    res["ms"] = min(res["ms_collective"], res["ms_p2p"])
    res["ms_per_ciphertext"] = res["ms"] / batch
    res["exchange"] = "p2p" if res["ms_p2p"] < res["ms_collective"] else "collective"
    return res


def keyswitch_leg_c_host(dist, rank, world, local_rank):
    """The same operation through the C host (aloha_ks_*, csrc/keyswitch_host.cpp): programs generated in C++, the two
    all-gathers issued by the library as ncclAllGather on a communicator of its own.  torch.distributed (gloo) only
    carries the 128-byte ncclUniqueId and the timing reduce.  Verified against a single-rank C host on every rank."""
    from aloha_amd import ks as cks
    from aloha_amd import vp as avp
    L, alpha, reps, warm = 24, 4, 10, 200
    mods = [avp.nth_prime(LOGN, i) for i in range(L)]
    specials = [avp.nth_prime(LOGN, L + k) for k in range(alpha)]
    dnum = -(-L // alpha)
    ct = np.stack([np.random.default_rng(539 + j).integers(0, mods[j % L], N, dtype=np.uint64) for j in range(2 * L)])

    def make(w, r, uid):
        spm, ksk = cks.rows_needed(LOGN, mods, specials, alpha, w, r)
        vp = avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=LOGN, device=local_rank)
        k = cks.KeySwitch(vp, LOGN, mods, specials, alpha, w, r)
        if uid is not None:
            k.comm_init(uid)
        for jl, j in enumerate(k.owned_all):                     # key words depend on the modulus only (as in keyswitch_leg)
            vp.write(1, jl * dnum * 2 * (N // 128), np.random.default_rng(7000 + j).integers(0, (mods + specials)[j], dnum * 2 * N, dtype=np.uint64))
        k.load_ct(ct); k.load_ct_b(ct[::-1].copy())
        return vp, k

    uid = [cks.nccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    vp, k = make(world, rank, uid[0])
    res, outs = {}, []
    for mode in ("collective", "p2p"):
        k.set_p2p(mode == "p2p")
        k.multiply(); vp.sync()
        for _ in range(warm):                                    # untimed, the same count on every rank: settled clocks
            k.multiply()
        vp.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            k.multiply()
        vp.sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[f"ms_{mode}"] = float(t[0]) / reps * 1e3
        outs.append(k.read_out())
    out, owned, info = outs[0], list(k.owned_ct), k.info()
    k.close(); vp.close()
    vp1, k1 = make(1, 0, None)
    k1.multiply(); vp1.sync()
    for _ in range(warm):
        k1.multiply()
    vp1.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        k1.multiply()
    vp1.sync()
    res["ms_single_rank"] = (time.perf_counter() - t0) / reps * 1e3
    ref = k1.read_out()
    k1.close(); vp1.close()
    good = np.array_equal(outs[0], outs[1]) and all(np.array_equal(out[:, i], ref[:, j]) for i, j in enumerate(owned))
    ok = torch.tensor([1.0 if good else 0.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    res.update({"ms": min(res["ms_collective"], res["ms_p2p"]), "verified": bool(ok[0] > 0.5),
                "bytes_sent_per_rank": info["bytes_sent"] // (2 * (reps + 1)),
                "exchange": "ncclAllGather / grouped ncclSend+ncclRecv issued by the C host on its own communicator, stream-ordered"})
    return res


def keyswitch_child() -> int:
    """One rank of the key-switch leg, in a process group of its own (see keyswitch_in_children)."""
    import torch.distributed as dist
    from datetime import timedelta
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("ALOHA_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("ALOHA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local_rank)
    c_host = os.environ.get("ALOHA_KS_CHILD_MODE") == "c_host"
    if backend == "nccl" and not c_host:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(seconds=120))
    else:                                                        # the C host brings its own RCCL communicator: gloo carries the id and the timings
        dist.init_process_group(backend="gloo", timeout=timedelta(seconds=120))
    if c_host:
        res = keyswitch_leg_c_host(dist, rank, world, local_rank)
    else:
        # one ciphertext per call, and four: a single key-switch spread over 8 GPUs leaves 3-4 moduli per rank and is
        # bound by kernel latency (rank 0's share 0.11 ms against 0.28 / 8 ideal); `batch` ciphertexts under one key
        # keep the per-rank launches large enough to fill a GPU
        res = keyswitch_leg(dist, backend, rank, world, local_rank, batch=1)
        res["batch4"] = keyswitch_leg(dist, backend, rank, world, local_rank, batch=4)
    if rank == 0:
        print("KEYSWITCH_JSON " + json.dumps(res), flush=True)
    dist.destroy_process_group()
    return 0


def keyswitch_child_env(environ: dict, mode: str, port_shift: int) -> dict:
    """Environment of one rank of a key-switch child group: the rank's own RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous
    port of its own (derived from the run's MASTER_PORT, different for each of the groups a run creates: the run itself,
    the `torch` child group, the `c_host` child group), and none of the elastic agent's variables (under torchrun the
    agent's store serves MASTER_PORT only: without them rank 0 of the child group hosts its own)."""
    env = {k: v for k, v in environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str((int(environ.get("MASTER_PORT", "29500")) + port_shift) % 60000 + 2048)
    env["ALOHA_KS_CHILD_MODE"] = mode
    return env


KS_CHILD_PORT_SHIFTS = {"torch": 1009, "c_host": 2027}


def keyswitch_in_children(rank: int, budget_s: float = 90.0):
    """The sharded key-switch leg runs in CHILD processes with a rendezvous of their own (every rank of this run starts
    one and waits for it with a time limit): its collectives have never met more than one GPU at the time of writing,
    and a failure or a hang there must not cost the headline line -- it becomes {"error": ...} in the `keyswitch` object.
    ONE budget for the whole leg: the second child gets what the first left, and is skipped if the first timed out."""
    import subprocess
    t_end = time.monotonic() + budget_s

    def child(mode: str):
        left = t_end - time.monotonic()
        if left < 10.0:
            return {"error": "skipped: the leg's time budget was spent"}
        env = keyswitch_child_env(dict(os.environ), mode, KS_CHILD_PORT_SHIFTS[mode])
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--keyswitch-child"], env=env, capture_output=True, text=True,
                               timeout=left)
        except subprocess.TimeoutExpired:
            return {"error": f"did not finish within {left:.0f} s (killed)", "timed_out": True}
        for line in p.stdout.splitlines():
            if line.startswith("KEYSWITCH_JSON "):
                return json.loads(line[len("KEYSWITCH_JSON "):])
        return {"error": f"exited with code {p.returncode}", "stderr_tail": p.stderr[-600:]}

    res = child("torch")                                         # Python host + torch.distributed (collective and point-to-point; batch 1 and 4)
    c_host = None
    if os.environ.get("ALOHA_BENCH_BACKEND", "nccl") == "nccl":  # (two ranks of a rehearsal on ONE GPU cannot form an RCCL communicator)
        c_host = {"error": "skipped: the first child timed out"} if res.get("timed_out") else child("c_host")
    if rank != 0:
        return None
    if c_host is not None:
        res["c_host"] = c_host
    return res


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--primes", type=int, default=32)
    ap.add_argument("--polys", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed settle loop (A/B: shows the clock transient)")
    ap.add_argument("--no-keyswitch", action="store_true", help="N > 1 only: skip the sharded key-switch leg")
    ap.add_argument("--no-configs", action="store_true", help="N = 1 only: skip the BASELINE configs 2-5 leg (~40 s)")
    ap.add_argument("--inverse", action="store_true", help="time vintt instead (side metric; headline is forward)")
    ap.add_argument("--keyswitch-child", action="store_true", help=argparse.SUPPRESS)   # internal: one rank of the key-switch leg
    args = ap.parse_args()
    if args.keyswitch_child:
        return keyswitch_child()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(sys.argv[1:], args.gpus)             # nothing above has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; the group size is what is reported", file=sys.stderr)
    if os.environ.get("ALOHA_BENCH_ECHO"):                       # (tests: shows what a rank was started with)
        print(f"bench.py rank {rank} of {world} local {local_rank} master {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
              file=sys.stderr)
    if os.environ.get("ALOHA_BENCH_TEST_DIE"):                   # (tests: "<rank>:<signal>" -- that rank dies by the signal, the others wait)
        r, sig = (int(x) for x in os.environ["ALOHA_BENCH_TEST_DIE"].split(":"))
        if rank == r:
            os.kill(os.getpid(), sig)
        time.sleep(120)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the engine has no CPU path", file=sys.stderr)
        return 2
    # Rehearsal knobs (not used by the driver): ALOHA_BENCH_BACKEND=gloo and ALOHA_BENCH_DEVICE=0
    # let several ranks share one GPU to exercise the multi-rank control flow on a 1-GPU box.
    backend = os.environ.get("ALOHA_BENCH_BACKEND", "nccl")
    if "ALOHA_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ALOHA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ALOHA_BENCH_FORCE_DIST"):      # (the knob exercises the RCCL path with one rank)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        world = dist.get_world_size()                            # n_gpus in the JSON line is the group's size

    from aloha_amd import vp as avp

    L, P = args.primes, args.polys
    B = L * P
    rows_per_limb = N // 128
    stream = torch.cuda.Stream()
    vp = avp.VP(spm_rows=B * rows_per_limb, ksk_rows=1, max_log_n=LOGN, device=local_rank,
                stream=stream.cuda_stream)
    primes = [avp.nth_prime(LOGN, i) for i in range(L)]
    moduli = [q for q in primes for _ in range(P)]
    # synthetic inputs: the L largest primes p < 2^60, p = 1 mod 2N; limb i = canonical residues uniform in
    # [0, p) from splitmix64(0xA10A + global limb index) with rejection, generated on the device
    vp.fill_synthetic(avp.SPM, 0, LOGN, moduli, seed_base=0xA10A, first_index=rank * B)
    vp.sync()
    first_in = vp.read(avp.SPM, 0, N)
    assert int(first_in.max()) < primes[0]
    # correctness guards inside the bench: the first limb of the first step is kept for the CPU leg
    # (cpu_baseline compares it word for word), ...
    vp.ntt_rows(avp.SPM, 0, 0, LOGN, moduli)
    vp.sync()
    first_out = vp.read(avp.SPM, 0, N)
    if rank == 0 and not os.environ.get("ALOHA_VP_LIB"):   # (ablation builds compute garbage on purpose)
        last = (B - 1) * rows_per_limb
        # ... and the last limb of the batch goes a round trip through the inverse transform
        before = vp.read(avp.SPM, last, N)
        vp.ntt_rows(avp.SPM, last, last, LOGN, [primes[-1]], inverse=True)
        vp.ntt_rows(avp.SPM, last, last, LOGN, [primes[-1]])
        if not np.array_equal(vp.read(avp.SPM, last, N), before):
            print("bench.py: NTT round trip failed on the last limb", file=sys.stderr)
            return 3
    inv = bool(args.inverse)
    step = lambda: vp.ntt_rows(avp.SPM, 0, 0, LOGN, moduli, inverse=inv)
    for _ in range(max(0, args.warmup - 1)):
        step()
    vp.sync()
    settle_ms, settle_steps = (0.0, 0) if args.no_settle else settle(step, vp.sync)
    launches0 = vp.stats()["kernel_launches"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region: EXACTLY K steps between barriers.  A process that overdraws the container's CPU quota is paused by the
    # kernel for 20-80 ms (gaps between launches in rocprofv3 traces, never long kernels; profiles/r03_z_launch_gaps.txt: the cause
    # found in round 3, the oracle's spinning OpenMP workers, is fixed); with K = 20 such a pause is longer than the whole region.  It shows as wall time far above the device time of the same K steps (HIP events on the
    # engine's stream); such an attempt is repeated (at most twice) and reported in `discarded_attempts`.
    discarded = []
    for attempt in range(3):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launches0 = vp.stats()["kernel_launches"]
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ev0.record(stream)
            for _ in range(args.steps):
                step()
            ev1.record(stream)
        vp.sync()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        dev_ms = ev0.elapsed_time(ev1)                          # device time of the K steps on the engine's stream
        launches = vp.stats()["kernel_launches"] - launches0
        if dist is not None:
            t = torch.tensor([elapsed, dev_ms], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, dev_ms = float(t[0]), float(t[1])
        if elapsed * 1e3 <= 1.25 * dev_ms + 2.0 or attempt == 2:   # (the same numbers on every rank: the same decision)
            break
        discarded.append({"wall_ms": elapsed * 1e3, "device_ms": dev_ms})
    clock_ghz = vp.valu_clock_ghz(60.0) if rank == 0 else None   # right behind the timed region: the same thermal state
    vp.close()
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            configs = configs_leg(local_rank)
        except Exception as e:                                   # (the headline line does not depend on this leg)
            configs = {"error": f"{type(e).__name__}: {e}"}

    ks = None
    if dist is not None and world > 1 and not args.no_keyswitch:
        if rank == 0:                                            # the headline is on record before the leg that has never met N GPUs starts
            print("BENCH_HEADLINE " + json.dumps({"value": world * B * args.steps / elapsed, "unit": "NTT/s", "n_gpus": world,
                                                  "ms_per_step": elapsed / args.steps * 1e3}), file=sys.stderr, flush=True)
        ks = keyswitch_in_children(rank)

    if rank == 0:
        total_ntts = world * B * args.steps
        alg_bytes_step = 16 * N * B                         # read N + write N words per transform
        dev_s_step = dev_ms / 1e3 / args.steps
        achieved = alg_bytes_step / dev_s_step / 1e9
        hbm_bound_ntt_s = HBM_PEAK_GBS * 1e9 / (16 * N)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_per_step(B),
                "note": "achieved = 16*N*B algorithmic bytes / device time of one step (both tile kernels, HIP events on "
                        f"the engine's stream); traffic = HBM bytes per step from profiles/{PMC_SUMMARY} (rocprofv3 --pmc, "
                        "FETCH_SIZE x2 + WRITE_SIZE; two passes + twiddle blocks => ~2.04x algorithmic; its `commit` field names "
                        "the kernels it was collected on); valu = the second bound (SURVEY 8d): lane-instructions per transform "
                        "(SQ_INSTS_VALU of the same summary) against 16 lanes x SIMDs x the VALU clock measured by a multiply-add "
                        "calibration kernel right after the timed region; `bound` names the tighter of the two"}
        insts = valu_insts_per_ntt()
        if insts and clock_ghz:
            simds = 1024                                     # 256 CUs x 4 SIMDs of 16 lanes (MI355X_MICROARCH.md)
            valu_bound = 16.0 * simds * clock_ghz * 1e9 / insts
            roof["valu"] = {"insts_per_ntt": insts, "insts_per_butterfly": insts / (N // 2 * LOGN), "clock_ghz": clock_ghz,
                            "bound_ntt_per_s": valu_bound, "frac_of_valu_bound": (B / dev_s_step) / valu_bound,
                            "as_frac_of_hbm_roofline": valu_bound / hbm_bound_ntt_s}
            roof["hbm_bound_ntt_per_s"] = hbm_bound_ntt_s
            roof["peak_is"] = "hbm"                          # achieved / peak / frac / traffic are the HBM roofline's whatever binds
            if valu_bound < hbm_bound_ntt_s:
                roof["bound"] = "valu"                       # the tighter of the two textbook bounds ("measure and print both")
        # what actually binds: the tightest of the design's measured ceilings (the two-pass memory floor, then the VALU bound)
        ceilings = {k: dict(v) for k, v in CEILINGS.items()}
        if "valu" in roof:
            ceilings["valu"] = {"frac": roof["valu"]["as_frac_of_hbm_roofline"], "source": f"this run + profiles/{PMC_SUMMARY}",
                                "what": "lane-instructions per transform (SQ_INSTS_VALU) at 16 lanes x 1024 SIMDs x the VALU clock measured after the timed region"}
        roof["ceilings"] = ceilings
        tight = min(ceilings, key=lambda k: ceilings[k]["frac"])
        roof["bound"] = "hbm"                                # the binding ceiling is a MEMORY one: two HBM round trips at the rate the access patterns reach
        roof["binding_ceiling"] = tight
        roof["frac_of_binding_ceiling"] = roof["frac"] / ceilings[tight]["frac"]
        pm = pmc_summary()
        if pm.get("commit"):
            roof["traffic_commit"] = pm["commit"]
        out = {
            "metric": ("inv-NTT/s" if inv else "fwd-NTT/s") + " at N=2^16, 60-bit prime", "value": total_ntts / elapsed, "unit": "NTT/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "discarded_attempts": discarded,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "settle_ms": settle_ms, "settle_steps": settle_steps,
            "config": {"workload": f"{'inverse negacyclic NTT (vintt)' if inv else 'forward negacyclic NTT (vntt)'}, N=2^16, batch of {B} limbs per GPU "
                                   f"({L} 60-bit primes x {P} limbs), in place in HBM ({B * N * 8 >> 20} MiB); inputs: splitmix64(0xA10A + limb) residues",
                       "log_n": LOGN, "limbs_per_gpu": B, "primes": L, "limbs_per_prime": P,
                       "kernels_per_step": launches // max(1, args.steps)},
            "roofline": roof,
        }
        if ks is not None:
            out["keyswitch"] = ks
        if configs is not None:
            out["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(check=(first_in, primes[0], first_out))
            if out["cpu_baseline"]["gpu_first_limb_equals_cpu"] is False and not os.environ.get("ALOHA_VP_LIB"):
                print("bench.py: GPU result differs from the CPU port", file=sys.stderr)
                return 3
        alg_tw = alg_bytes_step + 2 * 8 * N * L             # + one twiddle table and its Shoup companion per prime
        out["roofline"]["achieved_twiddle_inclusive"] = alg_tw / dev_s_step / 1e9
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
