#!/usr/bin/env python3
"""bench.py -- headline benchmark: forward negacyclic NTT/s at N = 2^16, 60-bit primes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--primes L] [--polys P]

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
PMC_SUMMARY = "r01_s_pmc_summary.json"   # written by tools/collect_pmc.sh for the kernels of this commit


def measured_traffic_per_step(limbs: int):
    """HBM bytes per step from the committed rocprofv3 PMC summary (FETCH_SIZE x2 per the guide's
    gfx950 correction + WRITE_SIZE, separate --pmc passes of this same command), scaled to the
    batch size; None when no summary is present."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    try:
        with open(path) as f:
            d = json.load(f)["derived"]
        return d["hbm_bytes_per_step_corrected"] * limbs / (d["algorithmic_bytes_per_step"] / (16 * N))
    except Exception:
        return None


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
    return {"value": multi, "unit": "NTT/s", "cores": cores, "kind": "port",
            "single_thread": single, "gpu_first_limb_equals_cpu": agrees,
            "sample": f"{nm} forward NTTs on {cores} threads + {n1} on one thread, N=2^16, one 60-bit prime, "
                      f"{seconds:.0f} s in all; Verilog simulators on this host: {sims or 'none'} "
                      "(the reference's own test bench cannot be run)",
            "reference_rtl_derived": {"note": "DERIVED, not measured: the RTL's cycle model at the test bench's 100 MHz "
                                              "clock, N=8192 only (ntt_fsm.sv:68, vp_defines.vh:79-81, top_noaxilite_tb.sv:232-235)",
                                      "cycles_per_ntt_n8192": 850, "ntt_per_s_n8192": 1e8 / 850}}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--primes", type=int, default=32)
    ap.add_argument("--polys", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inverse", action="store_true", help="time vintt instead (side metric; headline is forward)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}", file=sys.stderr)
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

    from aloha_amd import vp as avp

    L, P = args.primes, args.polys
    B = L * P
    rows_per_limb = N // 128
    stream = torch.cuda.Stream()
    vp = avp.VP(spm_rows=B * rows_per_limb, ksk_rows=1, max_log_n=LOGN, device=local_rank,
                stream=stream.cuda_stream)
    primes = [avp.nth_prime(LOGN, i) for i in range(L)]
    moduli = [q for q in primes for _ in range(P)]
    # synthetic canonical residues, uniform in [0, q), generated on the device straight into the
    # engine's memory (a zero-copy int64 view), prime by prime
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xA10A + rank)
    for i, q in enumerate(primes):
        class _View:
            __cuda_array_interface__ = {"shape": (P * N,), "typestr": "<i8", "version": 2,
                                        "data": (vp.device_ptr(avp.SPM, i * P * rows_per_limb), False)}
        torch.as_tensor(_View(), device="cuda").random_(0, q, generator=gen)
    torch.cuda.synchronize()
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
    for _ in range(max(0, args.warmup - 1)):
        vp.ntt_rows(avp.SPM, 0, 0, LOGN, moduli, inverse=inv)
    vp.sync()
    launches0 = vp.stats()["kernel_launches"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(args.steps):
            vp.ntt_rows(avp.SPM, 0, 0, LOGN, moduli, inverse=inv)
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

    if rank == 0:
        total_ntts = world * B * args.steps
        alg_bytes_step = 16 * N * B                         # read N + write N words per transform
        dev_s_step = dev_ms / 1e3 / args.steps
        achieved = alg_bytes_step / dev_s_step / 1e9
        out = {
            "metric": ("inv-NTT/s" if inv else "fwd-NTT/s") + " at N=2^16, 60-bit prime", "value": total_ntts / elapsed, "unit": "NTT/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{'inverse negacyclic NTT (vintt)' if inv else 'forward negacyclic NTT (vntt)'}, N=2^16, batch of {B} limbs per GPU "
                                   f"({L} 60-bit primes x {P} limbs), in place in HBM ({B * N * 8 >> 20} MiB)",
                       "log_n": LOGN, "limbs_per_gpu": B, "primes": L, "limbs_per_prime": P,
                       "kernels_per_step": launches // max(1, args.steps)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_per_step(B),
                         "note": "achieved = 16*N*B algorithmic bytes / device time of one step (both "
                                 "tile kernels, HIP events on the engine's stream); traffic = HBM bytes "
                                 f"per step from profiles/{PMC_SUMMARY} (rocprofv3 --pmc, FETCH_SIZE x2 + "
                                 "WRITE_SIZE; two passes + twiddle blocks => ~2.04x algorithmic); see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(check=(first_in, primes[0], first_out))
            if out["cpu_baseline"]["gpu_first_limb_equals_cpu"] is False and not os.environ.get("ALOHA_VP_LIB"):
                print("bench.py: GPU result differs from the CPU port", file=sys.stderr)
                return 3
        alg_tw = alg_bytes_step + 2 * 8 * N * L             # + one twiddle table and its Shoup companion per prime
        out["roofline"]["achieved_twiddle_inclusive"] = alg_tw / dev_s_step / 1e9
        print(json.dumps(out))
    vp.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
