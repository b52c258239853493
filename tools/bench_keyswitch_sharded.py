#!/usr/bin/env python3
"""BASELINE configs 4 / 5 on G GPUs: rotate + key-switch or ciphertext product + relinearisation at
N = 2^16 with the RNS limbs sharded over the ranks (limb j on rank j mod G), one process per GPU,
exchanges over torch.distributed (nccl = RCCL over xGMI; zero-copy views of engine memory).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \\
        tools/bench_keyswitch_sharded.py --L 24 --alpha 4 --op multiply

--alpha 0 runs the reference's algorithm (one digit per prime, one special prime: keyswitch.py), any
other value the hybrid key-switch (hybrid.py) with K = alpha special primes.  Rank 0 prints one JSON
line.  Rehearsal on a one-GPU box: ALOHA_BENCH_BACKEND=gloo ALOHA_BENCH_DEVICE=0 lets several ranks share
the GPU (the exchange then goes through host copies).  The result is checked: every rank's owned limbs
against a single-rank run of the oracle's ISA machine would take minutes at this size, so the check is the
2-rank == 1-rank parity of tests/ at small N plus, here, rank-count independence of a checksum when
--verify is given (a second, single-process engine computes the same operation on rank 0)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=24)
    ap.add_argument("--alpha", type=int, default=4)
    ap.add_argument("--op", choices=["rotate", "multiply", "hoisted"], default="multiply")
    ap.add_argument("--rotations", type=int, default=8, help="--op hoisted: rotation steps per decomposition (one key each)")
    ap.add_argument("--many", action="store_true", help="--op hoisted: all rotations in one pass of the phases (rotate_hoisted_many)")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=120.0, help="untimed back-to-back calls before the timed ones (0 = none: times the clock ramp of an idle GPU)")
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="ciphertexts per call (hybrid only)")
    ap.add_argument("--p2p", action="store_true", help="all-gathers as world-1 direct sends/receives per rank instead of the collective")
    ap.add_argument("--host-sync", action="store_true", help="bracket every exchange with host synchronisation (the old path)")
    ap.add_argument("--c-host", action="store_true",
                    help="run the key-switch through the C host (aloha_ks_*: programs generated in C++, ncclAllGather issued by the "
                         "library on its own communicator; torch.distributed only carries the 128-byte ncclUniqueId and the timing reduce)")
    ap.add_argument("--rom", type=int, default=4096, help="instruction ROM words of the engine (4096 = the reference's; deeper holds a whole phase at L = 32)")
    ap.add_argument("--stub-world", type=int, default=0,
                    help="time rank 0's share of a W-rank layout on this one GPU with the exchanges skipped (results meaningless)")
    args = ap.parse_args()
    if args.many:
        assert args.op == "hoisted" and args.batch == 1
        args.batch = args.rotations                              # the batch slots are the rotation slots
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.stub_world:
        assert world == 1 and not args.verify
        world = args.stub_world
    local = int(os.environ.get("ALOHA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("ALOHA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1 and not args.stub_world:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from aloha_amd import hybrid, keyswitch
    from aloha_amd import vp as avp

    logn, n, L = args.log_n, 1 << args.log_n, args.L
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    K = max(1, args.alpha)
    specials = [avp.nth_prime(logn, L + k) for k in range(K)]
    stream = torch.cuda.Stream()
    ex = keyswitch.Exchange(tensor_of=keyswitch.device_tensor if backend == "nccl" else None, always=False,
                            stream=None if (args.host_sync or backend != "nccl") else stream, stub=bool(args.stub_world), p2p=args.p2p)

    class CHost:                                                 # the C host behind the call surface the script uses
        def __init__(self, vp, w, r):
            from aloha_amd import ks as cks
            self.vp = vp
            self.k = cks.KeySwitch(vp, logn, mods, specials, args.alpha, w, r, args.batch)
            if w > 1:
                uid = [cks.nccl_unique_id() if r == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                self.k.comm_init(uid[0])
        def load_ct(self, ct, b=0): self.k.load_ct(ct, b)
        def load_ct_b(self, ct, b=0): self.k.load_ct_b(ct, b)
        def rotate(self, k): self.k.rotate(k)
        def multiply(self): self.k.multiply()
        def read_out(self, b=0): return self.k.read_out(b)

    def build(vp, w, r, exch):
        if args.c_host:
            assert args.alpha > 0 and args.op in ("rotate", "multiply") and not args.stub_world
            return CHost(vp, w, r)
        if args.alpha == 0:
            return keyswitch.ShardedKeySwitch(vp, n, mods, specials[0], world=w, rank=r, exchange=exch)
        return hybrid.HybridKeySwitch(vp, n, mods, specials, args.alpha, world=w, rank=r, exchange=exch, batch=args.batch)

    def layout(w, r):
        return keyswitch.Layout(n, mods, specials[0], w, r) if args.alpha == 0 else hybrid.HybridLayout(n, mods, specials, args.alpha, w, r, batch=args.batch)

    def load(ks, lo, seed=7):
        """Same pseudo-random ciphertexts and key on every rank count: limb contents depend on the limb only."""
        all_mods = mods + (specials[:1] if args.alpha == 0 else specials)
        digits = L if args.alpha == 0 else lo.dnum
        for jl, j in enumerate(lo.owned_all):
            rng = np.random.default_rng(seed * 1000 + j)
            kl = rng.integers(0, all_mods[j], digits * 2 * n, dtype=np.uint64)
            for r in range(nkeys):                                # (the same words serve as every rotation's key: timing only)
                ks.vp.write(1, (r * lo.Mr * digits + jl * digits) * 2 * lo.rows, kl)
        ct = np.stack([np.random.default_rng(seed * 77 + j).integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
        if args.alpha == 0:
            ks.load_ct(ct); ks.load_ct_b(ct[::-1].copy())
        else:
            for b in range(args.batch):
                ks.load_ct(ct, b); ks.load_ct_b(ct[::-1].copy(), b)

    nkeys = args.rotations if args.op == "hoisted" else 1
    assert args.op != "hoisted" or args.alpha > 0
    lo = layout(world, rank)
    vp = avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=max(1, nkeys * lo.ksk_limbs * lo.rows), max_log_n=logn, device=local,
                stream=stream.cuda_stream, iram_depth=args.rom)
    ks = build(vp, world, rank, ex)
    load(ks, lo)
    k = pow(3, 5, 2 * n)
    steps = [pow(3, r + 1, 2 * n) for r in range(nkeys)]

    def run_op(ks):
        if args.op == "rotate":
            ks.rotate(k)
        elif args.op == "multiply":
            ks.multiply()
        elif args.many:
            ks.hoist_one()
            ks.rotate_hoisted_many(steps)
        else:
            ks.hoist()
            for r, kr in enumerate(steps):
                ks.rotate_hoisted(kr, r)
    op = lambda: run_op(ks)

    def barrier():
        if dist.is_initialized() and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    op(); vp.sync()
    # settled clocks (as bench.py): the first ~100 ms of launches after an idle spell run up to 9 % slow (profiles/r04_configs_clock_settle.txt)
    t_settle = time.perf_counter()
    while args.settle_ms and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(8):
            op()
        vp.sync()
    barrier()
    launches0 = vp.stats()["kernel_launches"]
    t0 = time.perf_counter()
    for _ in range(args.reps):
        op()
    enqueue = time.perf_counter() - t0                          # host time to issue everything (the GPU may lag behind)
    vp.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist.is_initialized() and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    out = ks.read_out()
    ok = None
    if args.verify:                                              # every rank: its owned limbs == a single-rank engine's
        lo1 = layout(1, 0)
        with avp.VP(spm_rows=lo1.spm_rows_needed, ksk_rows=nkeys * lo1.ksk_limbs * lo1.rows, max_log_n=logn, device=local) as vp1:
            ks1 = build(vp1, 1, 0, keyswitch.Exchange())
            load(ks1, lo1)
            run_op(ks1)
            refs = [ks1.read_out(b) for b in range(args.batch)] if args.alpha else [ks1.read_out()]
        outs = [ks.read_out(b) for b in range(args.batch)] if args.alpha else [out]
        ok = all(np.array_equal(o[:, i], r[:, j]) for o, r in zip(outs, refs) for i, j in enumerate(lo.owned_ct))
        if dist.is_initialized() and world > 1:
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t[0] > 0.5)
    if rank == 0:
        print(json.dumps({"metric": f"{args.op} + key-switch at N=2^{logn}", "L": L, "alpha": args.alpha, "K": K, "n_gpus": world,
                          "batch": args.batch, "one_pass": bool(args.many), "host": "C (aloha_ks_*, own RCCL communicator)" if args.c_host else "python + torch.distributed",
                          "exchange": "stubbed (rank 0's share only)" if args.stub_world else
                          ("host-synchronised" if ex.stream is None else "stream-ordered"),
                          "launches_per_op": (vp.stats()["kernel_launches"] - launches0) / args.reps,
                          "rotations_per_call": nkeys, "ms_per_rotation": elapsed / args.reps * 1e3 / nkeys,
                          "ms": elapsed / args.reps * 1e3, "host_enqueue_ms": enqueue / args.reps * 1e3, "ops_per_s": args.reps / elapsed, "backend": backend,
                          "verified_against_single_rank": ok}))
    vp.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
