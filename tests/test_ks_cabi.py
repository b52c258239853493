"""The C host's key-switch (aloha_ks_*, csrc/keyswitch_host.cpp).  Its program generator is the only one: the Python driver
of aloha_amd/hybrid.py -- which runs the oracle's ISA machine in the parity tests -- fetches the same words through
aloha_ks_program.  On the CPU: the Python slot-layout mirror agrees with the C host, bad configurations are refused; on
the GPU: the C host's results equal the oracle machine's, with and without a (one-rank) RCCL communicator.  Reference step: the rotate / key-switch micro-program keyswitch.mem:1-122 and its driver run_rotate
(top_noaxilite_tb.sv:522-532)."""
import numpy as np
import pytest

import oracle
from aloha_amd import hybrid, isa, keyswitch, ks
from aloha_amd import vp as avp
from util import N, P, Q0, Q1, load, manifest


def _mnemonics(words):
    form = {0: ".vv", 1: ".vs", 2: ".sv"}
    return [i.name + (form[i.funct3] if i.name in ("vfqmul", "vfqadd", "vfqsub") else "") for i in isa.insts_of(words)]


def _setup(n, L, K):
    logn = n.bit_length() - 1
    return [oracle.nth_prime(logn, i) for i in range(L)], [oracle.nth_prime(logn, L + k) for k in range(K)]


@pytest.mark.parametrize("logn,L,K,alpha,world,batch", [
    (13, 2, 1, 1, 1, 1),       # the reference's own shape (keyswitch.mem)
    (10, 5, 2, 2, 1, 1), (10, 5, 2, 2, 2, 1), (10, 5, 2, 2, 3, 2),
    (12, 6, 2, 4, 1, 1),       # ragged last digit
    (16, 24, 4, 4, 8, 1),      # BASELINE config 4 on 8 ranks
    (16, 32, 4, 4, 8, 2), (16, 24, 1, 1, 1, 1),   # config 5; the reference's decomposition at L = 24 (programs split over the ROM)
])
def test_c_host_plan_and_the_python_layout_mirror_agree(logn, L, K, alpha, world, batch):
    """There is ONE generator of the phase programs, the C host's (aloha_ks_program); hybrid.py fetches its words and keeps
    only the slot layout for placing data.  That mirror must agree with the C host on everything a caller can observe:
    ownership, digit count, rows needed, the two gather buffers -- and the driver must hold exactly the plan's programs."""
    n = 1 << logn
    mods, specials = _setup(n, L, K)
    for rank in range(world):
        lo = hybrid.HybridLayout(n, mods, specials, alpha, world, rank, base_row=7, batch=batch)
        with ks.KeySwitch(None, logn, mods, specials, alpha, world, rank, batch, base_row=7) as k:
            assert (k.owned_ct, k.owned_all, k.dnum) == (lo.owned_ct, lo.owned_all, lo.dnum)
            assert ks.rows_needed(logn, mods, specials, alpha, world, rank, batch, 7) == (lo.spm_rows_needed, max(1, lo.ksk_limbs * lo.rows))
            assert k.gather_buffer(0) == (lo.row(lo.y), lo.Yr * lo.B * n) and k.gather_buffer(1) == (lo.row(lo.w), lo.Wr * lo.B * n)
            hs = hybrid.HybridKeySwitch(_RomOnly(), n, mods, specials, alpha, world=world, rank=rank, batch=batch, base_row=7)
            for phase in ks.PHASES:
                progs = hs._p[phase]
                for idx, words in enumerate(progs):
                    assert np.array_equal(k.program(phase, idx), words) and len(words) <= isa.IRAM_DEPTH
                assert k.program(phase, len(progs)).shape == (0, 3)
            # "mid" is cut by modulus: every program holds whole chains (sums, transforms, inner products of its moduli)
            for words in hs._p["mid"]:
                names = _mnemonics(words)
                assert "vntt" in names and "vfqmul.vv" in names and ("vfqmul.vs" in names or lo.dnum == 1)
                first_ntt, last_ntt = names.index("vntt"), len(names) - 1 - names[::-1].index("vntt")
                assert "vfqmul.vv" not in names[:last_ntt] and "vfqmul.vs" not in names[first_ntt:]
            # programs generated on demand: the decomposition of slot 0 only, and `batch` rotations in one pass
            steps = [pow(3, b + 2, 2 * n) for b in range(batch)]
            hs.vp.runs = 0
            hs.ex = keyswitch.Exchange(stub=True)
            hs.hoist_one(); hs.rotate_hoisted_many(steps, list(range(batch))[::-1])
            assert hs.vp.runs > 0


class _RomOnly:
    """Just enough of a VP for program generation."""
    runs = 0

    def load_isram(self, pc, words):
        assert pc + len(words) <= 4096

    def run(self, *a, **kw):
        self.runs += 1
        return 0


def test_a_deeper_rom_holds_a_whole_phase():
    """The reference's instruction ROM has 4096 words (vp_defines.vh:31); the mid phase of a key-switch at L = 32 (hybrid
    alpha = 4) needs ~6700.  With the default depth it is cut into two programs by modulus; a plan for a handle with a
    deeper ROM (aloha_vp_config.iram_depth) holds it in one, which the engine then runs as one fused launch pair."""
    mods, specials = _setup(1 << 16, 32, 4)
    with ks.KeySwitch(None, 16, mods, specials, 4) as k4, ks.KeySwitch(None, 16, mods, specials, 4, rom_depth=16384) as k16:
        two = [k4.program("mid", i) for i in range(3)]
        one = [k16.program("mid", i) for i in range(2)]
        assert [len(w) > 0 for w in two] == [True, True, False] and all(len(w) <= 4096 for w in two)
        assert len(one[0]) > 4096 and len(one[1]) == 0
        body = lambda w: [tuple(r) for r in w[1:-1]]             # without vsetvl / break
        assert body(two[0]) + body(two[1]) != body(one[0])       # (not a plain concatenation: each program holds whole chains ...)
        assert sorted(body(two[0]) + body(two[1])) == sorted(body(one[0]))     # ... of the same instructions
    with pytest.raises(avp.AlohaError):
        ks.KeySwitch(None, 16, mods, specials, 4, rom_depth=1000)
    with pytest.raises(avp.AlohaError):
        ks.KeySwitch(None, 16, mods, specials, 4, rom_depth=1 << 20)


def test_c_host_rejects_bad_configs():
    mods, specials = _setup(1024, 3, 1)
    for kw in (dict(alpha=0), dict(world=0), dict(rank=2, world=2), dict(batch=0)):
        args = dict(alpha=1, world=1, rank=0, batch=1)
        args.update(kw)
        with pytest.raises(avp.AlohaError):
            ks.KeySwitch(None, 10, mods, specials, **args)
    with pytest.raises(avp.AlohaError):
        ks.KeySwitch(None, 9, mods, specials, 1)
    with ks.KeySwitch(None, 10, mods, specials, 1) as plan:      # a plan cannot run
        with pytest.raises(avp.AlohaError):
            plan.rotate(3)


def _random(n, L, K, alpha, seed):
    mods, specials = _setup(n, L, K)
    rng = np.random.default_rng(seed)
    dnum = -(-L // alpha)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    key = np.stack([rng.integers(0, (mods + specials)[j], n, dtype=np.uint64) for j in range(L + K) for _ in range(2 * dnum)])
    return mods, specials, ct, key


@pytest.mark.gpu
def test_c_host_reproduces_the_golden_rotate():
    """alpha = K = 1 with the reference's primes and key: tv/case0_4_4 kernel1 (rotation step 2 -> k = 3^2)."""
    m = manifest()["case0"]
    ct = load(m["kernel1_ct_before_rotate"]).reshape(4, N)
    exp = load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N)
    spm, ksk = ks.rows_needed(13, [Q0, Q1], [P], 1)
    with avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=13) as vp, ks.KeySwitch(vp, 13, [Q0, Q1], [P], 1) as k:
        k.load_key(load(m["ksk_step2"])); k.load_ct(ct)
        k.rotate(pow(3, 2, 2 * N))
        assert np.array_equal(k.read_out(), exp)


@pytest.mark.gpu
@pytest.mark.parametrize("logn,L,K,alpha,batch", [(12, 5, 2, 2, 1), (16, 6, 2, 4, 2), (13, 4, 1, 1, 1)])
def test_c_host_on_gpu_equals_oracle_machine(logn, L, K, alpha, batch):
    n = 1 << logn
    mods, specials, ct, key = _random(n, L, K, alpha, 300 + L)
    cts = [np.roll(ct, b, axis=1) for b in range(batch)]
    kk = pow(3, 5, 2 * n)
    lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=batch)
    ref = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods, specials, alpha, batch=batch)
    ref.load_key(key)
    spm, ksk = ks.rows_needed(logn, mods, specials, alpha, batch=batch)
    with avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=logn) as vp, ks.KeySwitch(vp, logn, mods, specials, alpha, batch=batch) as k:
        k.load_key(key.reshape(-1))
        for b in range(batch):
            k.load_ct(cts[b], b); k.load_ct_b(cts[b][::-1].copy(), b)
            ref.load_ct(cts[b], b); ref.load_ct_b(cts[b][::-1].copy(), b)
        k.rotate(kk); ref.rotate(kk)
        for b in range(batch):
            assert np.array_equal(k.read_out(b), ref.read_out(b)), f"rotate slot {b}"
        k.multiply(); ref.multiply()
        for b in range(batch):
            assert np.array_equal(k.read_out(b), ref.read_out(b)), f"multiply slot {b}"


@pytest.mark.gpu
def test_c_host_with_a_one_rank_rccl_communicator():
    """The exchange path for real: ncclGetUniqueId / ncclCommInitRank through the C ABI, both all-gathers issued as
    ncclAllGather on the engine's stream (in place, one rank), repeated so that a missing ordering edge between a
    phase's kernels and the collective would show.  No torch.distributed anywhere."""
    n, logn, L, K, alpha = 1 << 13, 13, 6, 2, 2
    mods, specials, ct, key = _random(n, L, K, alpha, 77)
    lo = hybrid.HybridLayout(n, mods, specials, alpha)
    ref = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods, specials, alpha)
    ref.load_key(key); ref.load_ct(ct); ref.load_ct_b(ct[::-1].copy())
    ref.multiply()
    exp = ref.read_out()
    spm, ksk = ks.rows_needed(logn, mods, specials, alpha)
    with avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=logn) as vp, ks.KeySwitch(vp, logn, mods, specials, alpha) as k:
        k.comm_init(ks.nccl_unique_id())
        k.load_key(key.reshape(-1)); k.load_ct(ct); k.load_ct_b(ct[::-1].copy())
        for _ in range(3):
            k.multiply()
        assert np.array_equal(k.read_out(), exp)
        assert k.info()["bytes_sent"] == 0                       # one rank: nothing leaves the GPU, but the collectives ran


@pytest.mark.gpu
def test_c_host_hoisted_rotations_and_rescale_equal_oracle_machine():
    """One decomposition, three rotation steps (one at a time, and all in ONE pass of the phases with one key region per
    step), then the rescale of leveled CKKS: the C host on the GPU == hybrid.py on the oracle's ISA machine."""
    logn, n, L, K, alpha, R = 13, 1 << 13, 6, 2, 2, 3
    mods, specials, ct, key = _random(n, L, K, alpha, 91)
    keys = [np.roll(key, r + 1, axis=1) for r in range(R)]
    steps = [pow(3, r + 1, 2 * n) for r in range(R)]
    lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=R)
    ref = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=R * lo.ksk_limbs * lo.rows), n, mods, specials, alpha, batch=R)
    spm, ksk = ks.rows_needed(logn, mods, specials, alpha, batch=R)
    with avp.VP(spm_rows=spm, ksk_rows=R * ksk, max_log_n=logn) as vp, ks.KeySwitch(vp, logn, mods, specials, alpha, batch=R) as k:
        assert k.info()["key_rows"] == ref.key_rows == ksk
        for r in range(R):
            k.load_key_hoisted(keys[r].reshape(-1), steps[r], r); ref.load_key_hoisted(keys[r], steps[r], r)
        for b in range(R):
            c = np.roll(ct, b, axis=1)
            k.load_ct(c, b); ref.load_ct(c, b)
        k.hoist(); ref.hoist()
        for r in range(R):
            k.rotate_hoisted(steps[r], r); ref.rotate_hoisted(steps[r], r)
            for b in range(R):
                assert np.array_equal(k.read_out(b), ref.read_out(b)), f"hoisted step {r} slot {b}"
        k.hoist(first_slot_only=True); ref.hoist_one()
        k.rotate_hoisted_many(steps, [2, 0, 1]); ref.rotate_hoisted_many(steps, [2, 0, 1])
        for b in range(R):
            assert np.array_equal(k.read_out(b), ref.read_out(b)), f"one-pass rotation slot {b}"
        k.rescale(); ref.rescale()
        for b in range(R):
            assert np.array_equal(k.read_out(b)[:, :L - 1], ref.read_out(b)[:, :L - 1]), f"rescale slot {b}"


def _sparse_case():
    """The golden rotate's ciphertext with c1 replaced by a polynomial whose inverse transform is mostly ZERO: every zero
    at a sign-flipped position is stored as q_d by the reference's vaut (vxu_lane.sv:598) and lifted as q_d."""
    m = manifest()["case0"]
    ct = load(m["kernel1_ct_before_rotate"]).reshape(4, N).copy()
    ct[2:] = 0
    ct[2, :] = oracle.ntt(np.concatenate([np.arange(1, 9, dtype=np.uint64), np.zeros(N - 8, np.uint64)]), Q0)
    return ct, load(m["ksk_step2"]), pow(3, 2, 2 * N)


def test_generators_are_the_reference_algorithm_at_the_zero_corner():
    """alpha = K = 1: the generator's default phase 1 (vautn, one inverse transform, vautz) and its literal form of the
    reference's phase 1 (iNTT, vaut, NTT: ALOHA_KS_REFERENCE_PHASE1) == keyswitch.py's reference-shaped program == the
    semantics of keyswitch.mem, INCLUDING the flipped-zero corner (a crafted ciphertext whose c1 is mostly zeros).  Oracle
    machine."""
    from aloha_amd import keyswitch
    ct, ksk, k = _sparse_case()
    ref = keyswitch.ShardedKeySwitch(oracle.OracleVP(), N, [Q0, Q1], P, reference_shape=True)
    ref.load_key(ksk); ref.load_ct(ct); ref.rotate(k)
    exp = ref.read_out()
    for flag in (False, True):
        lo = hybrid.HybridLayout(N, [Q0, Q1], [P], 1)
        hs = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), N, [Q0, Q1], [P], 1,
                                    reference_phase1=flag)
        hs.load_key(ksk); hs.load_ct(ct); hs.rotate(k)
        assert np.array_equal(hs.read_out(), exp), flag
    with ks.KeySwitch(None, 13, [Q0, Q1], [P], 1, flags=ks.REFERENCE_PHASE1) as plan:
        names = _mnemonics(plan.program("rot1", 0))
        assert names.count("vaut") == 4 and names.count("vintt") == 4 and "vautn" not in names   # keyswitch.mem:4-7,:24-27 per limb
    with ks.KeySwitch(None, 13, [Q0, Q1], [P], 1) as plan:
        names = _mnemonics(plan.program("rot1", 0))
        assert names.count("vautn") == 4 and names.count("vintt") == 2 and names.count("vautz") == 2 and "vaut" not in names
    with pytest.raises(avp.AlohaError):
        ks.KeySwitch(None, 13, [Q0, Q1, oracle.nth_prime(13, 5)], [P], 2, flags=ks.REFERENCE_PHASE1)   # alpha = 1 only


@pytest.mark.gpu
def test_c_host_on_gpu_matches_the_reference_algorithm_at_the_zero_corner():
    """Both phase-1 forms of the C host on the GPU at the flipped-zero corner (the default folds vautz into the inverse
    transform's last pass at N = 2^16 and runs it as a pass of its own at this N = 2^13), and the golden rotate."""
    from aloha_amd import keyswitch
    ct, ksk, k = _sparse_case()
    ref = keyswitch.ShardedKeySwitch(oracle.OracleVP(), N, [Q0, Q1], P, reference_shape=True)
    ref.load_key(ksk); ref.load_ct(ct); ref.rotate(k)
    spm, kr = ks.rows_needed(13, [Q0, Q1], [P], 1)
    for flags in (0, ks.REFERENCE_PHASE1):
        with avp.VP(spm_rows=spm, ksk_rows=kr, max_log_n=13) as vp, ks.KeySwitch(vp, 13, [Q0, Q1], [P], 1, flags=flags) as kk:
            kk.load_key(ksk); kk.load_ct(ct); kk.rotate(k)
            assert np.array_equal(kk.read_out(), ref.read_out()), flags
            m = manifest()["case0"]                              # and the golden rotate, of course
            kk.load_ct(load(m["kernel1_ct_before_rotate"]).reshape(4, N)); kk.rotate(k)
            assert np.array_equal(kk.read_out(), load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N)), flags


def _c_host_rank(rank, world, port, logn, L, K, alpha, seed, op, k, out_dir):
    """One rank of a sharded key-switch run by the C host, the two all-gathers carried by gloo between the stages."""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << logn
        mods, specials, ct, key = _random(n, L, K, alpha, seed)
        spm, ksk = ks.rows_needed(logn, mods, specials, alpha, world, rank)
        with avp.VP(spm_rows=spm, ksk_rows=ksk, max_log_n=logn, device=0) as vp, ks.KeySwitch(vp, logn, mods, specials, alpha, world, rank) as kk:
            kk.load_key(key.reshape(-1)); kk.load_ct(ct); kk.load_ct_b(ct[::-1].copy())

            def exchange(which):
                row0, chunk = kk.gather_buffer(which)
                vp.sync()
                mine = torch.from_numpy(vp.read(0, row0 + rank * chunk // 128, chunk).view(np.int64).copy())
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                for r, part in enumerate(parts):
                    if r != rank:
                        vp.write(0, row0 + r * chunk // 128, part.numpy().view(np.uint64))

            kk.stage(op, 0, k); exchange(0)
            kk.stage(op, 1); exchange(1)
            kk.stage(op, 2)
            np.save(os.path.join(out_dir, f"out{rank}.npy"), kk.read_out())
            np.save(os.path.join(out_dir, f"own{rank}.npy"), np.array(kk.owned_ct))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,op", [(2, "rotate"), (3, "multiply")])
def test_c_host_sharded_over_ranks_with_the_exchanges_carried_by_the_host(tmp_path, world, op):
    """The C host's sharded layout end to end: `world` processes (sharing the one GPU of the box), each running its
    rank's stages through the C ABI, the digit and mod-down all-gathers carried between the stages by gloo (an RCCL
    communicator needs one GPU per rank).  L = 5, K = 2, alpha = 2: uneven shards, a special prime on two ranks.  Every
    rank's owned limbs == the single-rank oracle machine."""
    import socket
    import torch.multiprocessing as mp
    logn, L, K, alpha, seed = 12, 5, 2, 2, 23
    n = 1 << logn
    k = pow(3, 5, 2 * n)
    mods, specials, ct, key = _random(n, L, K, alpha, seed)
    lo = hybrid.HybridLayout(n, mods, specials, alpha)
    ref = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods, specials, alpha)
    ref.load_key(key); ref.load_ct(ct); ref.load_ct_b(ct[::-1].copy())
    ref.rotate(k) if op == "rotate" else ref.multiply()
    exp = ref.read_out()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_c_host_rank, args=(world, port, logn, L, K, alpha, seed, op, k, str(tmp_path)), nprocs=world, join=True)
    seen = set()
    for r in range(world):
        out, own = np.load(tmp_path / f"out{r}.npy"), np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, i], exp[:, j]), f"rank {r} limb {j}"
            seen.add(int(j))
    assert seen == set(range(L))


def test_c_host_rejects_bad_moduli():
    """A zero, even, composite, out-of-range, repeated or rootless modulus is refused by aloha_ks_create /
    aloha_ks_rows_needed with a status (a zero modulus used to end the process with SIGFPE in the layout's `1 % m`)."""
    good = [oracle.nth_prime(13, i) for i in range(3)]
    sp = [oracle.nth_prime(13, 3)]
    for mods, specials, status in [
            ([good[0], 0], sp, avp.ERR_MODULUS), ([good[0], good[1] + 1], sp, avp.ERR_MODULUS),       # zero; even
            ([good[0], good[0]], sp, avp.ERR_MODULUS), ([good[0], good[1]], [good[1]], avp.ERR_MODULUS),   # repeated (also across the two lists)
            ([good[0], (1 << 61) + 1], sp, avp.ERR_MODULUS), ([good[0], good[1] - 2], sp, avp.ERR_MODULUS),  # > 60 bits; composite or not 1 mod 2N
            ([good[0], P], sp, None)]:                                                                 # P = 1 mod 2^14 only: fine at N = 2^13 ...
        if status is None:
            with ks.KeySwitch(None, 13, mods, specials, 1):
                pass
            with pytest.raises(avp.AlohaError) as e:
                ks.KeySwitch(None, 14, mods, specials, 1)                                              # ... no 2N-th root at N = 2^14
            assert e.value.status == avp.ERR_ROOT
            continue
        with pytest.raises(avp.AlohaError) as e:
            ks.KeySwitch(None, 13, mods, specials, 1)
        assert e.value.status == status, (mods, e.value.status)
        with pytest.raises(avp.AlohaError):
            ks.rows_needed(13, mods, specials, 1)
