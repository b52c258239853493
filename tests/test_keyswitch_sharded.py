"""The general-L streamed key-switch programs and their limb-sharded (multi-GPU) orchestration.

CPU tests: the compute engine is the oracle's ISA machine (tests may use the oracle as the
checker/stand-in; the product engine needs a GPU), the exchange runs over gloo with two ranks.
GPU test: the same orchestration on the real engine, single rank, N = 2^16.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from aloha_amd import isa, keyswitch, programs
from util import N, P, Q0, Q1, load, manifest

SPM, KSK = 0, 1


def _random_case(n, L, seed):
    logn = n.bit_length() - 1
    mods = [oracle.nth_prime(logn, i) for i in range(L)]
    special = oracle.nth_prime(logn, L)
    rng = np.random.default_rng(seed)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    ksk = np.stack([rng.integers(0, (mods + [special])[j], n, dtype=np.uint64)
                    for j in range(L + 1) for _ in range(2 * L)])
    return mods, special, ct, ksk


def _single_rank(vp, n, mods, special, ct, ksk, k):
    ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
    ks.load_key(ksk)
    ks.load_ct(ct)
    ks.rotate(k)
    return ks.read_out()


def test_streamed_program_reproduces_golden_rotate():
    """N = 8192, the reference's primes and key: the streamed L-limb program must give the
    reference's own rotate result (tv/case0_4_4 kernel1, step 2 -> k = 3^2)."""
    m = manifest()["case0"]
    ct = load(m["kernel1_ct_before_rotate"]).reshape(4, N)
    exp = load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N)
    vp = oracle.OracleVP()
    got = _single_rank(vp, N, [Q0, Q1], P, ct, load(m["ksk_step2"]), pow(3, 2, 2 * N))
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("reference_shape", [False, True])
def test_phase1_in_the_ntt_domain_equals_the_reference_shape(reference_shape):
    """Phase 1 with the automorphism applied to the NTT-form limbs (vautn + vautz; one transform per limb: the default)
    against the reference's own sequence (iNTT, vaut, NTT: keyswitch.mem:4-7,:24-27,:100-121).  Both reproduce the golden
    rotate, agree on random data -- and agree word for word on a ciphertext whose c1 is mostly ZERO coefficients, where
    vaut stores a flipped zero as q_d (vxu_lane.sv:598) and the basis extension lifts q_d, not 0, under the other moduli:
    vautz restores exactly those words (without it the two forms differ there, pinned by the identity test below)."""
    m = manifest()["case0"]
    ct = load(m["kernel1_ct_before_rotate"]).reshape(4, N)
    exp = load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N)
    ksk, k = load(m["ksk_step2"]), pow(3, 2, 2 * N)

    def rotate(c):
        ks = keyswitch.ShardedKeySwitch(oracle.OracleVP(), N, [Q0, Q1], P, reference_shape=reference_shape)
        ks.load_key(ksk); ks.load_ct(c); ks.rotate(k)
        return ks.read_out()

    assert np.array_equal(rotate(ct), exp)
    sparse = ct.copy()
    sparse[2:] = 0                                               # c1 = 0: every digit coefficient is a (flipped) zero
    sparse[2, :] = oracle.ntt(np.concatenate([np.arange(1, 9, dtype=np.uint64), np.zeros(N - 8, np.uint64)]), Q0)
    ref = keyswitch.ShardedKeySwitch(oracle.OracleVP(), N, [Q0, Q1], P, reference_shape=True)
    ref.load_key(ksk); ref.load_ct(sparse); ref.rotate(k)
    assert np.array_equal(rotate(sparse), ref.read_out())
    n, L = 1024, 3
    mods, special, rct, rksk = _random_case(n, L, 11)
    outs = []
    for shape in (True, reference_shape):
        ks = keyswitch.ShardedKeySwitch(oracle.OracleVP(spm_rows=4096, ksk_rows=4096), n, mods, special, reference_shape=shape)
        ks.load_key(rksk); ks.load_ct(rct); ks.rotate(pow(3, 5, 2 * n))
        outs.append(ks.read_out())
    assert np.array_equal(outs[0], outs[1])


def test_streamed_program_equals_register_program():
    """Own generators against each other on random data, L = 3 (both forms exist)."""
    n, L = 1024, 3
    mods, special, ct, ksk = _random_case(n, L, 5)
    k = pow(3, 5, 2 * n)
    got = _single_rank(oracle.OracleVP(spm_rows=4096, ksk_rows=4096), n, mods, special, ct, ksk, k)
    vp = oracle.OracleVP(spm_rows=4096, ksk_rows=4096)
    vp.load_isram(0, programs.assemble(programs.keyswitch(n, mods, special)))
    rows = n // 128
    vp.write(SPM, 0, ct.reshape(-1)); vp.write(KSK, 0, ksk.reshape(-1))
    assert vp.run(0, src0=0, rslt=2 * L * rows, ksk=0, step=k) == 0
    exp = vp.read(SPM, 2 * L * rows, 2 * L * n).reshape(2, L, n)
    assert np.array_equal(got, exp)


def _toy_case(n=1024, L=3, seed=7):
    from ckks_toy import ToyCKKS
    logn = n.bit_length() - 1
    mods = [oracle.nth_prime(logn, i) for i in range(L)]
    special = oracle.nth_prime(logn, L)
    return ToyCKKS(n, mods, special, seed)


def _max_err(got, exp):
    return max(abs(int(a) - int(b)) for a, b in zip(got, exp))


def test_rotate_means_automorphism_under_decryption():
    """What the rotate program MEANS: with a key that switches sigma_k(s) -> s in the reference's
    key format, decrypt(rotate(ct)) = sigma_k(m) up to key-switch noise.  Exact integers, N = 1024."""
    toy = _toy_case()
    n, k = toy.n, pow(3, 5, 2 * toy.n)
    rng = np.random.default_rng(3)
    m = [int(x) << 30 for x in rng.integers(-1000, 1000, n)]
    ct = toy.encrypt(m)
    ksk = toy.switch_key(toy.automorph(toy.s, k))
    vp = oracle.OracleVP(spm_rows=8192, ksk_rows=8192)
    ks = keyswitch.ShardedKeySwitch(vp, n, toy.moduli, toy.special)
    ks.load_key(ksk); ks.load_ct(ct.reshape(2 * toy.L, n))
    ks.rotate(k)
    got = toy.decrypt(ks.read_out())
    assert _max_err(got, toy.automorph(m, k)) < 1 << 20          # noise << 2^30 message scale


def test_multiply_relinearise_means_product_under_decryption():
    """BASELINE config 4's operation: decrypt(multiply(ct_a, ct_b)) = m_a * m_b (negacyclic) up to noise."""
    toy = _toy_case(seed=9)
    n = toy.n
    rng = np.random.default_rng(4)
    ma = [int(x) << 20 for x in rng.integers(-100, 100, n)]
    mb = [int(x) << 20 for x in rng.integers(-100, 100, n)]
    cta, ctb = toy.encrypt(ma), toy.encrypt(mb)
    ksk = toy.switch_key(toy.negacyclic(toy.s, toy.s))
    vp = oracle.OracleVP(spm_rows=8192, ksk_rows=8192)
    ks = keyswitch.ShardedKeySwitch(vp, n, toy.moduli, toy.special)
    ks.load_key(ksk); ks.load_ct(cta.reshape(2 * toy.L, n)); ks.load_ct_b(ctb.reshape(2 * toy.L, n))
    ks.multiply()
    got = toy.decrypt(ks.read_out())
    exp = toy.negacyclic(ma, mb)
    # product scale 2^40 * (sum of 1024 terms up to 10^4): noise from e*m cross terms ~ 2^20 * 100 * 1024 * few
    assert _max_err(got, exp) < 1 << 36 and max(abs(x) for x in exp) > 1 << 50


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, L, seed, k, out_dir, op="rotate", engine="oracle"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mods, special, ct, ksk = _random_case(n, L, seed)
        if engine == "gpu":                                     # both ranks on GPU 0: the exchange still goes through gloo
            from aloha_amd import vp as avp
            vp = avp.VP(spm_rows=1 << 15, ksk_rows=1 << 15, max_log_n=16, device=0)
        else:
            vp = oracle.OracleVP(spm_rows=8192, ksk_rows=8192)
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special, world=world, rank=rank,
                                        exchange=keyswitch.Exchange(group=None, tensor_of=None))
        ks.load_key(ksk); ks.load_ct(ct)
        if op == "multiply":
            ks.load_ct_b(ct[::-1].copy())
            ks.multiply()
        else:
            ks.rotate(k)
        np.save(os.path.join(out_dir, f"out{rank}.npy"), ks.read_out())
        np.save(os.path.join(out_dir, f"own{rank}.npy"), np.array(ks.lo.owned_ct))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("L", [4, 5])
def test_two_rank_sharded_keyswitch_gloo(tmp_path, L):
    """world_size 2 over gloo: limbs sharded j mod 2 (L = 5: uneven shards, P on rank 1),
    all-gather of digits + broadcast of the mod-down terms; result == single-rank result."""
    n, seed = 1024, 11
    k = pow(3, 7, 2 * n)
    mods, special, ct, ksk = _random_case(n, L, seed)
    exp = _single_rank(oracle.OracleVP(spm_rows=8192, ksk_rows=8192), n, mods, special, ct, ksk, k)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, L, seed, k, str(tmp_path)), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        out = np.load(tmp_path / f"out{r}.npy")
        own = np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, i], exp[:, j]), f"rank {r} limb {j}"
            seen.add(int(j))
    assert seen == set(range(L))


def test_two_rank_sharded_multiply_gloo(tmp_path):
    """The ciphertext product + relinearisation, limbs over two ranks: == the single-rank result."""
    n, L, seed = 1024, 5, 12
    mods, special, ct, ksk = _random_case(n, L, seed)
    vp = oracle.OracleVP(spm_rows=8192, ksk_rows=8192)
    ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
    ks.load_key(ksk); ks.load_ct(ct); ks.load_ct_b(ct[::-1].copy())
    ks.multiply()
    exp = ks.read_out()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, L, seed, 0, str(tmp_path), "multiply"), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        out = np.load(tmp_path / f"out{r}.npy")
        own = np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, i], exp[:, j]), f"rank {r} limb {j}"
            seen.add(int(j))
    assert seen == set(range(L))


def test_layout_partition_properties():
    for world in (1, 2, 4, 8):
        los = [keyswitch.Layout(1 << 16, [1] * 24, 2, world, r) for r in range(world)]
        owned = sorted(j for lo in los for j in lo.owned_all)
        assert owned == list(range(25))                       # every modulus exactly once
        slots = sorted(lo.dig_slot(d) - lo.dig for lo in los[:1] for d in range(24))
        assert len(set(slots)) == 24 and max(slots) < world * los[0].Dr
        for lo in los:
            assert lo.spm_rows_needed == lo.slots * 512


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["rotate", "multiply"])
def test_two_rank_sharded_keyswitch_real_engines(tmp_path, op):
    """Two processes, each with its own engine on the GPU, limbs sharded j mod 2 (uneven: L = 5, P on
    rank 1), exchanges over gloo: every owned limb equals the single-rank oracle result."""
    n, L, seed = 4096, 5, 31
    k = pow(3, 9, 2 * n)
    mods, special, ct, ksk = _random_case(n, L, seed)
    vp = oracle.OracleVP(spm_rows=1 << 15, ksk_rows=1 << 15)
    ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
    ks.load_key(ksk); ks.load_ct(ct)
    if op == "multiply":
        ks.load_ct_b(ct[::-1].copy()); ks.multiply()
    else:
        ks.rotate(k)
    exp = ks.read_out()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, L, seed, k, str(tmp_path), op, "gpu"), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        out = np.load(tmp_path / f"out{r}.npy")
        own = np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, i], exp[:, j]), f"rank {r} limb {j}"
            seen.add(int(j))
    assert seen == set(range(L))


@pytest.mark.gpu
def test_streamed_keyswitch_on_gpu_n65536():
    """BASELINE config 4/5 shape at reduced limb count: N = 2^16, L = 4 (+1), one GPU; and the
    golden rotate through the streamed program on the GPU."""
    from aloha_amd import vp as avp
    n, L = 1 << 16, 4
    mods, special, ct, ksk = _random_case(n, L, 21)
    k = pow(3, 3, 2 * n)
    lo = keyswitch.Layout(n, mods, special)
    exp = _single_rank(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods,
                       special, ct, ksk, k)
    def product(vp):
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
        ks.load_key(ksk); ks.load_ct(ct); ks.load_ct_b(ct[::-1].copy())
        ks.multiply()
        return ks.read_out()
    exp_mul = product(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows))
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        got = _single_rank(vp, n, mods, special, ct, ksk, k)
        assert np.array_equal(got, exp)
        assert np.array_equal(product(vp), exp_mul)             # ciphertext product + relinearisation (config 4)
        # the RCCL exchange path (zero-copy int64 views of engine memory) with a one-rank group
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            ks = keyswitch.ShardedKeySwitch(vp, n, mods, special, world=1, rank=0,
                                            exchange=keyswitch.Exchange(tensor_of=keyswitch.device_tensor, always=True))
            ks.load_key(ksk); ks.load_ct(ct)
            ks.rotate(k)
            assert np.array_equal(ks.read_out(), exp)
            # ... and stream-ordered: the engine runs on a torch stream, the collectives are enqueued on it
            # and the host never blocks between the phases of a key-switch
            st = torch.cuda.Stream()
            with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16,
                        stream=st.cuda_stream) as vps:
                ks = keyswitch.ShardedKeySwitch(vps, n, mods, special, world=1, rank=0,
                                                exchange=keyswitch.Exchange(tensor_of=keyswitch.device_tensor, always=True,
                                                                            stream=st))
                ks.load_key(ksk); ks.load_ct(ct)
                for _ in range(3):                                   # repeated: a missing ordering edge would show as a race
                    ks.load_ct(ct)
                    ks.rotate(k)
                assert np.array_equal(ks.read_out(), exp)
        finally:
            dist.destroy_process_group()
    m = manifest()["case0"]
    with avp.VP(spm_rows=1 << 14, ksk_rows=1 << 12, max_log_n=13) as vp:
        got = _single_rank(vp, N, [Q0, Q1], P, load(m["kernel1_ct_before_rotate"]).reshape(4, N), load(m["ksk_step2"]),
                           pow(3, 2, 2 * N))
    assert np.array_equal(got, load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N))


@pytest.mark.gpu
def test_baseline_config4_full_size_reference_algorithm_gpu_equals_oracle_machine():
    """BASELINE config 4 as SURVEY 8d states it: HMult + relinearisation at N = 2^16, L = 24 (+1 special prime),
    one digit per prime (the reference's decomposition, 1200 key limbs): GPU == the oracle's ISA machine on every
    output word.  (The oracle takes ~20 s for the 674 transforms.)"""
    from aloha_amd import vp as avp
    n, L = 1 << 16, 24
    mods = [oracle.nth_prime(16, i) for i in range(L)]
    special = oracle.nth_prime(16, L)
    lo = keyswitch.Layout(n, mods, special)
    rng = np.random.default_rng(24)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])

    def run(vp):
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
        for jl, j in enumerate(lo.owned_all):                    # key limb group by group (600 MiB in all)
            vp.write(keyswitch.KSK, jl * L * 2 * lo.rows,
                     np.random.default_rng(9000 + j).integers(0, lo.modulus(j), L * 2 * n, dtype=np.uint64))
        ks.load_ct(ct); ks.load_ct_b(ct[::-1].copy())
        ks.multiply()
        return ks.read_out()

    exp = run(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows))
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        got = run(vp)
    assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_baseline_config5_full_size_reference_algorithm_rotate_gpu_equals_oracle_machine():
    """BASELINE config 5 with the reference's own decomposition (keyswitch.mem:4-121 generalised: one digit per prime,
    one special prime): rotate + key-switch at N = 2^16, L = 32 (+1), 2112 key limbs, GPU == the oracle's ISA machine
    on every output word."""
    from aloha_amd import vp as avp
    n, L = 1 << 16, 32
    mods = [oracle.nth_prime(16, i) for i in range(L)]
    special = oracle.nth_prime(16, L)
    lo = keyswitch.Layout(n, mods, special)
    rng = np.random.default_rng(32)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    k = pow(3, 7, 2 * n)

    def run(vp):
        ks = keyswitch.ShardedKeySwitch(vp, n, mods, special)
        for jl, j in enumerate(lo.owned_all):                    # key limb group by group (1 GiB in all)
            vp.write(keyswitch.KSK, jl * L * 2 * lo.rows,
                     np.random.default_rng(9500 + j).integers(0, lo.modulus(j), L * 2 * n, dtype=np.uint64))
        ks.load_ct(ct)
        ks.rotate(k)
        return ks.read_out()

    exp = run(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows))
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        got = run(vp)
    assert np.array_equal(got, exp)


def test_vautz_restores_the_flipped_zeros_of_vaut():
    """The identity that pins the vautz extension: vautz(vintt(vautn(X, k)), k) == vaut(vintt(X), k) word for word, zeros at
    sign-flipped targets included (q there, vxu_lane.sv:598), for X = vntt of a polynomial with many zero coefficients --
    and without vautz the two sides differ exactly at those positions."""
    for logn, k in ((10, 5), (13, pow(3, 2, 1 << 14)), (13, pow(3, 8, 1 << 14) + (1 << 13))):
        n = 1 << logn
        q = Q0 if logn == 13 else oracle.nth_prime(logn, 0)
        rng = np.random.default_rng(logn + k)
        x = rng.integers(0, q, n, dtype=np.uint64)
        x[rng.random(n) < 0.5] = 0
        X = oracle.ntt(x, q)
        want = oracle.vaut(oracle.intt(X, q), q, k)
        mid = oracle.intt(oracle.vautn(X, k), q)
        assert np.array_equal(oracle.vautz(mid, q, k), want)
        diff = np.nonzero(mid != want)[0]
        assert len(diff) and np.all(mid[diff] == 0) and np.all(want[diff] == q)
