"""Hybrid key-switch (several primes per digit, NTT-domain automorphism).

Pinning: at alpha = 1, K = 1 the generalisation is the reference's own algorithm and must give
the reference's golden rotate bit for bit.  For alpha > 1 there are no reference vectors: the
programs are checked for what they mean (exact-integer decryption) on the oracle's ISA machine,
sharded vs single rank over gloo, and GPU vs oracle bit-exactly."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from aloha_amd import hybrid, keyswitch
from util import N, P, Q0, Q1, load, manifest


def _setup(n, L, K, seed):
    logn = n.bit_length() - 1
    mods = [oracle.nth_prime(logn, i) for i in range(L)]
    specials = [oracle.nth_prime(logn, L + k) for k in range(K)]
    return mods, specials


def _random_case(n, L, K, alpha, seed):
    mods, specials = _setup(n, L, K, seed)
    rng = np.random.default_rng(seed)
    dnum = -(-L // alpha)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    key = np.stack([rng.integers(0, (mods + specials)[j], n, dtype=np.uint64) for j in range(L + K) for _ in range(2 * dnum)])
    return mods, specials, ct, key


def _run(vp, n, mods, specials, alpha, ct, key, op, k=0, **kw):
    if op == "many":                                            # two rotation slots; read_out() then returns both results
        kw = dict(kw, batch=2)
    hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, **kw)
    hs.load_key(key); hs.load_ct(ct)
    if op == "multiply":
        hs.load_ct_b(ct[::-1].copy()); hs.multiply()
    elif op == "rescale":
        hs.rescale()
    elif op == "hoisted":                                       # the same key bits serve as "the key of step k" and of step 3k
        hs.load_key_hoisted(key, k, 0); hs.load_key_hoisted(key, 3 * k % (2 * n), 1)
        hs.hoist(); hs.rotate_hoisted(k, 0); hs.rotate_hoisted(3 * k % (2 * n), 1)
    elif op == "many":                                          # (needs batch >= 2: slot 1 = the second rotation)
        hs.load_key_hoisted(key, k, 0); hs.load_key_hoisted(key, 3 * k % (2 * n), 1)
        hs.hoist_one(); hs.rotate_hoisted_many([k, 3 * k % (2 * n)])
        one = hs.read_out
        hs.read_out = lambda: np.concatenate([one(0), one(1)])
    else:
        hs.rotate(k)
    return hs


@pytest.mark.parametrize("case,kern,step", [("case0", "kernel1", 2), ("case1", "kernel12", 4)])
def test_alpha1_reproduces_the_reference_golden_rotate(case, kern, step):
    """alpha = 1, K = 1 on the reference's primes and keys: == tv/ ct_after_rotate, bit for bit
    (vautn + iNTT in place of iNTT + vaut, scalar-1 basis conversions in place of vcp/vfqmod)."""
    m = manifest()[case]
    ct = load(m[f"{kern}_ct_before_rotate"]).reshape(4, N)
    exp = load(m[f"{kern}_ct_after_rotate"]).reshape(2, 2, N)
    hs = _run(oracle.OracleVP(), N, [Q0, Q1], [P], 1, ct, load(m[f"ksk_step{step}"]), "rotate", pow(3, step, 2 * N))
    assert np.array_equal(hs.read_out(), exp)


def _toy(n, L, K, seed):
    from ckks_toy import ToyCKKS
    mods, specials = _setup(n, L, K, seed)
    return ToyCKKS(n, mods, specials[0], seed), specials


@pytest.mark.parametrize("alpha,K", [(2, 2), (3, 3)])   # K >= alpha: P must exceed every digit modulus Q_d
def test_hybrid_rotate_and_multiply_decrypt_correctly(alpha, K):
    """Exact-integer semantics, N = 1024, L = 4 (alpha = 3: a ragged last digit)."""
    n, L = 1024, 4
    toy, specials = _toy(n, L, K, 13)
    rng = np.random.default_rng(8)
    k = pow(3, 7, 2 * n)
    m = [int(x) << 30 for x in rng.integers(-1000, 1000, n)]
    ct = toy.encrypt(m)
    key = toy.switch_key_hybrid(toy.automorph(toy.s, k), specials, alpha)
    hs = _run(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 14), n, toy.moduli, specials, alpha, ct.reshape(2 * L, n), key, "rotate", k)
    got = toy.decrypt(hs.read_out())
    exp = toy.automorph(m, k)
    assert max(abs(a - b) for a, b in zip(got, exp)) < 1 << 22
    ma = [int(x) << 20 for x in rng.integers(-100, 100, n)]
    mb = [int(x) << 20 for x in rng.integers(-100, 100, n)]
    cta, ctb = toy.encrypt(ma), toy.encrypt(mb)
    key = toy.switch_key_hybrid(toy.negacyclic(toy.s, toy.s), specials, alpha)
    hs = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 14), n, toy.moduli, specials, alpha)
    hs.load_key(key); hs.load_ct(cta.reshape(2 * L, n)); hs.load_ct_b(ctb.reshape(2 * L, n))
    hs.multiply()
    got = toy.decrypt(hs.read_out())
    exp = toy.negacyclic(ma, mb)
    assert max(abs(a - b) for a, b in zip(got, exp)) < 1 << 36 and max(abs(x) for x in exp) > 1 << 50


def test_key_stored_under_inverse_automorphism_moves_sigma_out_of_the_inner_product():
    """The identity hoisting rests on, on stored words: vautn_k(a) (.) b == vautn_k(a (.) gather(b, k^-1)) -- in NTT
    form the automorphism is a permutation of positions, so it commutes with the pointwise product exactly."""
    n = 1024
    q = oracle.nth_prime(10, 0)
    rng = np.random.default_rng(4)
    a, b = rng.integers(0, q, n, dtype=np.uint64), rng.integers(0, q, n, dtype=np.uint64)
    for k in (3, pow(3, 5, 2 * n), 2 * n - 1):
        if k >= n:                                               # the instruction keeps log2 n bits of the step
            continue
        src_inv = hybrid.ntt_automorphism_gather(n, pow(k, -1, 2 * n))
        assert np.array_equal(oracle.vautn(a, k), a[hybrid.ntt_automorphism_gather(n, k)])
        mul = lambda x, y: np.array([int(u) * int(v) % q for u, v in zip(x, y)], dtype=np.uint64)
        assert np.array_equal(mul(oracle.vautn(a, k), b), oracle.vautn(mul(a, b[src_inv]), k))


def _hoisted(vp, n, mods, specials, alpha, ct, keys, steps, **kw):
    """hoist() once, then every step with its own pre-permuted key region; returns the outputs in order."""
    hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, **kw)
    for idx, (k, key) in enumerate(zip(steps, keys)):
        hs.load_key_hoisted(key, k, idx)
    hs.load_ct(ct)
    hs.hoist()
    outs = []
    for idx, k in enumerate(steps):
        hs.rotate_hoisted(k, idx)
        outs.append(hs.read_out())
    return outs


@pytest.mark.parametrize("alpha,K", [(2, 2), (3, 3)])
def test_hoisted_rotations_decrypt_correctly(alpha, K):
    """One decomposition, three rotation steps, each with its own key: every output decrypts to the rotated
    message (exact integers).  Not bit-identical to rotate(): the fast basis conversion does not commute with
    the sign flips of the automorphism (multiples of Q_d the key cancels), which is why this is a semantic test."""
    n, L = 1024, 4
    toy, specials = _toy(n, L, K, 17)
    rng = np.random.default_rng(9)
    m = [int(x) << 30 for x in rng.integers(-1000, 1000, n)]
    ct = toy.encrypt(m).reshape(2 * L, n)
    steps = [pow(3, r, 2 * n) for r in (1, 4, 6)]
    assert all(k < n for k in steps)
    keys = [toy.switch_key_hybrid(toy.automorph(toy.s, k), specials, alpha) for k in steps]
    outs = _hoisted(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 15), n, toy.moduli, specials, alpha, ct, keys, steps)
    for k, out in zip(steps, outs):
        got, exp = toy.decrypt(out), toy.automorph(m, k)
        assert max(abs(a - b) for a, b in zip(got, exp)) < 1 << 22


def test_hoisted_rotations_in_one_pass_equal_one_by_one():
    """rotate_hoisted_many: the batch slots as rotation slots, steps in the instruction immediates, keys by region --
    word for word the outputs of separate rotate_hoisted calls (also with the regions permuted)."""
    n, L, K, alpha = 1024, 4, 2, 2
    mods, specials, ct, key = _random_case(n, L, K, alpha, 5)
    steps = [pow(3, r, 2 * n) for r in (1, 4, 6)]
    keys = [np.roll(key, r, axis=1) for r in range(3)]
    one = _hoisted(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 15), n, mods, specials, alpha, ct, keys, steps)
    hs = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=1 << 15, ksk_rows=1 << 15), n, mods, specials, alpha, batch=4)
    for i, (k, kk) in enumerate(zip(steps, keys)):
        hs.load_key_hoisted(kk, k, i)
    hs.load_ct(ct); hs.hoist_one(); hs.rotate_hoisted_many(steps)
    assert all(np.array_equal(hs.read_out(b), one[b]) for b in range(3))
    hs.rotate_hoisted_many(steps[::-1], [2, 1, 0])
    assert all(np.array_equal(hs.read_out(b), one[2 - b]) for b in range(3))


def test_rescale_divides_by_the_last_prime_under_decryption():
    """The leveled-CKKS step after a product: decrypt(rescale(ct)) under q_0..q_{L-2} = round(m / q_{L-1}) up to
    the rounding noise (a few units times the secret's weight)."""
    from ckks_toy import ToyCKKS
    n, L, K = 1024, 4, 2
    toy, specials = _toy(n, L, K, 31)
    rng = np.random.default_rng(2)
    qL = toy.moduli[-1]
    m = [int(x) * qL + int(y) for x, y in zip(rng.integers(-2 ** 30, 2 ** 30, n), rng.integers(0, 1000, n))]
    ct = toy.encrypt(m)
    hs = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 14), n, toy.moduli, specials, 2)
    hs.load_ct(ct.reshape(2 * L, n))
    hs.rescale()
    low = ToyCKKS(n, toy.moduli[:-1], specials[0], 31)
    low.s = toy.s
    got = low.decrypt(hs.read_out()[:, :L - 1])
    assert max(abs(a - round(x / qL)) for a, x in zip(got, m)) < 64


class _RomOnly:
    """Engine stand-in that only accepts instruction images (program generation needs nothing else)."""
    def __init__(self):
        self.images = []

    def load_isram(self, pc, words):
        assert pc + len(words) <= 4096
        self.images.append((pc, len(words)))


@pytest.mark.parametrize("L,alpha,world", [(24, 4, 1), (32, 4, 1), (24, 8, 1), (24, 4, 8), (32, 4, 8), (24, 1, 1), (13, 3, 4)])
def test_programs_fit_the_rom_and_layouts_partition(L, alpha, world):
    """BASELINE configs 4/5 shapes: every generated program fits the 4096-word instruction ROM, every
    modulus and gather slot is owned exactly once, and the transform count is dnum*(L+K) - L + 3L + 2K."""
    from aloha_amd import isa
    n, K = 1 << 16, alpha
    mods = [oracle.nth_prime(16, i) for i in range(L)]
    specials = [oracle.nth_prime(16, L + k) for k in range(K)]
    owned, yslots, wslots, ntt, intt = [], set(), set(), 0, 0
    for r in range(world):
        hs = hybrid.HybridKeySwitch(_RomOnly(), n, mods, specials, alpha, world=world, rank=r)
        lo = hs.lo
        owned += lo.owned_all
        for name, progs in hs._p.items():
            for w in progs:
                assert len(w) <= isa.IRAM_DEPTH
                if name in ("rot1", "mid", "terms", "fin_rot"):
                    for ins in isa.insts_of(w):
                        ntt += ins.name == "vntt"
                        intt += ins.name == "vintt"
        for i in lo.owned_ct:
            yslots.add(lo.y_slot(i) - lo.y)
        for k in lo.owned_sp:
            wslots.update({lo.w_slot(k, 0) - lo.w, lo.w_slot(k, 1) - lo.w})
        assert lo.spm_rows_needed == lo.slots * 512
    dnum = -(-L // alpha)
    assert sorted(owned) == list(range(L + K))
    assert len(yslots) == L and max(yslots) < world * -(-L // world)
    assert len(wslots) == 2 * K
    assert intt == L + 2 * K and ntt == dnum * (L + K) - L + 2 * L


def _run_batch(vp, n, mods, specials, alpha, cts, key, op, k=0, **kw):
    hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, batch=len(cts), **kw)
    hs.load_key(key)
    for b, ct in enumerate(cts):
        hs.load_ct(ct, b)
        if op == "multiply":
            hs.load_ct_b(cts[(b + 1) % len(cts)], b)
    hs.multiply() if op == "multiply" else hs.rotate(k)
    return hs


@pytest.mark.parametrize("op", ["rotate", "multiply"])
def test_batched_switch_equals_one_ciphertext_at_a_time(op):
    """batch = 3 ciphertexts under one key in one call == three separate calls (oracle machine, ragged digits)."""
    n, L, K, alpha, k = 1024, 5, 2, 2, pow(3, 9, 2048)
    mods, specials, ct, key = _random_case(n, L, K, alpha, 23)
    rng = np.random.default_rng(5)
    cts = [ct] + [np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)]) for _ in range(2)]
    hs = _run_batch(oracle.OracleVP(spm_rows=1 << 15, ksk_rows=1 << 14), n, mods, specials, alpha, cts, key, op, k)
    for b in range(3):
        one = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 14), n, mods, specials, alpha)
        one.load_key(key); one.load_ct(cts[b])
        if op == "multiply":
            one.load_ct_b(cts[(b + 1) % 3]); one.multiply()
        else:
            one.rotate(k)
        assert np.array_equal(hs.read_out(b), one.read_out()), (op, b)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, L, K, alpha, seed, k, op, out_dir, engine, batch=1, p2p=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mods, specials, ct, key = _random_case(n, L, K, alpha, seed)
        if engine == "gpu":
            from aloha_amd import vp as avp
            vp = avp.VP(spm_rows=1 << 15, ksk_rows=1 << 15, max_log_n=16, device=0)
        else:
            vp = oracle.OracleVP(spm_rows=1 << 14, ksk_rows=1 << 14)
        if batch == 1:
            hs = _run(vp, n, mods, specials, alpha, ct, key, op, k, world=world, rank=rank,
                      exchange=keyswitch.Exchange(group=None, tensor_of=None, p2p=p2p))
            np.save(os.path.join(out_dir, f"out{rank}.npy"), hs.read_out())
        else:
            cts = [np.roll(ct, b, axis=1) for b in range(batch)]
            hs = _run_batch(vp, n, mods, specials, alpha, cts, key, op, k, world=world, rank=rank,
                            exchange=keyswitch.Exchange(group=None, tensor_of=None, p2p=p2p))
            np.save(os.path.join(out_dir, f"out{rank}.npy"), np.stack([hs.read_out(b) for b in range(batch)]))
        np.save(os.path.join(out_dir, f"own{rank}.npy"), np.array(hs.lo.owned_ct))
    finally:
        dist.destroy_process_group()


def _check_sharded(tmp_path, n, L, K, alpha, op, engine, world=2, p2p=False):
    seed, k = 17, pow(3, 5, 2 * n)
    mods, specials, ct, key = _random_case(n, L, K, alpha, seed)
    exp = _run(oracle.OracleVP(spm_rows=1 << 15, ksk_rows=1 << 15), n, mods, specials, alpha, ct, key, op, k).read_out()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, L, K, alpha, seed, k, op, str(tmp_path), engine, 1, p2p), nprocs=world, join=True)
    seen = set()
    for r in range(world):
        out = np.load(tmp_path / f"out{r}.npy")
        own = np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, i], exp[:, j]), f"rank {r} limb {j}"
            seen.add(int(j))
    assert seen == set(range(L))


@pytest.mark.parametrize("op", ["rotate", "rescale"])
def test_three_rank_hybrid_gloo_point_to_point_exchange(tmp_path, op):
    """The all-gathers as world-1 direct sends and receives per rank (one per xGMI link on the node) instead of the
    collective: three ranks, uneven shards, same words."""
    _check_sharded(tmp_path, 1024, 5, 2, 2, op, "oracle", world=3, p2p=True)


@pytest.mark.parametrize("op", ["rotate", "multiply", "rescale", "hoisted", "many"])
def test_two_rank_hybrid_gloo(tmp_path, op):
    """L = 5, K = 2, alpha = 2 over two ranks (uneven shards, a special prime on each rank)."""
    _check_sharded(tmp_path, 1024, 5, 2, 2, op, "oracle")


def test_two_rank_batched_hybrid_gloo(tmp_path):
    """Two ranks, two ciphertexts per call: the gather buffers carry rank-major, ciphertext-minor chunks."""
    n, L, K, alpha, seed, k, op = 1024, 5, 2, 2, 29, pow(3, 5, 2048), "multiply"
    mods, specials, ct, key = _random_case(n, L, K, alpha, seed)
    cts = [np.roll(ct, b, axis=1) for b in range(2)]
    ref = _run_batch(oracle.OracleVP(spm_rows=1 << 15, ksk_rows=1 << 14), n, mods, specials, alpha, cts, key, op, k)
    exp = np.stack([ref.read_out(b) for b in range(2)])
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, L, K, alpha, seed, k, op, str(tmp_path), "oracle", 2), nprocs=2, join=True)
    for r in range(2):
        out = np.load(tmp_path / f"out{r}.npy")
        own = np.load(tmp_path / f"own{r}.npy")
        for i, j in enumerate(own):
            assert np.array_equal(out[:, :, i], exp[:, :, j]), f"rank {r} limb {j}"


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["rotate", "multiply", "rescale", "hoisted", "many"])
def test_two_rank_hybrid_real_engines(tmp_path, op):
    _check_sharded(tmp_path, 4096, 5, 2, 2, op, "gpu")


@pytest.mark.gpu
def test_hybrid_on_gpu_matches_oracle_and_golden():
    from aloha_amd import vp as avp
    # the golden rotate through the hybrid programs at alpha = 1, on the GPU
    m = manifest()["case0"]
    ct = load(m["kernel1_ct_before_rotate"]).reshape(4, N)
    exp = load(m["kernel1_ct_after_rotate"]).reshape(2, 2, N)
    with avp.VP(spm_rows=1 << 16, ksk_rows=1 << 16, max_log_n=16) as vp:
        hs = _run(vp, N, [Q0, Q1], [P], 1, ct, load(m["ksk_step2"]), "rotate", pow(3, 2, 2 * N))
        assert np.array_equal(hs.read_out(), exp)
    # N = 2^16, L = 6, K = 2, alpha = 2 and 4 (ragged): GPU == oracle machine, rotate and multiply
    n, L, K = 1 << 16, 6, 2
    for alpha in (2, 4):
        mods, specials, ct, key = _random_case(n, L, K, alpha, 40 + alpha)
        lo = hybrid.HybridLayout(n, mods, specials, alpha)
        for op in ("rotate", "multiply", "rescale", "hoisted", "many"):
            k = pow(3, 3, 2 * n)
            if op == "many":
                lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=2)
            exp = _run(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=2 * lo.ksk_limbs * lo.rows), n, mods, specials, alpha,
                       ct, key, op, k).read_out()
            with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=2 * lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
                got = _run(vp, n, mods, specials, alpha, ct, key, op, k).read_out()
            assert np.array_equal(got, exp), (alpha, op)
    # two ciphertexts per call on the GPU == the oracle machine
    cts = [ct, np.roll(ct, 1, axis=1)]
    lo = hybrid.HybridLayout(n, mods, specials, 4, batch=2)
    ref = _run_batch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods, specials, 4, cts, key,
                     "multiply")
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        got = _run_batch(vp, n, mods, specials, 4, cts, key, "multiply")
        for b in range(2):
            assert np.array_equal(got.read_out(b), ref.read_out(b)), b


@pytest.mark.gpu
@pytest.mark.parametrize("L,op,rom", [(24, "multiply", 4096), (32, "rotate", 4096), (32, "rotate", 16384)])
def test_baseline_configs_4_and_5_at_full_size_gpu_equals_oracle_machine(L, op, rom):
    """BASELINE config 4 (HMult + relinearisation, N = 2^16, 24 limbs) and config 5 (rotate + key-switch, 32 limbs)
    at their full sizes, hybrid alpha = 4, K = 4: every owned output word of the GPU == the oracle's ISA machine
    running the same programs (the oracle needs a few seconds per key-switch at this size)."""
    from aloha_amd import vp as avp
    n, K, alpha = 1 << 16, 4, 4
    mods, specials = _setup(n, L, K, 0)
    lo = hybrid.HybridLayout(n, mods, specials, alpha)
    rng = np.random.default_rng(100 + L)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    k = pow(3, 5, 2 * n)

    def run(vp):
        hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha)
        for jl, j in enumerate(lo.owned_all):                    # key limb by limb (a whole key is 168-288 MiB)
            vp.write(keyswitch.KSK, jl * lo.dnum * 2 * lo.rows,
                     np.random.default_rng(7000 + j).integers(0, lo.mods_all[j], lo.dnum * 2 * n, dtype=np.uint64))
        hs.load_ct(ct)
        if op == "multiply":
            hs.load_ct_b(ct[::-1].copy()); hs.multiply()
        else:
            hs.rotate(k)
        return hs.read_out()

    # rom = 4096: the reference's instruction ROM -- the mid phase of L = 32 is then two programs (cut by modulus);
    # rom = 16384: engine and oracle machine created with a deeper ROM, the whole phase is one program (one fused launch pair)
    exp = run(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, iram_depth=rom))
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=16, iram_depth=rom) as vp:
        got = run(vp)
        assert vp.config()["iram_depth"] == rom
    assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_replayed_phases_on_new_data_and_new_steps():
    """One handle, the same phase programs run again and again: the second and later calls replay recorded launch
    lists.  New ciphertexts, another rotation step (a different CSR: a different list), a product in between --
    every result equals the oracle machine's, which sequences every call from scratch."""
    from aloha_amd import vp as avp
    n, L, K, alpha = 1 << 13, 6, 2, 2
    mods, specials, ct, key = _random_case(n, L, K, alpha, 71)
    lo = hybrid.HybridLayout(n, mods, specials, alpha)
    rng = np.random.default_rng(72)
    cts = [ct] + [np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)]) for _ in range(3)]
    steps = [pow(3, 2, 2 * n), pow(3, 2, 2 * n), pow(3, 7, 2 * n), pow(3, 2, 2 * n)]
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows, max_log_n=13) as vp:
        g = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha)
        o = hybrid.HybridKeySwitch(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows), n, mods, specials, alpha)
        for hs in (g, o):
            hs.load_key(key)
        for i, (c, k) in enumerate(zip(cts, steps)):
            for hs in (g, o):
                hs.load_ct(c); hs.rotate(k)
            assert np.array_equal(g.read_out(), o.read_out()), f"rotate #{i}"
            for hs in (g, o):
                hs.load_ct_b(cts[(i + 1) % len(cts)]); hs.multiply()
            assert np.array_equal(g.read_out(), o.read_out()), f"multiply #{i}"
            for hs in (g, o):
                hs.rescale()
            assert np.array_equal(g.read_out()[:, :L - 1], o.read_out()[:, :L - 1]), f"rescale #{i}"


@pytest.mark.gpu
def test_baseline_config5_rotation_batch_at_full_size_gpu_equals_oracle_machine():
    """BASELINE config 5 is a BATCH of rotations: N = 2^16, L = 32 (+4 special), hybrid alpha = 4, three rotation
    steps of one decomposed ciphertext in ONE pass of the phases (``hoist_one`` + ``rotate_hoisted_many``, each step
    with its own key region): every output word of every rotation slot on the GPU == the oracle's ISA machine
    running the same programs.  Reference step it widens: keyswitch.mem:4-121, run once per rotation."""
    from aloha_amd import vp as avp
    n, L, K, alpha, R = 1 << 16, 32, 4, 4, 3
    mods, specials = _setup(n, L, K, 0)
    lo = hybrid.HybridLayout(n, mods, specials, alpha, batch=R)
    rng = np.random.default_rng(532)
    ct = np.stack([rng.integers(0, mods[j % L], n, dtype=np.uint64) for j in range(2 * L)])
    steps = [pow(3, r + 1, 2 * n) for r in range(R)]

    def run(vp):
        hs = hybrid.HybridKeySwitch(vp, n, mods, specials, alpha, batch=R)
        for r in range(R):                                        # one key region per rotation; any words serve as a key
            for jl, j in enumerate(lo.owned_all):
                vp.write(keyswitch.KSK, r * hs.key_rows + jl * lo.dnum * 2 * lo.rows,
                         np.random.default_rng(8000 + 100 * r + j).integers(0, lo.mods_all[j], lo.dnum * 2 * n, dtype=np.uint64))
        hs.load_ct(ct)
        hs.hoist_one()
        hs.rotate_hoisted_many(steps)
        return np.stack([hs.read_out(b) for b in range(R)])

    exp = run(oracle.OracleVP(spm_rows=lo.spm_rows_needed, ksk_rows=R * lo.ksk_limbs * lo.rows))
    with avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=R * lo.ksk_limbs * lo.rows, max_log_n=16) as vp:
        got = run(vp)
    assert np.array_equal(got, exp)
    assert not np.array_equal(got[0], got[1])                    # the slots really are different rotations
