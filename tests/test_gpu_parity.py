"""GPU parity tests: the HIP engine, through the C ABI, against the CPU oracle and the
reference's golden vectors.  Bit-exact (integer path): every comparison is array equality.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import oracle
from aloha_amd import harness, isa, programs
from aloha_amd import vp as avp
from util import N, P, Q0, Q1, load, manifest, reference_images
from test_oracle_golden import _own_images, check_case0, check_case2, run_case0, run_case2

pytestmark = pytest.mark.gpu

SPM, KSK = 0, 1


@pytest.fixture(scope="module")
def engine():
    vp = avp.VP(spm_rows=1 << 16, ksk_rows=1 << 16, max_log_n=16)
    yield vp
    vp.close()


@pytest.fixture(scope="module")
def engine_radix2():
    vp = avp.VP(spm_rows=1 << 14, ksk_rows=64, max_log_n=16, flags=avp.FLAG_RADIX2_ONLY)
    yield vp
    vp.close()


def residues(rng, shape, q, edge=True):
    a = rng.integers(0, q, size=shape, dtype=np.uint64)
    if edge:                                      # force the extremes into every limb
        flat = a.reshape(-1)
        flat[:4] = [0, 1, q - 1, q - 2]
        flat[-2:] = [q - 1, 0]
    return a


# ---- golden chains through the C ABI -------------------------------------------------------------
@pytest.mark.parametrize("which", ["reference_images", "own_programs"])
def test_case0_chain_on_gpu(engine, which):
    tb = harness.TestBench(engine)
    tb.load_programs(reference_images() if which == "reference_images" else _own_images())
    check_case0(run_case0(engine, tb))


def test_case2_chain_29_ops_on_gpu(engine):
    tb = harness.TestBench(engine)
    tb.load_programs(reference_images())
    check_case2(*run_case2(engine, tb))


def test_model_kernel_rotate_step4_on_gpu(engine):
    m = manifest()["case1"]
    tb = harness.TestBench(engine)
    tb.load_programs(reference_images())
    tb.load_ksk([np.zeros(12 * N, np.uint64), load(m["ksk_step4"])])
    engine.write(SPM, 0, load(m["kernel12_ct_before_rotate"]))
    tb.run_rotate(0x100, 0x000, 4)
    assert np.array_equal(tb.copy_spm(0x100), load(m["kernel12_ct_after_rotate"]))


def test_isram_hex_text_is_accepted_unchanged(engine):
    img = reference_images()["mul_plain"]
    engine.load_isram_hex(64, "// comment line\n" + isa.to_mem_text(img))
    m = manifest()["case0"]
    ct, pt = load(m["inst_0_out"]), load(m["inst_2_out"])[:2 * N]
    engine.write(SPM, 0, ct); engine.write(SPM, 0x200, pt)
    assert engine.run(64, src0=0, src1=0x200, rslt=0x300) == 0
    got = engine.read(SPM, 0x300, 4 * N).reshape(2, 2, N)
    ctr, ptr = ct.reshape(2, 2, N), pt.reshape(2, N)
    for c in range(2):
        for j, q in enumerate((Q0, Q1)):
            exp = (ctr[c, j].astype(object) * ptr[j].astype(object)) % q
            assert np.array_equal(got[c, j], exp.astype(np.uint64))


# ---- transforms vs the oracle ----------------------------------------------------------------------
def _ntt_case(vp, logn, moduli, polys_per_prime, inverse, rng, inplace=True):
    n = 1 << logn
    mods = [q for q in moduli for _ in range(polys_per_prime)]
    data = np.stack([residues(rng, n, q) for q in mods])
    rows = n // 128
    vp.write(SPM, 0, data.reshape(-1))
    dst = 0 if inplace else len(mods) * rows
    vp.ntt_rows(SPM, dst, 0, logn, mods, inverse=inverse)
    got = vp.read(SPM, dst, len(mods) * n).reshape(len(mods), n)
    for i, q in enumerate(mods):
        exp = oracle.ntt(data[i], q, inverse=inverse)
        assert np.array_equal(got[i], exp), f"logn={logn} limb {i} inverse={inverse}"
    if not inplace:                               # source must be untouched
        assert np.array_equal(vp.read(SPM, 0, len(mods) * n).reshape(len(mods), n), data)


@pytest.mark.parametrize("logn", [10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_single_limb_all_sizes(engine, logn, inverse):
    rng = np.random.default_rng(100 + logn)
    q = oracle.nth_prime(logn, 0) if logn != 13 else Q0
    _ntt_case(engine, logn, [q], 1, inverse, rng)


@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_reference_primes_n8192(engine, inverse):
    rng = np.random.default_rng(7)
    _ntt_case(engine, 13, [Q0, Q1, P], 2, inverse, rng)


@pytest.mark.parametrize("ppp", [1, 2, 3, 4, 5, 8, 16, 19])
def test_ntt_n65536_bundles(engine, ppp):
    """Every bundle shape of the row pass: 16/8/4 limbs per prime (LDS twiddles), 1-2 (global)."""
    rng = np.random.default_rng(200 + ppp)
    mods = [oracle.nth_prime(16, i) for i in range(2)]
    _ntt_case(engine, 16, mods, ppp, False, rng, inplace=(ppp % 2 == 0))
    _ntt_case(engine, 16, mods[:1], ppp, True, rng, inplace=(ppp % 2 == 1))


def test_ntt_worst_case_ranges(engine):
    """All-(q-1) and alternating inputs drive the lazy ranges to their bounds."""
    logn, n = 16, 1 << 16
    for idx in (0, 7):
        q = oracle.nth_prime(logn, idx)
        pats = np.stack([np.full(n, q - 1, np.uint64), np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64),
                         np.zeros(n, np.uint64), np.full(n, (1 << 60) - 1, np.uint64)])   # last: >= q, < 2q
        engine.write(SPM, 0, pats.reshape(-1))
        engine.ntt_rows(SPM, 0, 0, logn, [q] * 4)
        got = engine.read(SPM, 0, 4 * n).reshape(4, n)
        for i in range(4):
            src = pats[i] % np.uint64(q)
            assert np.array_equal(got[i], oracle.ntt(src, q)), f"pattern {i}"
        engine.ntt_rows(SPM, 0, 0, logn, [q] * 4, inverse=True)
        back = engine.read(SPM, 0, 4 * n).reshape(4, n)
        assert np.array_equal(back, pats % np.uint64(q))


def _is_prime(p):
    if p % 2 == 0:
        return p == 2
    d, r = p - 1, 0
    while d % 2 == 0:
        d //= 2; r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):       # deterministic below 3.3e24
        x = pow(a, d, p)
        if x in (1, p - 1):
            continue
        for _ in range(r - 1):
            x = x * x % p
            if x == p - 1:
                break
        else:
            return False
    return True


@pytest.mark.parametrize("logn", [12, 13, 16])
def test_ntt_primes_across_the_60_bit_range(engine, logn):
    """The lazy ranges are stated in units of q, so the margins differ between q just above 2^59 (memory words up
    to 2^60 - 1 are almost 2q) and q just below 2^60 (16q almost 2^64): primes drawn from the bottom, the middle and
    the top of (2^59, 2^60), adversarial and random words, forward, inverse and round trip against the oracle."""
    n = 1 << logn
    rng = np.random.default_rng(900 + logn)
    two_n = 2 * n
    primes = []
    for base in ((1 << 59) + 1, 3 << 58, (1 << 60) - (1 << 41)):   # the search below moves up by < 2^30 from base + 2^40
        r = (base + int(rng.integers(0, 1 << 40))) // two_n * two_n + 1
        for _ in range(100000):
            if _is_prime(r):
                break
            r += two_n
        assert _is_prime(r) and (1 << 59) < r < (1 << 60)
        primes.append(r)
    primes.append(oracle.nth_prime(logn, 0))                        # and the largest one there is
    for q in primes:
        pats = np.stack([np.full(n, q - 1, np.uint64), np.full(n, (1 << 60) - 1, np.uint64),
                         np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64),
                         np.where(np.arange(n) < n // 2, q - 1, 1).astype(np.uint64),
                         rng.integers(0, 1 << 60, n, dtype=np.uint64), rng.integers(0, q, n, dtype=np.uint64)])
        k = len(pats)
        canon = pats % np.uint64(q)
        for inverse, ref in ((False, oracle.ntt), (True, oracle.intt)):
            engine.write(SPM, 0, pats.reshape(-1))
            engine.ntt_rows(SPM, 0, 0, logn, [q] * k, inverse=inverse)
            got = engine.read(SPM, 0, k * n).reshape(k, n)
            for i in range(k):
                assert np.array_equal(got[i], ref(canon[i], q)), (hex(q), inverse, i)
            engine.ntt_rows(SPM, 0, 0, logn, [q] * k, inverse=not inverse)
            assert np.array_equal(engine.read(SPM, 0, k * n).reshape(k, n), canon), (hex(q), inverse, "round trip")


@pytest.mark.parametrize("logn", [12, 13, 14])
def test_single_pass_and_two_pass_forms_agree(logn):
    """N = 2^12..2^14 exist in two kernel forms (one pass through LDS, two passes of tiles): both against
    the oracle, both directions, several bundle shapes, in and out of place, worst-case ranges."""
    n, rows = 1 << logn, (1 << logn) // 128
    rng = np.random.default_rng(900 + logn)
    mods = [oracle.nth_prime(logn, i) for i in (0, 0, 0, 0, 0, 1, 1, 2)] + [Q0 if logn == 13 else oracle.nth_prime(logn, 3)]
    data = np.stack([residues(rng, n, q) for q in mods])
    data[0, :] = np.uint64((1 << 60) - 1)                          # non-canonical maximum (< 2q): widest lazy ranges
    data[1, ::2] = 0
    exp = np.stack([oracle.ntt(d % np.uint64(q), q) for d, q in zip(data, mods)])   # memory words < 2q mean d mod q
    for flags in (avp.FLAG_NTT_TWO_PASS, avp.FLAG_NTT_SINGLE_14):
        with avp.VP(spm_rows=4 * len(mods) * rows, ksk_rows=1, max_log_n=14, flags=flags) as vp:
            vp.write(SPM, 0, data.reshape(-1))
            vp.ntt_rows(SPM, len(mods) * rows, 0, logn, mods)                      # out of place
            f = vp.read(SPM, len(mods) * rows, len(mods) * n).reshape(len(mods), n)
            assert np.array_equal(f, exp), flags
            vp.ntt_rows(SPM, len(mods) * rows, len(mods) * rows, logn, mods, inverse=True)   # in place
            back = vp.read(SPM, len(mods) * rows, len(mods) * n).reshape(len(mods), n)
            assert np.array_equal(back, np.stack([d % np.uint64(q) for d, q in zip(data, mods)])), flags


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_tile_kernels_agree_with_radix2_fallback(engine, engine_radix2, logn):
    rng = np.random.default_rng(300 + logn)
    n = 1 << logn
    q = oracle.nth_prime(logn, 1)
    data = residues(rng, (5, n), q)
    for vp in (engine, engine_radix2):
        vp.write(SPM, 0, data.reshape(-1))
    out = []
    for vp in (engine, engine_radix2):
        vp.ntt_rows(SPM, 0, 0, logn, [q] * 5)
        f = vp.read(SPM, 0, 5 * n)
        vp.ntt_rows(SPM, 0, 0, logn, [q] * 5, inverse=True)
        out.append((f, vp.read(SPM, 0, 5 * n)))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], data.reshape(-1)) and np.array_equal(out[1][1], data.reshape(-1))


def test_ntt_full_batch_properties(engine):
    """BASELINE-size batch (256 limbs, N = 2^16): spot-check limbs against the oracle and
    check the size-independent properties on all of them: round trip and linearity."""
    logn, n, L, PPP = 16, 1 << 16, 16, 16
    rng = np.random.default_rng(11)
    mods = [oracle.nth_prime(logn, i) for i in range(L) for _ in range(PPP)]
    a = np.stack([residues(rng, n, q, edge=False) for q in mods])
    b = np.stack([residues(rng, n, q, edge=False) for q in mods])
    s = np.stack([(a[i] + b[i]) % np.uint64(q) for i, q in enumerate(mods)])        # a + b < 2^61: no wrap
    rows = len(mods) * n // 128
    vp = avp.VP(spm_rows=3 * rows, ksk_rows=1, max_log_n=16)
    try:
        for k, arr in enumerate((a, b, s)):
            vp.write(SPM, k * rows, arr.reshape(-1))
            vp.ntt_rows(SPM, k * rows, k * rows, logn, mods)
        fa, fb, fs = (vp.read(SPM, k * rows, len(mods) * n).reshape(len(mods), n) for k in range(3))
        for i in (0, 15, 16, 100, 255):
            assert np.array_equal(fa[i], oracle.ntt(a[i], mods[i]))
        for i, q in enumerate(mods):               # linearity: NTT(a) + NTT(b) == NTT(a + b)
            lhs = fa[i] + fb[i]
            lhs = np.where(lhs >= np.uint64(q), lhs - np.uint64(q), lhs)
            assert np.array_equal(lhs, fs[i]), f"linearity limb {i}"
            assert int(fa[i].max()) < q
        vp.ntt_rows(SPM, 0, 0, logn, mods, inverse=True)
        assert np.array_equal(vp.read(SPM, 0, len(mods) * n).reshape(len(mods), n), a)
    finally:
        vp.close()


# ---- element-wise ops, permutations: ISA level -----------------------------------------------------
def _run_asm(vp, text, **csr):
    vp.load_isram(1024, isa.assemble(text))
    return vp.run(1024, **csr)


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_elementwise_isa_ops_match_oracle(engine, logn):
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(400 + logn)
    q = Q1 if logn == 13 else oracle.nth_prime(logn, 3)
    a = residues(rng, n, q); b = residues(rng, n, q)
    a[5], b[5] = (1 << 60) - 1, (1 << 60) - 2          # non-canonical inputs (< 2q) get one conditional subtract
    s = int(rng.integers(0, q))
    prog = f"""
        vsetvl {64 * n:#x}
        vsetq {q:#x}
        vle v0, src0+0x0
        vle v1, src1+0x0
        vfqmul.vv v2, v0, v1
        vse v2, rslt+{0 * 8 * n:#x}
        vfqadd.vv v2, v0, v1
        vse v2, rslt+{1 * 8 * n:#x}
        vfqsub.vv v2, v0, v1
        vse v2, rslt+{2 * 8 * n:#x}
        vfqsub.vv v3, v1, v0
        vse v3, rslt+{3 * 8 * n:#x}
        vfqmul.vs v2, v0, {s:#x}
        vse v2, rslt+{4 * 8 * n:#x}
        vfqadd.vs v2, v1, {s:#x}
        vse v2, rslt+{5 * 8 * n:#x}
        vfqsub.vs v2, v0, {s:#x}
        vse v2, rslt+{6 * 8 * n:#x}
        vfqsub.sv v2, {s:#x}, v1
        vse v2, rslt+{7 * 8 * n:#x}
        vfqmod v2, v0
        vse v2, rslt+{8 * 8 * n:#x}
        vcp v4, v1
        vse v4, rslt+{9 * 8 * n:#x}
        break
    """
    o = oracle.OracleVP(spm_rows=16 * rows, ksk_rows=1)
    results = []
    for vp in (engine, o):
        vp.write(SPM, 0, a); vp.write(SPM, rows, b)
        vp.load_isram(1024, isa.assemble(prog))
        assert vp.run(1024, src0=0, src1=rows, rslt=2 * rows) == 0
        vp.sync()
        results.append(vp.read(SPM, 2 * rows, 10 * n))
    assert np.array_equal(results[0], results[1])
    assert int(results[0].max()) < q


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_vaut_vroli_match_oracle(engine, logn):
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(500 + logn)
    q = Q0 if logn == 13 else oracle.nth_prime(logn, 0)
    a = residues(rng, n, q)
    for k in (1, 3, pow(3, 2, 2 * n), pow(3, 8, 2 * n) % n | 1, n - 1, n + 5):
        prog = f"vsetvl {64 * n:#x}\nvsetq {q:#x}\nvle v0, src0+0x0\nvaut v2, v0, 0x0\nvse v2, rslt+0x0\n" \
               f"vaut v0, v0, 0x0\nvse v0, rslt+{8 * n:#x}\nvroli v3, v2, {k % 97:#x}\nvse v3, rslt+{16 * n:#x}\nbreak"
        engine.write(SPM, 0, a)
        assert _run_asm(engine, prog, src0=0, rslt=rows, step=k) == 0
        got = engine.read(SPM, rows, 3 * n).reshape(3, n)
        exp = oracle.vaut(a, q, k)
        assert np.array_equal(got[0], exp), f"vaut k={k}"
        assert np.array_equal(got[1], exp), f"in-place vaut k={k}"
        # scalar_iconn = csr_step + imm for vroli as well (expander.v:554-558)
        assert np.array_equal(got[2], oracle.vroli(exp, k + k % 97))


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_vautz_extension_matches_oracle(engine, logn):
    """vautz (a zero at a sign-flipped target of the automorphism becomes q): as a batched pass (vle; vautz; vse), in
    register form, folded into the scaled inverse transform (vintt; vfqmul.vs; [vfqadd.vs;] vautz; vse -- at N = 2^16 inside
    the last pass's kernel), and the identity it exists for, vautz(vintt(vautn(X))) == vaut(vintt(X)), on the engine."""
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(650 + logn)
    q = Q0 if logn == 13 else oracle.nth_prime(logn, 0)
    x = residues(rng, n, q)
    x[rng.random(n) < 0.6] = 0                                    # many zero coefficients
    X = oracle.ntt(x, q)
    c, add = 0x1234567, 0                                         # (a constant added on top would make the zeros non-zero)
    for k in (3, pow(3, 5, 2 * n), n + 7):
        prog = f"vsetvl {64 * n:#x}\nvsetq {q:#x}\nvle v0, src0+0x0\nvautz v2, v0, 0x0\nvse v2, rslt+0x0\n" \
               f"vle v1, src0+0x0\nvautz v3, v1, 0x0\nvse v3, rslt+{8 * n:#x}\nvse v3, rslt+{40 * n:#x}\n" \
               f"vle v4, src0+{8 * n:#x}\nvautn v6, v4, 0x0\nvintt v8, v6\nvfqmul.vs v10, v8, 0x1\nvautz v12, v10, 0x0\nvse v12, rslt+{16 * n:#x}\n" \
               f"vle v5, src0+{8 * n:#x}\nvintt v7, v5\nvaut v9, v7, 0x0\nvse v9, rslt+{24 * n:#x}\n" \
               f"vle v11, src0+{8 * n:#x}\nvintt v13, v11\nvfqmul.vs v15, v13, {c:#x}\nvautz v17, v15, 0x0\nvse v17, rslt+{32 * n:#x}\nbreak"
        engine.write(SPM, 0, np.concatenate([x, X]))
        assert _run_asm(engine, prog, src0=0, rslt=2 * rows, step=k) == 0
        got = engine.read(SPM, 2 * rows, 6 * n).reshape(6, n)
        assert np.array_equal(got[0], oracle.vautz(x, q, k)), f"batched vautz k={k}"
        assert np.array_equal(got[1], oracle.vautz(x, q, k)) and np.array_equal(got[5], got[1]), f"register vautz k={k}"   # (stored twice: not the batched form)
        assert np.array_equal(got[2], got[3]), f"vautz(vintt(vautn)) != vaut(vintt), k={k}"
        assert np.array_equal(got[3], oracle.vaut(x, q, k))
        scaled = np.array([(int(v) * c) % q for v in x], dtype=np.uint64)
        assert np.array_equal(got[4], oracle.vautz(scaled, q, k)), f"scaled inverse + vautz k={k}"


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_vautn_extension_matches_oracle(engine, logn):
    """vautn (NTT-form automorphism): fused batch (vle; vautn; vse), register form, and the identity
    vntt(vaut(x)) == vautn(vntt(x)) evaluated on the engine itself."""
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(600 + logn)
    q = Q1 if logn == 13 else oracle.nth_prime(logn, 1)
    a = residues(rng, n, q)
    for k in (3, pow(3, 5, 2 * n), n + 7, 2 * n - 1):
        prog = f"vsetvl {64 * n:#x}\nvsetq {q:#x}\nvle v0, src0+0x0\nvautn v2, v0, 0x0\nvse v2, rslt+0x0\n" \
               f"vle v1, src0+0x0\nvautn v1, v1, 0x0\nvcp v3, v1\nvse v3, rslt+{8 * n:#x}\n" \
               f"vle v4, src0+0x0\nvaut v6, v4, 0x0\nvntt v8, v6\nvse v8, rslt+{16 * n:#x}\n" \
               f"vle v5, src0+0x0\nvntt v7, v5\nvautn v9, v7, 0x0\nvse v9, rslt+{24 * n:#x}\nbreak"
        engine.write(SPM, 0, a)
        assert _run_asm(engine, prog, src0=0, rslt=rows, step=k) == 0
        got = engine.read(SPM, rows, 4 * n).reshape(4, n)
        assert np.array_equal(got[0], oracle.vautn(a, k)), f"fused vautn k={k}"
        assert np.array_equal(got[1], oracle.vautn(a, k)), f"register vautn k={k}"
        assert np.array_equal(got[2], got[3]), f"vntt(vaut) != vautn(vntt), k={k}"
        assert np.array_equal(got[3], oracle.vautn(oracle.ntt(a, q), k))


def test_sequencer_fusion_is_exact_and_batches(engine):
    """the fusing sequencer (zero-copy vle, direct stores, batched vle->vntt->vse
    runs) must leave every byte a vse writes identical to plain sequential execution."""
    logn, n, L = 16, 1 << 16, 8
    rows = n // 128
    mods = [oracle.nth_prime(logn, i // 4) for i in range(L)]          # 2 primes x 4 limbs
    rng = np.random.default_rng(77)
    data = np.stack([residues(rng, n, q) for q in mods])
    plain = avp.VP(spm_rows=4 * L * rows, ksk_rows=1, max_log_n=16, flags=avp.FLAG_NO_FUSION)
    try:
        # (1) encode_post for L limbs: one batched transform instead of L
        words = programs.assemble(programs.encode_post(n, mods))
        outs, launches = [], []
        for vp in (engine, plain):
            vp.write(SPM, 0, data.reshape(-1))
            vp.load_isram(0, words)
            k0 = vp.stats()["kernel_launches"]
            assert vp.run(0, src0=0, rslt=L * rows) == 0
            vp.sync()
            launches.append(vp.stats()["kernel_launches"] - k0)
            outs.append(vp.read(SPM, L * rows, L * n))
        assert np.array_equal(outs[0], outs[1])
        for i, q in enumerate(mods):
            assert np.array_equal(outs[0].reshape(L, n)[i], oracle.ntt(data[i], q))
        assert launches[0] == 2 and launches[1] == 2 * L, launches
        # (2) a program that mixes aliases, in-place stores, overlapping loads and live registers
        q = mods[0]
        prog = f"""
            vsetvl {64 * n:#x}
            vsetq {q:#x}
            vle v0, src0+0x0
            vle v1, src0+{8 * n:#x}
            vfqadd.vv v2, v0, v1
            vse v2, src0+0x0            # overwrites what v0 aliases while v0 is still live
            vfqmul.vv v4, v0, v1        # must see the OLD v0
            vse v4, rslt+0x0
            vle v3, src0+0x0
            vntt v5, v3
            vse v5, src0+0x0            # in-place fused transform
            vle v6, src0+0x0            # reload after the transform
            vfqsub.vv v7, v6, v1
            vse v7, rslt+{8 * n:#x}
            vaut v8, v6, 0x0
            vse v8, rslt+{16 * n:#x}
            vintt v10, v6
            vse v10, rslt+{24 * n:#x}
            vse v6, rslt+{32 * n:#x}     # v6 still readable after being a transform source? (RTL: undefined; skipped)
            break
        """
        prog = "\n".join(l for l in prog.splitlines() if "undefined; skipped" not in l)
        w2 = isa.assemble(prog)
        res = []
        o = oracle.OracleVP(spm_rows=8 * rows, ksk_rows=1)
        for vp in (engine, plain, o):
            vp.write(SPM, 0, data[:2].reshape(-1))
            vp.load_isram(0, np.zeros((64, 3), np.uint32)); vp.load_isram(0, w2)
            assert vp.run(0, src0=0, rslt=2 * rows, step=3) == 0
            vp.sync()
            res.append(vp.read(SPM, 0, 6 * n))
        assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    finally:
        plain.close()


def _random_program(rng, n, primes, nslots, length):
    """Random straight-line program over v0..v9 and `nslots` limb slots of src0.  Registers that
    a transform left undefined (its vs1) or that were never written are not read."""
    lines = [f"vsetvl {64 * n:#x}", f"vsetq {primes[0]:#x}"]
    defined = set()
    regs = list(range(10))

    def pick_def(parity=None):
        c = [r for r in sorted(defined) if parity is None or r % 2 == parity]
        return int(rng.choice(c)) if c else None

    def slot():
        return f"src0+{int(rng.integers(nslots)) * 8 * n:#x}"

    for _ in range(length):
        op = rng.choice(["vle", "vle", "vse", "vv", "vs", "un", "perm", "ntt", "setq", "dot", "sdot", "stores", "iscale"])
        vd = int(rng.choice(regs))
        if op == "vle":
            lines.append(f"vle v{vd}, {slot()}"); defined.add(vd)
        elif op == "vse":
            r = pick_def()
            if r is not None:
                lines.append(f"vse v{r}, {slot()}")
        elif op == "vv":
            a = pick_def()
            b = pick_def(1 - a % 2) if a is not None else None
            if b is not None:
                lines.append(f"{rng.choice(['vfqmul', 'vfqadd', 'vfqsub'])}.vv v{vd}, v{a}, v{b}"); defined.add(vd)
        elif op == "vs":
            a = pick_def()
            if a is not None:
                s = int(rng.integers(1, 1 << 59))
                m = rng.choice(["vfqmul.vs", "vfqadd.vs", "vfqsub.vs", "vfqsub.sv"])
                lines.append(f"{m} v{vd}, {s:#x}, v{a}" if m.endswith(".sv") else f"{m} v{vd}, v{a}, {s:#x}")
                defined.add(vd)
        elif op == "un":
            a = pick_def()
            if a is not None:
                lines.append(f"{rng.choice(['vfqmod', 'vcp'])} v{vd}, v{a}"); defined.add(vd)
        elif op == "perm":
            a = pick_def()
            if a is not None:
                # (the raw `q - x` of vaut needs a canonical x: real programs permute iNTT outputs)
                t2 = int(rng.choice(regs))
                lines.append(f"vcp v{t2}, v{a}")
                lines.append(f"{rng.choice(['vaut', 'vroli', 'vautn'])} v{vd}, v{t2}, {2 * int(rng.integers(8)):#x}")
                defined.update({t2, vd})
        elif op == "ntt":
            a = pick_def()
            if a is not None and a != vd:
                lines.append(f"{rng.choice(['vntt', 'vintt'])} v{vd}, v{a}")
                defined.discard(a); defined.add(vd)
        elif op == "setq":
            lines.append(f"vsetq {int(rng.choice(primes)):#x}")
        elif op == "dot":                                    # the accumulate idiom of the key-switch inner product
            acc, t = 6, 3
            for i in range(int(rng.integers(1, 4))):
                lines += [f"vle v0, {slot()}", f"vle v1, {slot()}"]
                lines += [f"vfqmul.vv v{acc}, v0, v1"] if i == 0 else [f"vfqmul.vv v{t}, v0, v1", f"vfqadd.vv v{acc}, v{acc}, v{t}"]
            defined.update({0, 1, acc}); defined.add(t) if i else None
            if rng.integers(2):
                lines.append(f"vse v{acc}, {slot()}")
        elif op == "sdot":                                   # the basis-conversion idiom: memory operands times scalars, summed
            acc, t = 6, 3
            mixed = bool(rng.integers(2))
            for i in range(int(rng.integers(1, 5))):
                sc = int(rng.integers(1, 1 << 60))
                if mixed and i % 2:
                    lines += [f"vle v0, {slot()}", f"vle v1, {slot()}"]
                    lines += [f"vfqmul.vv v{acc}, v0, v1"] if i == 0 else [f"vfqmul.vv v{t}, v0, v1", f"vfqadd.vv v{acc}, v{acc}, v{t}"]
                    defined.add(1)
                else:
                    lines += [f"vle v0, {slot()}"]
                    lines += [f"vfqmul.vs v{acc}, v0, {sc:#x}"] if i == 0 else [f"vfqmul.vs v{t}, v0, {sc:#x}", f"vfqadd.vv v{acc}, v{acc}, v{t}"]
            defined.update({0, acc}); defined.add(t) if i else None
            choice = int(rng.integers(4))
            if choice == 3:                                  # constant terms folded into the sum, then stored
                for _ in range(int(rng.integers(1, 3))):
                    lines.append(f"{rng.choice(['vfqadd.vs', 'vfqsub.vs'])} v{acc}, v{acc}, {int(rng.integers(0, 1 << 60)):#x}")
                lines.append(f"vse v{acc}, {slot()}")
            elif choice == 0:
                lines.append(f"vse v{acc}, {slot()}")
            elif choice == 1:                                # a lazy value read by another op: must be evaluated first
                lines.append(f"vfqsub.vs v2, v{acc}, {int(rng.integers(1, 1 << 59)):#x}"); defined.add(2)
        elif op == "iscale":                                 # inverse transform, scalar, [constant], store: one scaled transform job
            for _ in range(int(rng.integers(1, 4))):
                e = int(rng.choice([2, 4]))
                lines += [f"vle v0, {slot()}", "vintt v2, v0", f"vfqmul.vs v{e}, v2, {int(rng.integers(1, 1 << 60)):#x}"]
                if rng.integers(2):
                    lines.append(f"vfqadd.vs v{e}, v{e}, {int(rng.integers(0, 1 << 60)):#x}")
                lines.append(f"vse v{e}, {slot()}")
            defined.difference_update({0, 2, 4})
            if rng.integers(3) == 0:                         # ... or the product is read again afterwards: no fusion
                defined.add(e)
        elif op == "stores":                                 # several sums stored back to back (one deferred launch), with hazards
            for _ in range(int(rng.integers(2, 4))):
                lines += [f"vle v0, {slot()}", f"vle v1, {slot()}", "vfqmul.vv v6, v0, v1", f"vle v0, {slot()}",
                          f"vfqmul.vs v3, v0, {int(rng.integers(1, 1 << 59)):#x}", "vfqadd.vv v6, v6, v3", f"vse v6, {slot()}"]
            defined.update({0, 1, 3, 6})
    for r in sorted(defined):                                # flush every live register to memory
        lines.append(f"vse v{r}, src0+{(nslots + r) * 8 * n:#x}")
    return "\n".join(lines + ["break"])


@pytest.mark.parametrize("logn", [11, 12, 13, 14, 15, 16])
def test_scaled_inverse_transform_idiom_matches_oracle(engine, logn):
    """`vle; vintt; vfqmul.vs c; [vfqadd.vs a;] vse` leaves the sequencer as ONE transform job whose last stage carries
    the scalar (and whose canonical result gets the constant): every kernel form of the inverse transform (single pass,
    small-N column kernels, the N = 2^16 column kernel; N = 2^11 stays unfused), several primes in one batch, scalars
    and constants at the edges, and variants the fusion must refuse (the transform's register read again later)."""
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(300 + logn)
    primes = [oracle.nth_prime(max(logn, 13), i) for i in range(3)]
    nin = 6
    data = np.stack([rng.integers(0, 1 << 60, n, dtype=np.uint64) for _ in range(nin)])
    lines = [f"vsetvl {64 * n:#x}"]
    off = lambda slot: slot * 8 * n
    out = nin
    for i in range(nin):
        q = primes[i % 3]
        c = [1, q - 1, int(rng.integers(1, q)), (1 << 60) - 1, int(rng.integers(1, q)), 2][i]
        e = 2 if i % 2 else 4
        lines += [f"vsetq {q:#x}", f"vle v0, src0+{off(i):#x}", "vintt v2, v0", f"vfqmul.vs v{e}, v2, {c:#x}"]
        if i % 3:
            lines.append(f"vfqadd.vs v{e}, v{e}, {[0, q - 1, int(rng.integers(0, q))][i % 3]:#x}")
        lines.append(f"vse v{e}, src0+{off(out):#x}")
        out += 1
    # refused: the untransformed... the transform result is read again (v2 stored as well), and a vfqsub.vs in the chain
    q = primes[0]
    lines += [f"vsetq {q:#x}", f"vle v0, src0+{off(0):#x}", "vintt v2, v0", "vfqmul.vs v4, v2, 0x5", f"vse v4, src0+{off(out):#x}",
              f"vse v2, src0+{off(out + 1):#x}"]
    out += 2
    lines += [f"vle v0, src0+{off(1):#x}", "vintt v2, v0", "vfqmul.vs v4, v2, 0x7", "vfqsub.vs v4, v4, 0x3", f"vse v4, src0+{off(out):#x}"]
    out += 1
    words = isa.assemble("\n".join(lines + ["break"]))
    o = oracle.OracleVP(spm_rows=out * rows, ksk_rows=1)
    res = []
    for vp in (engine, o):
        vp.write(SPM, 0, np.zeros(out * n, np.uint64)); vp.write(SPM, 0, data.reshape(-1))
        vp.load_isram(0, np.zeros((isa.IRAM_DEPTH, 3), np.uint32)); vp.load_isram(0, words)
        assert vp.run(0, src0=0) == 0
        vp.sync()
        res.append(vp.read(SPM, 0, out * n))
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("logn", [10, 13, 16])
def test_grouped_basis_conversion_sums_match_oracle(engine, logn):
    """Sums of scaled operands that SHARE their operands (the shape of a ModUp / mod-down basis conversion) leave the
    sequencer as groups: T = 1..8 shared operands (both product forms of the kernel), 5-9 outputs per operand set under
    different moduli, with and without constant terms, operands that are NOT canonical for the target modulus, and
    jobs that do not qualify (a vector x vector term, a lone operand set) in the same deferred batch."""
    n = 1 << logn
    rows = n // 128
    rng = np.random.default_rng(77 + logn)
    primes = [oracle.nth_prime(max(logn, 13), i) for i in range(9)]
    nops = 8
    ops = np.stack([rng.integers(0, 1 << 60, n, dtype=np.uint64) for _ in range(nops)])      # any 60-bit words
    ops[0, :4] = [0, (1 << 60) - 1, primes[0], primes[0] - 1]
    lines = [f"vsetvl {64 * n:#x}"]
    off = lambda slot: slot * 8 * n
    out = nops
    for T in range(1, 9):
        for j in range(5 + T % 5):
            q = primes[(j + T) % len(primes)]
            lines.append(f"vsetq {q:#x}")
            for t in range(T):
                sc = int(rng.integers(0, q)) if (t + j) % 4 else (1 if t % 2 else q - 1)
                lines += [f"vle v0, src0+{off(t):#x}", f"vfqmul.vs v{6 if t == 0 else 3}, v0, {sc:#x}"]
                if t:
                    lines.append("vfqadd.vv v6, v6, v3")
            if j % 3 == 1:
                lines.append(f"vfqsub.vs v6, v6, {int(rng.integers(0, q)):#x}")
            if j % 3 == 2:
                lines.append(f"vfqadd.vs v6, v6, {int(rng.integers(0, q)):#x}")
            lines.append(f"vse v6, src0+{off(out):#x}")
            out += 1
    # the same batch also carries jobs that must stay ordinary dot products
    lines += [f"vsetq {primes[0]:#x}", f"vle v0, src0+{off(0):#x}", f"vle v1, src0+{off(1):#x}", "vfqmul.vv v6, v0, v1",
              f"vle v0, src0+{off(2):#x}", f"vfqmul.vs v3, v0, 0x5", "vfqadd.vv v6, v6, v3", f"vse v6, src0+{off(out):#x}"]
    out += 1
    lines += [f"vle v0, src0+{off(7):#x}", f"vfqmul.vs v6, v0, 0x7", f"vle v0, src0+{off(3):#x}", "vfqmul.vs v3, v0, 0x9",
              "vfqadd.vv v6, v6, v3", f"vse v6, src0+{off(out):#x}"]
    out += 1
    words = isa.assemble("\n".join(lines + ["break"]))
    assert len(words) <= isa.IRAM_DEPTH
    o = oracle.OracleVP(spm_rows=out * rows, ksk_rows=1)
    res = []
    for vp in (engine, o):
        vp.write(SPM, 0, np.zeros(out * n, np.uint64)); vp.write(SPM, 0, ops.reshape(-1))
        vp.load_isram(0, np.zeros((isa.IRAM_DEPTH, 3), np.uint32)); vp.load_isram(0, words)
        assert vp.run(0, src0=0) == 0
        vp.sync()
        res.append(vp.read(SPM, 0, out * n))
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("seed", range(int(os.environ.get("ALOHA_FUZZ_SEEDS", "32"))))   # (a longer soak: ALOHA_FUZZ_SEEDS=400)
def test_random_programs_fused_unfused_oracle(engine, seed):
    """Differential fuzz of the sequencer: fused engine == unfused engine == oracle on all memory."""
    rng = np.random.default_rng(1000 + seed)
    logn = int(rng.choice([10, 12, 13]))
    n, rows, nslots = 1 << logn, (1 << logn) // 128, 6
    primes = [oracle.nth_prime(13, i) for i in range(2)]          # q = 1 mod 2^14 serves every N <= 8192
    mem = np.stack([residues(rng, n, primes[0]) for _ in range(nslots)])
    mem[1, :8] = 0                                                # zeros exercise the flipped-zero corner of vaut
    words = isa.assemble(_random_program(rng, n, primes, nslots, 40 if seed < 12 else 70))
    total = (nslots + 10) * n
    plain = avp.VP(spm_rows=(nslots + 10) * rows, ksk_rows=1, max_log_n=13, flags=avp.FLAG_NO_FUSION)
    o = oracle.OracleVP(spm_rows=(nslots + 10) * rows, ksk_rows=1)
    try:
        res = []
        for vp in (engine, plain, o):
            vp.write(SPM, 0, np.zeros(total, np.uint64))
            vp.write(SPM, 0, mem.reshape(-1))
            vp.load_isram(0, np.zeros((1024, 3), np.uint32)); vp.load_isram(0, words)
            rc = vp.run(0, src0=0, step=2 * seed + 1)
            vp.sync()
            res.append((rc, vp.read(SPM, 0, total)))
        assert res[0][0] == res[1][0] == res[2][0] == 0
        assert np.array_equal(res[0][1], res[2][1]), "fused engine vs oracle"
        assert np.array_equal(res[1][1], res[2][1]), "unfused engine vs oracle"
        # the same program again on OTHER data: the fused engine now replays its recorded launch list
        mem2 = np.stack([residues(rng, n, primes[0]) for _ in range(nslots)])
        again = []
        for vp in (engine, o):
            vp.write(SPM, 0, np.zeros(total, np.uint64))
            vp.write(SPM, 0, mem2.reshape(-1))
            assert vp.run(0, src0=0, step=2 * seed + 1) == 0
            vp.sync()
            again.append(vp.read(SPM, 0, total))
        assert np.array_equal(again[0], again[1]), "replayed run vs oracle"
        # ... and twice more (with ALOHA_RUN_GRAPH=1 the list is captured into a hipGraph on its second replay)
        for rep in range(2):
            mem3 = np.stack([residues(rng, n, primes[0]) for _ in range(nslots)])
            third = []
            for vp in (engine, o):
                vp.write(SPM, 0, np.zeros(total, np.uint64))
                vp.write(SPM, 0, mem3.reshape(-1))
                assert vp.run(0, src0=0, step=2 * seed + 1) == 0
                vp.sync()
                third.append(vp.read(SPM, 0, total))
            assert np.array_equal(third[0], third[1]), f"graph replay #{rep} vs oracle"
    finally:
        plain.close()


def test_case0_chain_fused_equals_unfused():
    plain = avp.VP(spm_rows=1 << 16, ksk_rows=1 << 16, max_log_n=13, flags=avp.FLAG_NO_FUSION)
    try:
        tb = harness.TestBench(plain)
        tb.load_programs(reference_images())
        check_case0(run_case0(plain, tb))
    finally:
        plain.close()


def test_error_codes_match_oracle(engine):
    progs = [
        (f"vsetvl 0x80000\nvsetq {Q0:#x}\nvfqadd.vv v2, v0, v4\nbreak", {}),
        ("vsetvl 0x80001\nbreak", {}),
        ("vsetvl 0x80000\nvsetq 0x123\nbreak", {}),
        ("vsetvl 0x80000\nvle v0, src0+0x0\nbreak", {"src0": (1 << 16) - 10}),
        (f"vsetvl 0x80000\nvsetq {Q0:#x}\nvaut v2, v0, 0x2\nbreak", {}),
        (f"vsetvl 0x100000\nvsetq {P:#x}\nvle v0, src0+0x0\nvntt v2, v0\nbreak", {}),
        (f"vsetvl 0x80000\nvsetq {Q0:#x}\nvfqadd.vs v2, v0, {2 * Q0:#x}\nbreak", {}),
        (f"vsetvl 0x80000\nvsetq {Q0:#x}\nvle v0, src0+0x0\nvcp v2, v0", {}),        # no break
    ]
    o = oracle.OracleVP()
    for text, csr in progs:
        words = isa.assemble(text)
        codes = []
        for vp in (engine, o):
            vp.load_isram(2048, np.zeros((2048, 3), np.uint32))
            vp.load_isram(2048, words)
            codes.append(vp.run(2048, **csr))
        assert codes[0] == codes[1] and codes[0] != 0, (text, codes)
    engine.load_isram(2048, np.array([[0xFC00000B, 0, 0]], dtype=np.uint32))
    assert engine.run(2048) == avp.ERR_DECODE


def test_boundary_and_edge_cases(engine):
    """Empty batches, exact-fit and one-past-the-end row ranges, ROM bounds, pointer-select
    corners of vle/vse (vp_top_full.sv:105-117) -- engine and oracle agree everywhere."""
    n, rows = 8192, 64
    q = Q0
    rng = np.random.default_rng(9)
    a = residues(rng, n, q)
    # empty batch is a no-op; a batch ending exactly at the last row is fine; one row further is not
    engine.ntt_rows(SPM, 0, 0, 13, [])
    last = engine.spm_rows - rows
    engine.write(SPM, last, a)
    engine.ntt_rows(SPM, last, last, 13, [q])
    assert np.array_equal(engine.read(SPM, last, n), oracle.ntt(a, q))
    with pytest.raises(avp.AlohaError) as e:
        engine.ntt_rows(SPM, last + 1, last + 1, 13, [q])
    assert e.value.status == avp.ERR_ADDR
    with pytest.raises(avp.AlohaError) as e:
        engine.write(SPM, engine.spm_rows - 1, np.zeros(129, np.uint64))
    assert e.value.status == avp.ERR_ADDR
    with pytest.raises(avp.AlohaError) as e:
        engine.read(KSK, engine.ksk_rows, 1)
    assert e.value.status == avp.ERR_ADDR
    with pytest.raises(avp.AlohaError) as e:
        engine.ntt_rows(SPM, 0, 0, 13, [q + 2])
    assert e.value.status == avp.ERR_MODULUS
    with pytest.raises(avp.AlohaError) as e:
        engine.ntt_rows(SPM, 0, 0, 9, [q])
    assert e.value.status == avp.ERR_VL
    # instruction ROM bounds
    with pytest.raises(avp.AlohaError) as e:
        engine.load_isram(4095, np.zeros((2, 3), np.uint32))
    assert e.value.status == avp.ERR_ADDR
    engine.load_isram(4095, isa.assemble("break"))
    assert engine.run(4095) == 0
    engine.load_isram(4095, isa.assemble("nop"))
    o = oracle.OracleVP()
    o.load_isram(4095, isa.assemble("nop"))
    assert engine.run(4095) == o.run(4095) == avp.ERR_NOBREAK
    # pointer selects: 1 = src1, 2 = rslt, 15 = KSK for loads but SPM base 0 for stores, others base 0
    prog = f"""
        vsetvl {64 * n:#x}
        vsetq {q:#x}
        vle v0, ksk+0x0
        vle v1, src1+0x0
        vfqadd.vv v2, v0, v1
        vse v2, ksk+{8 * n:#x}
        vle v3, abs+{8 * n:#x}
        vfqmul.vv v4, v0, v3
        vse v4, rslt+0x0
        break
    """
    kdat, sdat = residues(rng, n, q), residues(rng, n, q)
    outs = []
    for vp in (engine, o):
        vp.write(KSK, 5 * rows, kdat); vp.write(SPM, 7 * rows, sdat)
        vp.write(SPM, 0, np.zeros(4 * n, np.uint64))
        vp.load_isram(0, np.zeros((16, 3), np.uint32)); vp.load_isram(0, isa.assemble(prog))
        assert vp.run(0, src0=0, src1=7 * rows, rslt=2 * rows, ksk=5 * rows) == 0
        vp.sync()
        outs.append(vp.read(SPM, 0, 4 * n))
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0][n:2 * n], (kdat + sdat) % np.uint64(q))       # the `vse ..., ksk+` landed in SPM row `rows`


def test_poly_mul_program_config3(engine):
    """BASELINE config 3: c = iNTT(NTT(a) . NTT(b)), N = 2^14, 8 limbs, via the ISA."""
    logn, n, L = 14, 1 << 14, 8
    rows = n // 128
    mods = [oracle.nth_prime(logn, i) for i in range(L)]
    rng = np.random.default_rng(33)
    a = np.stack([residues(rng, n, q) for q in mods]); b = np.stack([residues(rng, n, q) for q in mods])
    words = programs.assemble(programs.poly_mul(n, mods))
    words_b = programs.assemble(programs.poly_mul_batched(n, mods, scratch_row=3 * L * rows))
    o = oracle.OracleVP(spm_rows=4 * L * rows, ksk_rows=1)
    res = []
    for vp, w in ((engine, words), (o, words), (engine, words_b), (o, words_b)):
        vp.write(SPM, 0, a.reshape(-1)); vp.write(SPM, L * rows, b.reshape(-1))
        vp.load_isram(512, np.zeros((512, 3), np.uint32)); vp.load_isram(512, w)
        k0 = vp.stats()["kernel_launches"] if hasattr(vp, "stats") else 0
        assert vp.run(512, src0=0, src1=L * rows, rslt=2 * L * rows) == 0
        vp.sync()
        res.append(vp.read(SPM, 2 * L * rows, L * n))
        if w is words_b and hasattr(vp, "stats"):
            assert vp.stats()["kernel_launches"] - k0 <= 8        # 3 transform batches (2 kernels each) + 1 product batch
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[2], res[3]) and np.array_equal(res[0], res[2])
    # and the answer is the negacyclic product: check one coefficient by direct convolution
    q = mods[0]
    j = 12345
    acc = 0
    for i in range(n):                             # X^n = -1: indices that wrap pick up a minus sign
        if i <= j:
            acc += int(a[0][i]) * int(b[0][j - i])
        else:
            acc -= int(a[0][i]) * int(b[0][j - i + n])
    assert int(res[0][j]) == acc % q


def test_recorded_runs_survive_cache_evictions():
    """Recorded launch lists point to cached device job arrays and transform plans; both caches evict (512 arrays,
    128 plans).  700 distinct programs push everything of the first ones out; running those again -- on new data --
    must notice (the epoch) and sequence afresh instead of replaying launches that point to freed memory."""
    n, rows = 1024, 8
    q = oracle.nth_prime(13, 0)
    rng = np.random.default_rng(5)

    def prog(i):
        return isa.assemble("\n".join([
            f"vsetvl {64 * n:#x}", f"vsetq {q:#x}",
            "vle v0, src0+0x0", f"vfqmul.vs v2, v0, {i + 2:#x}", f"vse v2, src0+{2 * 8 * n:#x}",
            f"vle v0, src0+{8 * n:#x}", f"vfqmul.vs v2, v0, {3 * i + 5:#x}", f"vse v2, src0+{3 * 8 * n:#x}",
            "vle v0, src0+0x0", "vntt v2, v0", f"vse v2, src0+{4 * 8 * n:#x}",
            f"vle v0, src0+{8 * n:#x}", "vntt v2, v0", f"vse v2, src0+{5 * 8 * n + (i % 200) * 1024:#x}", "break"]))   # 200 distinct plans

    S = 48                                                       # limb slots of memory
    with avp.VP(spm_rows=S * rows, ksk_rows=1, max_log_n=10) as vp:
        o = oracle.OracleVP(spm_rows=S * rows, ksk_rows=1)

        def check(i):
            data = rng.integers(0, q, 2 * n, dtype=np.uint64)
            out = []
            for e in (vp, o):
                e.write(SPM, 0, np.zeros(S * n, np.uint64)); e.write(SPM, 0, data)
                e.load_isram(0, prog(i))
                assert e.run(0, src0=0) == 0
                e.sync()
                out.append(e.read(SPM, 0, S * n))
            assert np.array_equal(out[0], out[1]), i

        for i in (0, 1, 2):
            check(i); check(i)                                   # recorded, then replayed
        for i in range(3, 700):                                  # evict
            vp.load_isram(0, prog(i))
            assert vp.run(0, src0=0) == 0
        vp.sync()
        for i in (0, 1, 2, 699, 0):
            check(i)


def test_more_than_65535_limbs_in_one_batch():
    """the stage kernels of N < 2^12 carry the limb index in gridDim.y (<= 65535); a larger batch used to pass
    validation and fail at launch.  70 000 limbs of N = 1024: spot checks against the oracle across the chunk boundary
    and a full round trip."""
    logn, n, count = 10, 1024, 70000
    q = oracle.nth_prime(10, 0)
    rng = np.random.default_rng(65535)
    data = rng.integers(0, q, (count, n), dtype=np.uint64)
    with avp.VP(spm_rows=count * n // 128, ksk_rows=1, max_log_n=10) as vp:
        vp.write(SPM, 0, data.reshape(-1))
        vp.ntt_rows(SPM, 0, 0, logn, [q] * count)
        got = vp.read(SPM, 0, count * n).reshape(count, n)
        for i in (0, 65534, 65535, 65536, 69999):
            assert np.array_equal(got[i], oracle.ntt(data[i], q)), i
        vp.ntt_rows(SPM, 0, 0, logn, [q] * count, inverse=True)
        assert np.array_equal(vp.read(SPM, 0, count * n).reshape(count, n), data)


def test_twiddle_cache_generations(monkeypatch):
    """The twiddle cache is bounded; a call that would overflow it empties it first (a new generation: plans, argument
    caches and recorded runs go with it).  With room for 3 sets: transforms under 7 primes in turn, batches that need 2
    and 3 new sets at once, a program run again after its sets were dropped -- every result equals the oracle's -- and a
    batch needing more sets than the cache holds is refused."""
    monkeypatch.setenv("ALOHA_MAX_TWIDDLE_SETS", "3")
    logn, n = 12, 1 << 12
    primes = [oracle.nth_prime(logn, i) for i in range(7)]
    rng = np.random.default_rng(12)
    with avp.VP(spm_rows=8 * n // 128, ksk_rows=1, max_log_n=logn) as vp:
        def check(mods):
            data = np.stack([residues(rng, n, q) for q in mods])
            vp.write(SPM, 0, data.reshape(-1))
            vp.ntt_rows(SPM, 0, 0, logn, mods)
            got = vp.read(SPM, 0, len(mods) * n).reshape(len(mods), n)
            for i, q in enumerate(mods):
                assert np.array_equal(got[i], oracle.ntt(data[i], q)), (mods, i)
        for q in primes:
            check([q])
        check([primes[0], primes[0], primes[5]])
        check([primes[1], primes[2], primes[3]])
        check([primes[1], primes[2], primes[3]])                # cached arguments of the current generation
        assert vp.stats()["twiddle_generations"] >= 3
        with pytest.raises(avp.AlohaError):
            check(primes[:4])                                    # four sets at once do not fit three slots
        # a program (two moduli) recorded, its sets dropped by other work, run again
        words = isa.assemble("\n".join([f"vsetvl {64 * n:#x}", f"vsetq {primes[0]:#x}", "vle v0, src0+0x0", "vntt v2, v0", "vse v2, rslt+0x0",
                                         f"vsetq {primes[1]:#x}", f"vle v0, src0+{8 * n:#x}", "vntt v2, v0", f"vse v2, rslt+{8 * n:#x}", "break"]))
        vp.load_isram(0, words)
        a = np.stack([residues(rng, n, primes[0]), residues(rng, n, primes[1])])
        for rep in range(3):
            vp.write(SPM, 0, a.reshape(-1))
            assert vp.run(0, src0=0, rslt=2 * n // 128) == 0
            got = vp.read(SPM, 2 * n // 128, 2 * n).reshape(2, n)
            assert np.array_equal(got[0], oracle.ntt(a[0], primes[0])) and np.array_equal(got[1], oracle.ntt(a[1], primes[1]))
            check([primes[4], primes[5], primes[6]])             # evicts the program's two sets
        # a program that transforms under the modulus it INHERITS (no vsetq of its own) on a full cache: the inherited
        # (q, N) pair has no set any more and must be counted when deciding whether to start a new generation
        head = isa.assemble("\n".join([f"vsetvl {64 * n:#x}", f"vsetq {primes[0]:#x}", "break"]))
        body = isa.assemble("\n".join(["vle v0, src0+0x0", "vntt v2, v0", "vse v2, rslt+0x0", "break"]))
        vp.load_isram(0, head); vp.load_isram(16, body)
        assert vp.run(0) == 0                                    # q = primes[0], VL = N; no transform yet
        check([primes[4], primes[5], primes[6]])                 # the cache is full of other sets
        vp.write(SPM, 0, a[0])
        assert vp.run(16, src0=0, rslt=2 * n // 128) == 0, vp.last_error
        assert np.array_equal(vp.read(SPM, 2 * n // 128, n), oracle.ntt(a[0], primes[0]))
