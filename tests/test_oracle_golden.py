"""Pins the CPU oracle against the reference's own golden vectors.

Everything here runs on the CPU.  The fixtures under tests/golden/ were converted from the
reference's ``tv/`` directory by tests/golden/make_fixtures.py; where /root/reference is
present the complete ``tv/`` tree is swept as well (``-m reference`` tests).
"""
import glob
import os
import random

import numpy as np
import pytest

import oracle
from aloha_amd import harness, isa, programs
from util import (HAVE_REFERENCE, N, P, PRIMES, Q0, Q1, REFERENCE, load, manifest, read_tv,
                  reference_images, sha)

# Appendix D of SURVEY.md; reference sim/vp/tf_rom_generator/tf_rom_generator.sv:75-77
PSI = {Q0: 3825716582911, Q1: 79932510954937, P: 101017252977188}
IPSI = {Q0: 264250557364078134, Q1: 101614808487310449, P: 106746493840490977}
# vsetiq operands of the shipped programs (keyswitch.mem:3,8,12)
MU = {Q0: 0x3fffff78000120f7, Q1: 0x3ffffec000063ff7, P: 0x3ffff00001fdfff9}


# ---- scalar layer --------------------------------------------------------------------------
def test_primes_and_roots_match_reference_constants():
    for q in PRIMES:
        assert oracle.is_prime(q)
        assert oracle.find_psi(q, 13) == PSI[q]
        assert oracle.mulmod(PSI[q], IPSI[q], q) == 1
        assert oracle.powmod(PSI[q], N, q) == q - 1


def test_barrett_mu_matches_shipped_vsetiq():
    assert oracle.barrett_mu(Q0) == MU[Q0]
    assert oracle.barrett_mu(Q1) == MU[Q1]
    # the special prime ships a rounded-up constant
    assert oracle.barrett_mu(P) + 1 == MU[P]


def test_rtl_barrett_is_exact_on_canonical_operands():
    """modmul_tb.sv:12-54,:193-214: random a, b < 2^60 against (a*b) % mod."""
    rnd = random.Random(1234)
    for q in PRIMES:
        for mu in (oracle.barrett_mu(q), MU[q]):
            for _ in range(20000):
                a, b = rnd.randrange(q), rnd.randrange(q)
                assert oracle.barrett_rtl(a, b, q, mu) == (a * b) % q
            for a, b in ((0, 0), (q - 1, q - 1), (1, q - 1), (q - 1, 1), (q - 1, 2)):
                assert oracle.barrett_rtl(a, b, q, mu) == (a * b) % q


def _model(op, a, b, s, q):
    """modalu_tb.sv:133-165 (AluFunc) in Python integers."""
    a, b, s = a % q, b % q, s % q
    half = lambda x: (x >> 1) + (((q + 1) >> 1) if x & 1 else 0)
    t = {0x00: a * b, 0x04: a * s, 0x01: a + b, 0x05: a + s, 0x02: a - b, 0x06: a - s, 0x0a: s - a,
         0x03: a, 0x15: a * b + s, 0x16: a * b - s, 0x1a: s - a * b, 0x11: (a - b) * s}
    if op in t:
        return t[op] % q, 0
    if op == 0x10:
        return (a + b * s) % q, (a - b * s) % q
    if op == 0x13:
        return half((a + b) % q), half(((a - b) * s) % q)
    raise KeyError(op)


def test_alu_opcodes_match_testbench_model():
    rnd = random.Random(99)
    ops = [0x00, 0x04, 0x01, 0x05, 0x02, 0x06, 0x0a, 0x03, 0x15, 0x16, 0x1a, 0x10, 0x13, 0x11]
    for q in PRIMES:
        for _ in range(3000):
            op = rnd.choice(ops)
            a, b, s = (rnd.randrange(1 << 60) for _ in range(3))   # modalu_tb.sv:82-92: < 2^60
            assert oracle.alu(op, a, b, s, q) == _model(op, a, b, s, q), hex(op)
        for op in ops:
            for a, b, s in ((0, 0, 0), (q - 1, q - 1, q - 1), (q, q, q), ((1 << 60) - 1,) * 3):
                assert oracle.alu(op, a, b, s, q) == _model(op, a, b, s, q)
    with pytest.raises(ValueError):
        oracle.alu(0x1f, 1, 2, 3, Q0)


def test_half_is_division_by_two():
    for q in PRIMES:
        inv2 = (q + 1) // 2
        for x in (0, 1, 2, 3, q - 1, q - 2, 12345678901234567):
            assert int(oracle.lib().orc_half(x, q)) == (x * inv2) % q


# ---- transforms ------------------------------------------------------------------------------
def test_ntt_golden_pair_from_model_kernel():
    m = manifest()["case1"]
    src = load(m["kernel1_pt_after_encode_fft_mod"]).reshape(2, N)
    exp = load(m["kernel1_pt_after_encode"]).reshape(2, N)
    for limb, q in enumerate((Q0, Q1)):
        got = oracle.ntt(src[limb], q)
        assert np.array_equal(got, exp[limb])
        assert np.array_equal(oracle.intt(got, q), src[limb])


def test_ntt_is_negacyclic_convolution_small():
    """Independent check of the transform definition: pointwise product == X^N+1 convolution."""
    logn, n = 10, 1024
    q = oracle.nth_prime(logn, 0)
    rng = np.random.default_rng(5)
    a = rng.integers(0, q, n, dtype=np.uint64)
    b = np.zeros(n, np.uint64); b[1] = 1; b[7] = 3            # X + 3 X^7
    fa, fb = oracle.ntt(a, q), oracle.ntt(b, q)
    prod = np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    c = oracle.intt(prod, q)
    exp = [0] * n
    for i in range(n):
        for sh, coef in ((1, 1), (7, 3)):
            j = i + sh
            v = int(a[i]) * coef
            if j >= n:
                j -= n; v = -v
            exp[j] = (exp[j] + v) % q
    assert [int(x) for x in c] == exp


def test_intt_halving_equals_scaled_gs():
    """per-stage halving == one multiplication by N^-1."""
    logn, n = 10, 1024
    q = oracle.nth_prime(logn, 3)
    psi = oracle.find_psi(q, logn)
    tab, itab = oracle.make_tables(q, psi, logn)
    rng = np.random.default_rng(6)
    x = [int(v) for v in rng.integers(0, q, n, dtype=np.uint64)]
    a = list(x)
    t, m = 1, n
    while m > 1:
        h = m // 2
        for i in range(h):
            w = int(itab[h + i])
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = (u + v) % q, ((u - v) * w) % q
        t *= 2; m = h
    ninv = pow(n, q - 2, q)
    exp = np.array([(v * ninv) % q for v in a], dtype=np.uint64)
    assert np.array_equal(oracle.intt(np.array(x, dtype=np.uint64), q), exp)


def test_vaut_and_vroli_definitions():
    n, logn, q = 1024, 10, Q0
    rng = np.random.default_rng(7)
    a = rng.integers(0, q, n, dtype=np.uint64); a[3] = 0
    for k in (1, 3, 9, 81, 6561 % n, n - 1):
        out = oracle.vaut(a, q, k)
        for i in (0, 1, 2, 3, 5, n - 1):
            t = (i * (k % n)) % (2 * n)
            exp = (q - int(a[i])) if t >= n else int(a[i])      # flipped zero is stored as q
            assert int(out[t % n]) == exp
    for s in (0, 1, 5, n - 1, n + 3):
        out = oracle.vroli(a, s)
        assert np.array_equal(out, np.roll(a, -(s % n)))


def test_vautn_extension_is_the_ntt_image_of_vaut():
    """The one ISA extension (funct6 011101): vautn permutes an NTT-form limb exactly as vaut acts on
    its coefficients -- vntt(vaut(x, k)) == vautn(vntt(x), k) -- checked against the PINNED vntt
    and vaut, for exponents below and above N, and through the ISA machine."""
    from aloha_amd import isa as _isa
    for logn, q in ((10, oracle.nth_prime(10, 0)), (13, Q0), (13, P)):
        n = 1 << logn
        x = np.random.default_rng(logn).integers(1, q, n, dtype=np.uint64)
        for k in (1, 3, 9, pow(3, 11, 2 * n), n + 1, 2 * n - 1):
            assert np.array_equal(oracle.ntt(oracle.vaut(x, q, k), q), oracle.vautn(oracle.ntt(x, q), k))
    n, q, k = 8192, Q1, pow(3, 4, 2 * 8192)
    x = np.random.default_rng(2).integers(1, q, n, dtype=np.uint64)
    vp = oracle.OracleVP()
    vp.write(0, 0, x)
    vp.load_isram(0, _isa.assemble(f"vsetvl {64 * n:#x}\nvsetq {q:#x}\nvle v0, src0+0x0\nvntt v2, v0\nvautn v4, v2, 0x0\n"
                                   f"vse v4, rslt+0x0\nbreak"))
    assert vp.run(0, src0=0, rslt=64, step=k) == 0
    assert np.array_equal(vp.read(0, 64, n), oracle.ntt(oracle.vaut(x, q, k), q))
    vp.load_isram(0, _isa.assemble(f"vsetvl {64 * n:#x}\nvsetq {q:#x}\nvautn v4, v2, 0x2\nbreak"))
    assert vp.run(0, step=0) == oracle.ERR_AUT                         # even exponent


# ---- ISA machine + harness on the golden chains -------------------------------------------------
def _bench(images):
    vp = oracle.OracleVP()
    tb = harness.TestBench(vp)
    tb.load_programs(images)
    return vp, tb


def _own_images():
    mod = (Q0, Q1)
    return {"encode_post": programs.assemble(programs.encode_post(N, mod)),
            "mul_plain": programs.assemble(programs.mul_plain(N, mod)),
            "hom_add": programs.assemble(programs.hom_add(N, mod)),
            "keyswitch": programs.assemble(programs.keyswitch(N, mod, P))}


def run_case0(vp, tb):
    """The recovered 8-op chain of tv/case0_4_4; returns {k: dump}."""
    m = manifest()["case0"]
    tb.load_ksk([load(m["ksk_step2"])])
    tb.dram_ct[0] = load(m["inst_0_out"])
    tb.dram_encoder[0x0] = load(m["inst_2_0_out"])
    tb.dram_encoder[0x40000] = load(m["inst_3_0_out"])
    H = harness
    ops = [H.Op(H.OP_LOAD, 0x000, dram_addr=0), H.Op(H.OP_ROTATE, 0x100, src1=0x000, step=2),
           H.Op(H.OP_ENCODE, 0x200, dram_addr=0x0), H.Op(H.OP_ENCODE, 0x280, dram_addr=0x40000),
           H.Op(H.OP_MUL_PLAIN, 0x300, src1=0x000, src2=0x200),
           H.Op(H.OP_MUL_PLAIN, 0x400, src1=0x100, src2=0x280),
           H.Op(H.OP_HOM_ADD, 0x500, src1=0x300, src2=0x400), H.Op(H.OP_STORE, 0x500, dram_addr=0x1000)]
    dumps = {}
    tb.dump = lambda k, sub, data: dumps.__setitem__(f"{k}{sub}", data)
    tb.run(ops)
    return dumps


def check_case0(dumps):
    m = manifest()["case0"]
    for k in range(1, 8):
        exp = load(m[f"inst_{k}_out"])
        got = dumps[str(k)][:exp.size]            # encode dumps: only the numeric prefix is defined
        assert np.array_equal(got, exp), f"case0 inst_{k}_out"


@pytest.mark.parametrize("which", ["reference_images", "own_programs"])
def test_case0_chain(which):
    vp, tb = _bench(reference_images() if which == "reference_images" else _own_images())
    check_case0(run_case0(vp, tb))


def run_case2(vp, tb):
    m = manifest()["case2"]
    ops = [harness.parse_op(a) for a in m["oplist"]]
    # key slots: step 2 -> slot 0, step 4 -> slot 1 (unused by this case), step 8 -> slot 2
    tb.load_ksk([load(m["ksk_step2"]), np.zeros(12 * N, np.uint64), load(m["ksk_step8"])])
    tb.dram_ct[0] = load(m["inst_0_out"])
    for k, e in m["encoder_out"].items():
        tb.dram_encoder[ops[int(k)].dram_addr] = load(e)
    dumps = {}
    tb.dump = lambda k, sub, data: dumps.__setitem__(f"{k}{sub}", data)
    tb.run(ops)
    return dumps, tb


def check_case2(dumps, tb):
    m = manifest()["case2"]
    for k, digest in m["expected_sha256"].items():
        words = m["expected_words"][k]
        assert sha(dumps[k][:words]) == digest, f"case2 inst_{k}_out"
    final = load(m["case3_expected_result"])
    assert np.array_equal(dumps["28"], final)                  # check_result (:640-683)


@pytest.mark.parametrize("which", ["reference_images", "own_programs"])
def test_case2_chain_29_ops(which):
    vp, tb = _bench(reference_images() if which == "reference_images" else _own_images())
    check_case2(*run_case2(vp, tb))


@pytest.mark.parametrize("case,kern,step,slot_keys", [
    ("case0", "kernel1", 2, ["ksk_step2"]),
    ("case1", "kernel12", 4, [None, "ksk_step4"]),
])
def test_model_kernel_rotates(case, kern, step, slot_keys):
    m = manifest()[case]
    vp, tb = _bench(reference_images())
    tb.load_ksk([load(m[k]) if k else np.zeros(12 * N, np.uint64) for k in slot_keys])
    vp.write(harness.SPM, 0, load(m[f"{kern}_ct_before_rotate"]))
    tb.run_rotate(0x100, 0x000, step)
    assert np.array_equal(tb.copy_spm(0x100), load(m[f"{kern}_ct_after_rotate"]))


def test_own_program_images_differ_only_in_allocation():
    """The own generators reproduce the three simple shipped images word for word."""
    own, ref = _own_images(), reference_images()
    for name in ("encode_post", "mul_plain", "hom_add"):
        assert np.array_equal(own[name], ref[name]), name
    assert isa.parse_mem_text(isa.to_mem_text(ref["keyswitch"])).tolist() == ref["keyswitch"].tolist()


# ---- error behaviour of the machine ------------------------------------------------------------
def test_machine_rejects_what_the_rtl_leaves_undefined():
    vp = oracle.OracleVP(spm_rows=1024, ksk_rows=64)
    asm = lambda t: isa.assemble(t)
    vp.load_isram(0, asm("vsetvl 0x80000\nvsetq %#x\nvfqadd.vv v2, v0, v4\nbreak" % Q0))
    assert vp.run(0) == oracle.ERR_BANK
    vp.load_isram(0, asm("vsetvl 0x80001\nbreak"))
    assert vp.run(0) == oracle.ERR_VL
    vp.load_isram(0, asm("vsetvl 0x80000\nvsetq 0x123\nbreak"))
    assert vp.run(0) == oracle.ERR_MODULUS
    vp.load_isram(0, asm("vsetvl 0x80000\nvle v0, src0+0x0\nbreak"))
    assert vp.run(0, src0=1000) == oracle.ERR_ADDR
    vp.load_isram(0, np.array([[0xFC00000B, 0, 0]], dtype=np.uint32))
    assert vp.run(0) == oracle.ERR_DECODE
    vp.load_isram(0, asm("vsetvl 0x80000\nvsetq %#x\nvaut v2, v0, 0x2\nbreak" % Q0))
    assert vp.run(0, step=0) == oracle.ERR_AUT
    vp.load_isram(0, asm("vsetvl 0x10000\nvsetq %#x\nvle v0, src0+0x0\nvntt v2, v0\nbreak" % P))
    assert vp.run(0) == oracle.OK                                # N = 1024: P has 2-adic valuation 14
    vp.load_isram(0, asm("vsetvl 0x100000\nvsetq %#x\nvle v0, src0+0x0\nvntt v2, v0\nbreak" % P))
    assert vp.run(0) == oracle.ERR_ROOT                          # N = 16384 > 8192 impossible for P


# ---- full sweep of the reference's tv/ tree (only where it exists) ------------------------------
@pytest.mark.reference
@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present")
def test_full_tv_sweep():
    tv = os.path.join(REFERENCE, "tv")
    counts = {"mulplain": 0, "homadd": 0, "encode": 0, "rotate": 0}
    for case in sorted(os.listdir(tv)):
        d = os.path.join(tv, case)
        if not os.path.isdir(d):
            continue
        for f in sorted(glob.glob(os.path.join(d, "kernel*_ct_after_mulplain.txt"))):
            k = os.path.basename(f).split("_")[0]
            ct = read_tv(os.path.join(d, f"{k}_ct_before_mulplain.txt")).reshape(2, 2, N)
            pt = read_tv(os.path.join(d, f"{k}_pt_before_mulplain.txt")).reshape(2, N)
            exp = read_tv(f).reshape(2, 2, N)
            for c in range(2):
                for j, q in enumerate((Q0, Q1)):
                    got = (ct[c, j].astype(object) * pt[j].astype(object)) % q
                    assert np.array_equal(got.astype(np.uint64), exp[c, j])
            counts["mulplain"] += 1
        for f in sorted(glob.glob(os.path.join(d, "kernel*_ct_after_homadd.txt"))):
            k = os.path.basename(f).split("_")[0]
            a = read_tv(os.path.join(d, f"{k}_ct_before_homaddct1.txt")).reshape(2, 2, N)
            b = read_tv(os.path.join(d, f"{k}_ct_before_homaddct2.txt")).reshape(2, 2, N)
            exp = read_tv(f).reshape(2, 2, N)
            for c in range(2):
                for j, q in enumerate((Q0, Q1)):
                    got = (a[c, j].astype(object) + b[c, j].astype(object)) % q
                    assert np.array_equal(got.astype(np.uint64), exp[c, j])
            counts["homadd"] += 1
        for f in sorted(glob.glob(os.path.join(d, "kernel*_pt_after_encode_fft_mod.txt"))):
            k = os.path.basename(f).split("_")[0]
            src = read_tv(f).reshape(2, N)
            exp = read_tv(os.path.join(d, f"{k}_pt_after_encode.txt")).reshape(2, N)
            for j, q in enumerate((Q0, Q1)):
                assert np.array_equal(oracle.ntt(src[j], q), exp[j])
            counts["encode"] += 1
        for f in sorted(glob.glob(os.path.join(d, "kernel*_ct_after_rotate.txt"))):
            k = os.path.basename(f).split("_")[0]
            before = read_tv(os.path.join(d, f"{k}_ct_before_rotate.txt"))
            after = read_tv(f)
            hit = None
            for step in (2, 4, 8):
                kp = os.path.join(d, f"ksk_step{step}.txt")
                if not os.path.exists(kp):
                    continue
                vp, tb = _bench(reference_images())
                keys = [np.zeros(12 * N, np.uint64)] * 3
                keys[harness.clog2(step) - 1] = read_tv(kp)
                tb.load_ksk(keys)
                vp.write(harness.SPM, 0, before)
                tb.run_rotate(0x100, 0, step)
                if np.array_equal(tb.copy_spm(0x100), after):
                    hit = step
                    break
            assert hit is not None, f"{case}/{k}: no shipped key reproduces the rotate"
            counts["rotate"] += 1
        # RTL encode dumps -> RTL NTT outputs
        for f in sorted(glob.glob(os.path.join(d, "rtl_result", "inst_*_0_out.txt"))):
            k = os.path.basename(f).split("_")[1]
            src = read_tv(f)[:2 * N].reshape(2, N)
            exp = read_tv(os.path.join(d, "rtl_result", f"inst_{k}_out.txt"))[:2 * N].reshape(2, N)
            for j, q in enumerate((Q0, Q1)):
                assert np.array_equal(oracle.ntt(src[j], q), exp[j])
    assert counts == {"mulplain": 14, "homadd": 11, "encode": 12, "rotate": 7}, counts
