"""The CKKS encoder block (src/encoder): oracle restatement pinned against the
reference's golden encode pairs, and the FP64 HIP kernels against the oracle and the goldens.

Floating point: the tolerance is written where it is used.  The golden cleartexts carry six
significant digits, so the software model's coefficients are reproduced to about 5e-7 of the
largest coefficient, never exactly; GPU vs oracle differ only in FFT summation order (at most
one unit in a rounded coefficient, on a handful of coefficients)."""
import glob
import os

import numpy as np
import pytest

import oracle
from oracle import ckks_encode as ce
from util import HAVE_REFERENCE, N, Q0, Q1, REFERENCE, load, manifest

GOLDEN_REL_TOL = 1e-6          # 6-digit cleartexts: measured 3.6e-7 .. 6.9e-7 over the 12 pairs of tv/


def cleartext(ent) -> np.ndarray:
    z = np.zeros(ent["n"], dtype=np.float64)
    for i, v in ent["nonzero"]:
        z[i] = v
    return z


def check_against_fft_mod(limbs, gold):
    """limbs: (2, N) residues mod q0, q1; gold: the model's 2N words."""
    for l, q in enumerate((Q0, Q1)):
        g = ce.centred(gold[l * N:(l + 1) * N], q)
        m = ce.centred(limbs[l], q)
        assert np.abs(m - g).max() <= GOLDEN_REL_TOL * np.abs(g).max()


ENC_KEYS = [k for k in manifest()["encode"] if "fft_mod" in manifest()["encode"][k]]


@pytest.mark.parametrize("key", ENC_KEYS)
def test_oracle_encode_matches_model_fft_mod(key):
    ent = manifest()["encode"][key]
    limbs = ce.encode(cleartext(ent["cleartext"]), 13, [Q0, Q1])
    check_against_fft_mod(limbs, load(ent["fft_mod"]))


def test_oracle_encode_then_ntt_matches_model_plaintext_case0():
    """case0 ships only the NTT-domain plaintext: compare in the coefficient domain through the
    (bit-exact, pinned) inverse transform."""
    ent = manifest()["encode"]["case0/kernel2"]
    limbs = ce.encode(cleartext(ent["cleartext"]), 13, [Q0, Q1])
    gold = load(ent["ntt"])
    back = np.stack([oracle.intt(gold[:N], Q0), oracle.intt(gold[N:], Q1)])
    check_against_fft_mod(limbs, back.reshape(-1))


def test_npoint_derivation_equals_the_definition():
    rng = np.random.default_rng(5)
    for logn in (10, 12, 13):
        z = rng.uniform(-1, 1, 1 << logn)
        assert np.array_equal(ce.encode_coeffs(z, logn), ce.encode_direct(z, logn))


def test_encode_is_an_embedding_round_trip():
    """m(zeta^(3^j)) / Delta gives the slots back (to 2^-38-ish), conjugates at -3^j."""
    logn, Nn = 11, 1 << 11
    rng = np.random.default_rng(6)
    z = rng.uniform(-1, 1, Nn)
    m = ce.encode_coeffs(z, logn).astype(np.float64)
    zeta = np.exp(1j * np.pi / Nn)
    e = 1
    for j in range(16):
        val = np.sum(m * zeta ** ((e * np.arange(Nn)) % (2 * Nn))) / 2.0 ** 38
        assert abs(val - (z[2 * j] + 1j * z[2 * j + 1])) < 1e-7
        e = e * 3 % (2 * Nn)


def test_oracle_decode_inverts_encode_and_reads_the_golden_plaintexts():
    """decode(encode(z)) = z to rounding (|coefficient error| <= 1/2 spread over n slots / 2^38), and the
    golden model plaintexts decode to their 6-digit cleartexts."""
    rng = np.random.default_rng(11)
    for logn in (10, 13):
        z = rng.uniform(-1, 1, 1 << logn)
        back = ce.decode(ce.encode(z, logn, [Q0])[0], logn, Q0)
        assert np.abs(back - z).max() < 1e-7
    for key in ENC_KEYS:
        ent = manifest()["encode"][key]
        z = cleartext(ent["cleartext"])
        back = ce.decode(load(ent["fft_mod"])[:N], 13, Q0)
        assert np.abs(back - z).max() < 2e-6                            # six printed digits


@pytest.mark.reference
@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present")
def test_oracle_encode_full_tv_sweep():
    files = sorted(glob.glob(os.path.join(REFERENCE, "tv", "case*", "kernel*_cleartext_before_encode_re_im.txt")))
    assert len(files) == 14
    checked = 0
    for f in files:
        g = f.replace("cleartext_before_encode_re_im", "pt_after_encode_fft_mod")
        if not os.path.exists(g):
            continue
        limbs = ce.encode(np.loadtxt(f), 13, [Q0, Q1])
        gold = np.array([int(t) for t in open(g).read().split()], dtype=np.uint64)
        check_against_fft_mod(limbs, gold)
        checked += 1
    assert checked == 12


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def engine():
    from aloha_amd import vp as avp
    with avp.VP(spm_rows=1 << 16, ksk_rows=16, max_log_n=16) as vp:
        yield vp


def gpu_vs_oracle(engine, z, logn, mods, scale=2.0 ** 38):
    from aloha_amd import vp as avp
    n = 1 << logn
    z = np.asarray(z, dtype=np.float64).reshape(-1, n)
    engine.encode(avp.SPM, 0, z, logn, mods, scale)
    got = engine.read(avp.SPM, 0, z.shape[0] * len(mods) * n).reshape(z.shape[0], len(mods), n)
    worst, ndiff = 0.0, 0
    for p in range(z.shape[0]):
        ref = ce.encode(z[p], logn, mods, scale)
        for l, q in enumerate(mods):
            d = np.abs(ce.centred(got[p, l], q) - ce.centred(ref[l], q))
            worst = max(worst, d.max())
            ndiff += int(np.count_nonzero(d))
    return got, worst, ndiff


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [10, 11, 12, 13, 14, 15, 16])
def test_gpu_encode_matches_oracle_all_sizes(engine, logn):
    """Same arithmetic, different summation order: a rounded coefficient may differ by one unit
    when its fractional part sits within ~1e-7 of one half -- allowed on at most 4 coefficients."""
    from aloha_amd import vp as avp
    rng = np.random.default_rng(100 + logn)
    mods = [avp.nth_prime(logn, 0), avp.nth_prime(logn, 1), avp.nth_prime(logn, 2)]
    z = rng.uniform(-1, 1, (2, 1 << logn))
    _, worst, ndiff = gpu_vs_oracle(engine, z, logn, mods)
    assert worst <= 1.0 and ndiff <= 4, (worst, ndiff)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ENC_KEYS)
def test_gpu_encode_matches_model_fft_mod(engine, key):
    ent = manifest()["encode"][key]
    got, worst, _ = gpu_vs_oracle(engine, cleartext(ent["cleartext"]), 13, [Q0, Q1])
    assert worst <= 1.0
    check_against_fft_mod(got[0], load(ent["fft_mod"]))


@pytest.mark.gpu
def test_gpu_encode_then_encode_post_matches_model_plaintext(engine):
    """cleartext -> encoder -> the reference's encode_post program (NTT), compared with the
    model's NTT-domain plaintext in the coefficient domain (after the engine's own inverse)."""
    from aloha_amd import isa
    from aloha_amd import vp as avp
    from util import reference_images
    ent = manifest()["encode"]["case0/kernel2"]
    engine.encode(avp.SPM, 0x200, cleartext(ent["cleartext"]), 13, [Q0, Q1])
    engine.load_isram(isa.PC_ENCODE_POST, reference_images()["encode_post"])
    assert engine.run(isa.PC_ENCODE_POST, src0=0x200, rslt=0x200) == 0
    engine.sync()
    ntt_form = engine.read(avp.SPM, 0x200, 2 * N)
    gold = load(ent["ntt"])
    # NTT outputs of nearly equal inputs are unrelated word by word: compare after inverting both
    back_gold = np.stack([oracle.intt(gold[:N], Q0), oracle.intt(gold[N:], Q1)]).reshape(-1)
    engine.ntt_rows(avp.SPM, 0x200, 0x200, 13, [Q0, Q1], inverse=True)
    back = engine.read(avp.SPM, 0x200, 2 * N).reshape(2, N)
    check_against_fft_mod(back, back_gold)
    assert np.array_equal(np.stack([oracle.intt(ntt_form[:N], Q0), oracle.intt(ntt_form[N:], Q1)]), back)


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [10, 12, 13, 14, 16])
def test_gpu_decode_matches_oracle_and_inverts_encode(engine, logn):
    """Tolerance: 1e-9 absolute against the oracle on slots of magnitude <= 1 (same arithmetic, different
    summation order); the round trip returns the cleartext to 1e-7 (coefficient rounding at scale 2^38)."""
    from aloha_amd import vp as avp
    n = 1 << logn
    rng = np.random.default_rng(200 + logn)
    q = avp.nth_prime(logn, 0)
    z = rng.uniform(-1, 1, (3, n))
    engine.encode(avp.SPM, 0, z, logn, [q, avp.nth_prime(logn, 1)])       # 2 limbs per plaintext: stride 2
    got = engine.decode(avp.SPM, 0, logn, q, nplain=3, limb_stride=2)
    limbs = engine.read(avp.SPM, 0, 6 * n).reshape(3, 2, n)
    for p in range(3):
        assert np.abs(got[p] - ce.decode(limbs[p, 0], logn, q)).max() < 1e-9
        assert np.abs(got[p] - z[p]).max() < 1e-7


@pytest.mark.gpu
def test_gpu_decode_reads_the_golden_plaintext(engine):
    from aloha_amd import vp as avp
    ent = manifest()["encode"]["case2/kernel13"]
    engine.write(avp.SPM, 0, load(ent["fft_mod"]))
    got = engine.decode(avp.SPM, 0, 13, Q0)[0]
    assert np.abs(got - cleartext(ent["cleartext"])).max() < 2e-6


@pytest.mark.gpu
def test_gpu_encode_argument_checks(engine):
    from aloha_amd import vp as avp
    z = np.zeros(8192)
    for bad in (lambda: engine.encode(avp.SPM, 0, z, 13, [Q0 + 2]),
                lambda: engine.encode(avp.SPM, engine.spm_rows - 1, z, 13, [Q0]),
                lambda: engine.encode(avp.SPM, 0, np.zeros(512), 9, [Q0])):
        with pytest.raises(avp.AlohaError):
            bad()
