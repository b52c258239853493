"""(Run by tests/test_experiments.py in a process of its own, with ALOHA_VP_LIB = the experiments variant of the library.)
GPU parity of the fused single-launch forward transform at N = 2^16 (csrc/experiments/ntt_tile_knobs.hip: ntt_fused_fwd_kernel --
column and row tiles in ONE persistent launch, the intermediate handed over through the Infinity Cache) against the
oracle and against the two-launch form of the same engine.  Bit-exact.  Reference: vntt, one pass over the register
file (src/vp/ntt/ntt_fsm.sv:49-81, modalu.sv:160-165); results per SURVEY A.2.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/ (util, conftest)

import oracle  # noqa: E402
from aloha_amd import isa  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

pytestmark = pytest.mark.gpu

SPM = 0
LOGN, N = 16, 1 << 16
ROWS = N // 128


def _engine(fused_min, rows, flags=0, delay=None, wgs=None, mode="dispatch"):
    old = {k: os.environ.get(k) for k in ("ALOHA_NTT_FUSED", "ALOHA_NTT_FUSED_DELAY", "ALOHA_NTT_FUSED_WGS", "ALOHA_NTT_FUSED_MODE")}
    os.environ["ALOHA_NTT_FUSED"] = str(fused_min)
    os.environ["ALOHA_NTT_FUSED_MODE"] = mode
    if delay is not None:
        os.environ["ALOHA_NTT_FUSED_DELAY"] = str(delay)
    if wgs is not None:
        os.environ["ALOHA_NTT_FUSED_WGS"] = str(wgs)
    try:
        return avp.VP(spm_rows=rows, ksk_rows=16, max_log_n=16, flags=flags)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _data(rng, mods, pattern="random"):
    out = np.empty((len(mods), N), dtype=np.uint64)
    for i, q in enumerate(mods):
        if pattern == "random":
            a = rng.integers(0, q, N, dtype=np.uint64)
            a[:4] = [0, 1, q - 1, q - 2]
        elif pattern == "max":
            a = np.full(N, q - 1, dtype=np.uint64)
        else:                                   # non-canonical memory words: anything below 2^60 (modalu.sv:44-46)
            a = rng.integers(0, 1 << 60, N, dtype=np.uint64)
            a[:2] = [(1 << 60) - 1, q]
        out[i] = a
    return out


@pytest.mark.parametrize("mode", ["dispatch", "persistent"])
@pytest.mark.parametrize("per_prime,nprimes,inplace,pattern", [
    (16, 2, True, "random"),       # whole 16-limb bundles
    (28, 2, True, "random"),       # bundles of 16 + 8 + 4
    (12, 3, False, "random"),      # 8 + 4, out of place
    (4, 5, True, "max"),           # 4-limb bundles only, every coefficient q - 1
    (20, 2, False, "wide"),        # non-canonical inputs (< 2^60)
])
def test_fused_forward_equals_oracle(per_prime, nprimes, inplace, pattern, mode):
    rng = np.random.default_rng(per_prime * 100 + nprimes)
    mods = [avp.nth_prime(LOGN, 3 * i) for i in range(nprimes) for _ in range(per_prime)]
    data = _data(rng, mods, pattern)
    count = len(mods)
    with _engine(4, 2 * count * ROWS, mode=mode) as vp:
        vp.write(SPM, 0, data.reshape(-1))
        dst = 0 if inplace else count * ROWS
        vp.ntt_rows(SPM, dst, 0, LOGN, mods)
        got = vp.read(SPM, dst, count * N).reshape(count, N)
        assert vp.stats()["fused_launches"] == 1
        for i, q in enumerate(mods):
            assert np.array_equal(got[i], oracle.ntt(data[i] % np.uint64(q) if pattern == "wide" else data[i], q)), f"limb {i}"
        if not inplace:
            assert np.array_equal(vp.read(SPM, 0, count * N).reshape(count, N), data)
        # the same plan again (cached schedule, counters re-zeroed) on fresh data, then back through the inverse
        data2 = _data(rng, mods)
        vp.write(SPM, 0, data2.reshape(-1))
        vp.ntt_rows(SPM, dst, 0, LOGN, mods)
        got2 = vp.read(SPM, dst, count * N).reshape(count, N)
        for i in (0, count // 2, count - 1):
            assert np.array_equal(got2[i], oracle.ntt(data2[i], mods[i]))
        vp.ntt_rows(SPM, dst, dst, LOGN, mods, inverse=True)
        assert np.array_equal(vp.read(SPM, dst, count * N).reshape(count, N), data2)
        assert vp.stats()["fused_launches"] == 2


@pytest.mark.parametrize("delay,wgs,mode", [(1, 4, "persistent"), (2, 1, "persistent"), (8, 3, "persistent"), (1, 0, "dispatch"), (3, 0, "dispatch")])
def test_fused_schedule_parameters_do_not_change_results(delay, wgs, mode):
    """Scheduling distance and persistent-grid size only move work around: a distance of ONE group (row tiles drawn
    right behind their column tiles: the waits actually spin) and one workgroup per CU give the same words."""
    rng = np.random.default_rng(7 + delay)
    mods = [avp.nth_prime(LOGN, i) for i in range(6) for _ in range(16)]
    data = _data(rng, mods)
    count = len(mods)
    with _engine(4, count * ROWS, delay=delay, wgs=wgs or None, mode=mode) as vp, _engine(0, count * ROWS) as ref:
        for e in (vp, ref):
            e.write(SPM, 0, data.reshape(-1))
            e.ntt_rows(SPM, 0, 0, LOGN, mods)
        assert vp.stats()["fused_launches"] == 1 and ref.stats()["fused_launches"] == 0
        got, exp = vp.read(SPM, 0, count * N), ref.read(SPM, 0, count * N)
        assert np.array_equal(got, exp)
        assert np.array_equal(got.reshape(count, N)[37], oracle.ntt(data[37], mods[37]))


@pytest.mark.parametrize("mode", ["dispatch", "persistent"])
def test_fused_large_batch_equals_two_launches_and_round_trips(mode):
    """512 limbs (256 MiB: the size of the Infinity Cache) through the default settings: fused == two launches word
    for word, oracle spot checks, inverse(forward(x)) == x, linearity of the transform."""
    rng = np.random.default_rng(99)
    primes = [avp.nth_prime(LOGN, i) for i in range(8)]
    mods = [q for q in primes for _ in range(64)]
    count = len(mods)
    data = _data(rng, mods)
    with _engine(256, count * ROWS, mode=mode) as vp, _engine(256, count * ROWS, flags=avp.FLAG_NTT_TWO_PASS) as ref:
        for e in (vp, ref):
            e.write(SPM, 0, data.reshape(-1))
            e.ntt_rows(SPM, 0, 0, LOGN, mods)
        assert vp.stats()["fused_launches"] == 1 and ref.stats()["fused_launches"] == 0
        got = vp.read(SPM, 0, count * N).reshape(count, N)
        assert np.array_equal(got, ref.read(SPM, 0, count * N).reshape(count, N))
        for i in (0, 63, 64, 255, 300, 511):
            assert np.array_equal(got[i], oracle.ntt(data[i], mods[i]))
        # linearity inside one prime: NTT(a) + NTT(b) == NTT(a + b)
        q = primes[0]
        s = (data[0].astype(object) + data[1].astype(object)) % q
        assert np.array_equal((got[0].astype(object) + got[1].astype(object)) % q, oracle.ntt(np.array(s, dtype=np.uint64), q).astype(object))
        vp.ntt_rows(SPM, 0, 0, LOGN, mods, inverse=True)
        assert np.array_equal(vp.read(SPM, 0, count * N).reshape(count, N), data)


def test_small_or_ragged_batches_keep_the_two_launch_form():
    """Below the threshold, or with a prime that has fewer than 4 adjacent limbs (no bundle tile), nothing changes."""
    rng = np.random.default_rng(5)
    q0, q1 = avp.nth_prime(LOGN, 0), avp.nth_prime(LOGN, 1)
    for mods, fused_min in (([q0] * 8, 256), ([q0] * 16 + [q1] * 3, 4)):
        data = _data(rng, mods)
        with _engine(fused_min, len(mods) * ROWS) as vp:
            vp.write(SPM, 0, data.reshape(-1))
            vp.ntt_rows(SPM, 0, 0, LOGN, mods)
            assert vp.stats()["fused_launches"] == 0
            got = vp.read(SPM, 0, len(mods) * N).reshape(len(mods), N)
            for i, q in enumerate(mods):
                assert np.array_equal(got[i], oracle.ntt(data[i], q))


def test_recorded_runs_replayed_as_hipgraphs(monkeypatch):
    """ALOHA_RUN_GRAPH=1 (variant builds): a recorded run becomes a hipGraph on its second replay; results stay those of
    the oracle's ISA machine."""
    monkeypatch.setenv("ALOHA_RUN_GRAPH", "1")
    logn, n = 12, 1 << 12
    rows = n // 128
    q = oracle.nth_prime(13, 0)
    prog = [f"vsetvl {64 * n:#x}", f"vsetq {q:#x}"]
    for i in range(4):
        prog += [f"vle v0, src0+{i * 8 * n:#x}", "vntt v2, v0", f"vse v2, rslt+{i * 8 * n:#x}"]
    prog += [f"vle v0, rslt+0x0", f"vle v1, rslt+{8 * n:#x}", "vfqmul.vv v2, v0, v1", f"vse v2, rslt+{4 * 8 * n:#x}", "break"]
    words = isa.assemble("\n".join(prog))
    rng = np.random.default_rng(3)
    with avp.VP(spm_rows=16 * rows, ksk_rows=1, max_log_n=logn) as vp:
        o = oracle.OracleVP(spm_rows=16 * rows, ksk_rows=1)
        vp.load_isram(0, words); o.load_isram(0, words)
        for rep in range(4):
            data = rng.integers(0, q, 4 * n, dtype=np.uint64)
            for e in (vp, o):
                e.write(SPM, 0, data)
                assert e.run(0, src0=0, rslt=8 * rows) == 0
                e.sync()
            assert np.array_equal(vp.read(SPM, 8 * rows, 5 * n), o.read(SPM, 8 * rows, 5 * n)), rep
        assert vp.stats()["graph_launches"] >= 1
