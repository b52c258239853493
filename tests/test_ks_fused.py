"""The fused key-switch phases at N = 2^16 (csrc/ntt_ks.hip; the staged batches of the sequencer, csrc/vp_engine.cpp):
basis-conversion sums folded into the column pass of the transform that follows them, the inner product with the key and
the mod-down's last sum folded into the row pass (keyswitch.mem:9-14 -> :10,:14 -> :46-79; :86-91 -> :94-99 -> :102-117).
Every result must equal the oracle's ISA machine, which executes the same programs one instruction at a time, and the
engine's own unfused execution -- with and without the scratch window that lets the engine keep the transformed digits
out of memory."""
import numpy as np
import pytest

import oracle
from aloha_amd import hybrid, isa, keyswitch
from aloha_amd import vp as avp

pytestmark = pytest.mark.gpu
SPM, KSK = 0, 1
LOGN, N = 16, 1 << 16
ROWS = N // 128


def _case(L, K, seed):
    mods = [oracle.nth_prime(LOGN, i) for i in range(L)]
    specials = [oracle.nth_prime(LOGN, L + k) for k in range(K)]
    rng = np.random.default_rng(seed)
    ct = np.stack([rng.integers(0, mods[j % L], N, dtype=np.uint64) for j in range(2 * L)])
    ct[0, :3] = [0, mods[0] - 1, 1]
    return mods, specials, ct


def _run(vp, mods, specials, alpha, ct, op, batch=1, hint=True):
    L = len(mods)
    lo = hybrid.HybridLayout(N, mods, specials, alpha, batch=batch)
    if not hint:
        vp.set_scratch = lambda *a: None                           # the host never declares a window: every transformed limb is stored
    hs = hybrid.HybridKeySwitch(vp, N, mods, specials, alpha, batch=batch)
    for jl, j in enumerate(lo.owned_all):
        vp.write(KSK, jl * lo.dnum * 2 * lo.rows, np.random.default_rng(900 + j).integers(0, lo.mods_all[j], lo.dnum * 2 * N, dtype=np.uint64))
    outs = []
    for b in range(batch):
        hs.load_ct(np.roll(ct, b, axis=1), b); hs.load_ct_b(np.roll(ct[::-1], 3 * b, axis=1).copy(), b)
    if op == "rotate":
        hs.rotate(pow(3, 5, 2 * N))
    elif op == "multiply":
        hs.multiply()
    else:
        hs.rescale()
    for b in range(batch):
        outs.append(hs.read_out(b)[:, :L - 1] if op == "rescale" else hs.read_out(b))
    sd = vp.read(SPM, lo.row(lo.sd), lo.Mr * lo.dnum * batch * N)
    return np.stack(outs), sd


@pytest.mark.parametrize("L,K,alpha,batch", [
    (6, 2, 2, 1),      # dnum 3: two transformed digits + the own one per ciphertext modulus, three per special prime
    (9, 1, 1, 1),      # the reference's decomposition: 8 transformed digits (two groups of four) + 1, and 9 = 2 x 4 + 1
    (10, 2, 2, 2),     # dnum 5: exactly four transformed + own; five under the special primes; two ciphertexts per call
    (7, 1, 1, 1),      # 8 groups: the one-row-per-wave shape of the inner-product kernel (a rank's share at W = 8) on a group of four
                       # transformed digits, a remainder tile with two of its four lines live and a plain digit alone in its tile
])
def test_fused_phases_equal_oracle_and_unfused(L, K, alpha, batch, monkeypatch):
    mods, specials, ct = _case(L, K, 10 * L + alpha)
    lo = hybrid.HybridLayout(N, mods, specials, alpha, batch=batch)
    sizes = dict(spm_rows=lo.spm_rows_needed, ksk_rows=lo.ksk_limbs * lo.rows)
    for op in ("multiply", "rotate", "rescale"):
        exp, _ = _run(oracle.OracleVP(**sizes), mods, specials, alpha, ct, op, batch)
        with avp.VP(max_log_n=16, **sizes) as vp:
            got, _ = _run(vp, mods, specials, alpha, ct, op, batch)
            st = vp.stats()
        assert np.array_equal(got, exp), (op, "fused, scratch window")
        assert st["ks_fused_launches"] >= 2, (op, st)              # column pass with sums + row pass with epilogue / inner product
        with avp.VP(max_log_n=16, **sizes) as vp:
            got2, sd2 = _run(vp, mods, specials, alpha, ct, op, batch, hint=False)
            assert vp.stats()["ks_fused_launches"] >= 2
        assert np.array_equal(got2, exp), (op, "fused, no window")
        monkeypatch.setenv("ALOHA_NO_STAGES", "1")
        with avp.VP(max_log_n=16, **sizes) as vp:
            got3, sd3 = _run(vp, mods, specials, alpha, ct, op, batch)
            assert vp.stats()["ks_fused_launches"] == 0
        monkeypatch.delenv("ALOHA_NO_STAGES")
        assert np.array_equal(got3, exp), (op, "stages off")
        if op != "rescale":
            assert np.array_equal(sd2, sd3), (op, "without a window the lifted digits in memory are those of the unfused run")


def _program(lines):
    return isa.assemble("\n".join([f"vsetvl {64 * N:#x}"] + lines + ["break"]))


def _sum(q, srcs, scalars, dst, const=None):
    out = []
    for t, (s, c) in enumerate(zip(srcs, scalars)):
        out += [f"vle v0, abs+{s * 8 * N:#x}"] + ([f"vfqmul.vs v6, v0, {c:#x}"] if t == 0 else [f"vfqmul.vs v3, v0, {c:#x}", "vfqadd.vv v6, v6, v3"])
    if const is not None:
        out.append(f"vfqadd.vs v6, v6, {const:#x}")
    return out + [f"vse v6, abs+{dst * 8 * N:#x}"]


def _ntt(slot, dst=None):
    return [f"vle v0, abs+{slot * 8 * N:#x}", "vntt v2, v0", f"vse v2, abs+{(slot if dst is None else dst) * 8 * N:#x}"]


def _dot(xs, ks, dst):
    out = []
    for t, (x, k) in enumerate(zip(xs, ks)):
        out += [f"vle v0, abs+{x * 8 * N:#x}", f"vle v1, abs+{k * 8 * N:#x}"]
        out += ["vfqmul.vv v8, v0, v1"] if t == 0 else ["vfqmul.vv v3, v0, v1", "vfqadd.vv v8, v8, v3"]
    return out + [f"vse v8, abs+{dst * 8 * N:#x}"]


def _ew(a, b, dst):
    return [f"vle v0, abs+{a * 8 * N:#x}", f"vle v1, abs+{b * 8 * N:#x}", "vfqmul.vv v2, v0, v1", f"vse v2, abs+{dst * 8 * N:#x}"]


def test_chains_the_kernels_do_not_cover_fall_back_exactly():
    """Hand-written sum -> transform -> sum chains with the irregularities the matcher must refuse or handle: a sum whose
    operand another transform of the batch overwrites, an out-of-place transform of a sum, a later load of a transformed
    limb inside the scratch window, a transformed limb shared by two inner products, a final sum over TWO transformed
    limbs, a single-output inner product.  Fused engine == unfused engine == oracle on every row outside the window."""
    q = oracle.nth_prime(LOGN, 0)
    q1 = oracle.nth_prime(LOGN, 1)
    nslots = 24
    rng = np.random.default_rng(77)
    mem = np.stack([rng.integers(0, q, N, dtype=np.uint64) for _ in range(nslots)])
    setq = lambda m: [f"vsetq {m:#x}"]
    progs = {
        # 0: the regular shape (two sums -> two in-place transforms -> an inner-product pair), then the same limbs loaded again
        "regular_then_reload": setq(q) + _sum(q, [0, 1], [3, 5], 8) + _sum(q, [2, 3], [7, 11], 9, const=13) + _ntt(8) + _ntt(9) +
                               _dot([8, 9, 4], [5, 6, 7], 16) + _dot([8, 9, 4], [6, 7, 5], 17) + _ntt(8, 18),
        # 1: a sum's operand (slot 1) is the in-place destination of another transform of the batch
        "operand_overwritten": setq(q) + _sum(q, [0, 1], [3, 5], 8) + _ntt(8) + _ntt(1) + _dot([8, 1], [5, 6], 16) + _dot([8, 1], [6, 7], 17),
        # 2: the transform of a sum goes elsewhere: the sum's store is live
        "out_of_place": setq(q) + _sum(q, [0, 1], [3, 5], 8) + _ntt(8, 10) + _dot([10, 4], [5, 6], 16) + _dot([10, 4], [6, 7], 17),
        # 3: one transformed limb in two different inner products, and an inner product with a single output
        "shared_digit": setq(q) + _ntt(8) + _ntt(9) + _dot([8, 9], [5, 6], 16) + _dot([8, 9], [6, 7], 17) + _dot([8, 4], [5, 6], 18) + _dot([9], [7], 19),
        # 4: the mod-down tail, once regular (t * c0 + acc * c1 + base), once over two transformed limbs
        "epilogues": setq(q) + _sum(q, [0, 1, 2], [3, 5, 7], 8, const=9) + _ntt(8) + _ntt(9) + _sum(q, [8, 4, 5], [21, 22, 1], 16) +
                     _sum(q, [8, 9], [2, 3], 17),
        # 5: two moduli in one chain
        "two_moduli": setq(q) + _sum(q, [0, 1], [3, 5], 8) + setq(q1) + _sum(q1, [0, 1], [3, 5], 9) + setq(q) + _ntt(8) + setq(q1) + _ntt(9) +
                      setq(q) + _dot([8, 4], [5, 6], 16) + _dot([8, 4], [6, 7], 17) + setq(q1) + _dot([9, 4], [5, 6], 18) + _dot([9, 4], [6, 7], 19),
    }
    progs.update({
        # 6: the tensor product of two ciphertexts (products held, the cross sum matched to them), operands in every order
        "tensor": setq(q) + _ew(0, 1, 8) + _ew(2, 3, 9) + _ew(4, 5, 12) + _ew(7, 6, 13) + _ew(0, 5, 20) +
                  _dot([0, 2], [3, 1], 16) + _dot([5, 6], [7, 4], 17) + _dot([0, 1], [2, 3], 18),
        # 7: a product in place on an operand of the sum, a sum that overwrites a product's operand: no fusion, same results
        "tensor_hazards": setq(q) + _ew(0, 1, 0) + _ew(2, 3, 9) + _dot([0, 2], [3, 1], 16) + _ew(4, 5, 12) + _ew(6, 7, 13) + _dot([4, 6], [7, 5], 5),
    })
    for name, lines in progs.items():
        words = _program(lines)
        res = {}
        for kind in ("oracle", "fused", "window", "plain"):
            vp = oracle.OracleVP(spm_rows=nslots * ROWS, ksk_rows=1) if kind == "oracle" else \
                avp.VP(spm_rows=nslots * ROWS, ksk_rows=1, max_log_n=16, flags=avp.FLAG_NO_FUSION if kind == "plain" else 0)
            vp.write(SPM, 0, mem.reshape(-1))
            vp.load_isram(0, words)
            if kind == "window":
                vp.set_scratch(8 * ROWS, 4 * ROWS)                 # slots 8..11
            assert vp.run(0) == 0, (name, kind, getattr(vp, "last_error", ""))
            vp.sync()
            res[kind] = vp.read(SPM, 0, nslots * N).reshape(nslots, N)
            if kind != "oracle":
                vp.close()
        assert np.array_equal(res["plain"], res["oracle"]), name
        assert np.array_equal(res["fused"], res["oracle"]), name
        keep = [i for i in range(nslots) if not 8 <= i < 12]
        assert np.array_equal(res["window"][keep], res["oracle"][keep]), name
        if name == "regular_then_reload":                         # slot 8 is loaded again after the inner products: it must have been stored
            assert np.array_equal(res["window"][8], res["oracle"][8])


@pytest.mark.parametrize("rows", [1, 2, 4])
def test_every_shape_of_the_inner_product_kernel(rows):
    """csrc/ntt_ks.hip picks the rows per wave of an inner-product launch from its groups (rowdot_rows); ALOHA_ROWDOT_ROWS forces one shape
    for the whole process (read once), so each shape runs the 8- and the 24-group case above in a process of its own: tiles of four digits,
    tiles shared by left-over transformed and plain digits (1, 2 or 4 of them per tile), idle lines -- all equal to the oracle machine."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALOHA_ROWDOT_ROWS=str(rows))
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_ks_fused.py"),
                        "-k", "test_fused_phases_equal_oracle_and_unfused and (10-2-2-2 or 7-1-1-1)"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-1000:]
