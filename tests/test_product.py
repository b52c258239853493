"""The polynomial product at N = 2^14 as ONE launch (csrc/ntt_product.hip; BASELINE config 3): the sequencer recognises
`vle; vntt; vle; vntt; vfqmul.vv; vintt; vse` in a program and hands every such limb to one kernel.  GPU == the oracle's
ISA machine running the same program, word for word; the CPU part checks the kernel's LDS layouts on the host."""
import numpy as np
import pytest

import oracle
from aloha_amd import isa, programs
from aloha_amd import vp as avp

SPM, KSK = 0, 1
LOGN, N = 14, 1 << 14
ROWS = N // 128


# ---- CPU: the three register layouts of a 512-point half row under the LDS swizzle (ntt_product.hip) -----------------------
def _swh(y):
    return y ^ ((y >> 3) & 31)


_LAYOUTS = {
    "1": lambda l, r: (r << 6) | l,
    "2": lambda l, r: ((l >> 3) << 6) | (r << 3) | (l & 7),
    "3": lambda l, r: (l << 3) | r,
}


@pytest.mark.parametrize("name", sorted(_LAYOUTS))
def test_half_row_layouts_are_bijections_without_bank_conflicts(name):
    lay = _LAYOUTS[name]
    assert sorted(lay(l, r) for l in range(64) for r in range(8)) == list(range(512))
    assert sorted(_swh(y) for y in range(512)) == list(range(512))
    for r in range(8):
        for half in range(2):                                    # a 64-bit LDS access serves 32 lanes at a time: 32 distinct 8-byte banks
            banks = {_swh(lay(l, r)) % 32 for l in range(32 * half, 32 * half + 32)}
            assert len(banks) == 32, (name, r, half)


def test_half_row_twiddle_indices():
    """Stage s of the transform takes twiddle entry 2^s + (e >> (14 - s)); the kernel's per-layout index arithmetic must say the
    same for every element a lane holds (R = half row, e = R * 512 + y; stage 4 pairs the halves of row rho = R >> 1)."""
    for R in (0, 11, 31):
        assert (1 << 4) + ((R * 512) >> 10) == 16 + (R >> 1)
        for l in range(64):
            for r in range(8):
                for ss in range(3):
                    g = r >> (3 - ss)                            # group of sub-stage ss: the top ss bits of the register index
                    e = R * 512 + _LAYOUTS["1"](l, r)            # stages 5..7: wave-uniform
                    assert (1 << (5 + ss)) + (e >> (9 - ss)) == (32 << ss) + (R << ss) + g
                    e = R * 512 + _LAYOUTS["2"](l, r)            # stages 8..10: per lane (y bits 8..6 = l >> 3)
                    assert (1 << (8 + ss)) + (e >> (6 - ss)) == (256 << ss) + ((R * 8 + (l >> 3)) << ss) + g
                    e = R * 512 + _LAYOUTS["3"](l, r)            # stages 11..13: per lane (y >> 3 = l)
                    assert (1 << (11 + ss)) + (e >> (3 - ss)) == (2048 << ss) + ((R * 64 + l) << ss) + g


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
def residues(rng, n, q):
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    a[:4] = [0, 1, q - 1, q - 2]
    a[-2:] = [q - 1, 0]
    return a


def run_both(words, inputs, spm_rows, read, launches=None, flags=0, repeat=1, **csr):
    """inputs: {row: array}; read: (row, words).  Returns the engine's words after checking them against the oracle machine
    (every repetition: the first run is sequenced, later ones replay its recorded launches)."""
    o = oracle.OracleVP(spm_rows=spm_rows, ksk_rows=1)
    for row, data in inputs.items():
        o.write(SPM, row, data)
    o.load_isram(0, words)
    assert o.run(0, **csr) == 0
    want = o.read(SPM, read[0], read[1])
    with avp.VP(spm_rows=spm_rows, ksk_rows=1, max_log_n=LOGN, flags=flags) as vp:
        vp.load_isram(0, words)
        for rep in range(repeat):
            for row, data in inputs.items():                     # (in-place programs: every repetition starts from the inputs)
                vp.write(SPM, row, data)
            k0 = vp.stats()["kernel_launches"]
            assert vp.run(0, **csr) == 0
            vp.sync()
            if launches is not None:
                assert vp.stats()["kernel_launches"] - k0 == launches, rep
            got = vp.read(SPM, read[0], read[1])
            assert np.array_equal(got, want), rep
    return got


@pytest.mark.gpu
def test_product_program_is_one_launch_and_equals_the_oracle_machine():
    L = 8
    mods = [oracle.nth_prime(LOGN, i) for i in range(L)]
    rng = np.random.default_rng(314)
    a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
    words = programs.assemble(programs.poly_mul(N, mods))
    got = run_both(words, {0: a, L * ROWS: b}, 3 * L * ROWS, (2 * L * ROWS, L * N), launches=1, repeat=3,
                   src0=0, src1=L * ROWS, rslt=2 * L * ROWS)
    # and the answer is the negacyclic product: one coefficient by direct convolution
    q, j = mods[0], 4321
    acc = 0
    for i in range(N):
        acc += int(a[i]) * int(b[j - i]) if i <= j else -int(a[i]) * int(b[j - i + N])
    assert int(got[j]) == acc % q


@pytest.mark.gpu
def test_product_in_place_and_squaring():
    L = 3
    mods = [oracle.nth_prime(LOGN, 5 + i) for i in range(L)]
    rng = np.random.default_rng(315)
    a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
    words = programs.assemble(programs.poly_mul(N, mods))
    # result over operand a, over operand b: NOT matched since round 4 (a one-launch product that gave up on a wait would have
    # destroyed its own input) -- the program runs loop by loop, same words; a * a elsewhere: one launch
    run_both(words, {0: a, L * ROWS: b}, 2 * L * ROWS, (0, 2 * L * N), launches=None, repeat=2, src0=0, src1=L * ROWS, rslt=0)
    run_both(words, {0: a, L * ROWS: b}, 2 * L * ROWS, (0, 2 * L * N), launches=None, src0=0, src1=L * ROWS, rslt=L * ROWS)
    run_both(words, {0: a}, 2 * L * ROWS, (0, 2 * L * N), launches=1, src0=0, src1=0, rslt=L * ROWS)
    with avp.VP(spm_rows=2 * L * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:   # ... and the in-place form really is more than one launch
        vp.write(SPM, 0, a); vp.write(SPM, L * ROWS, b); vp.load_isram(0, words)
        k0 = vp.stats()["kernel_launches"]
        assert vp.run(0, src0=0, src1=L * ROWS, rslt=0) == 0
        vp.sync()
        assert vp.stats()["kernel_launches"] - k0 > 1


@pytest.mark.gpu
def test_product_five_launch_form_is_word_identical(monkeypatch):
    """The form a handle falls back to when another handle's one-launch product is in flight on the device (or its own status word
    was once set): forward transforms of both operands as one batch, the products, the inverse transforms -- ALOHA_PRODUCT_FALLBACK
    forces it.  Same words as the one-launch form and the oracle machine; the counter says which form ran."""
    L = 8
    mods = [oracle.nth_prime(LOGN, i % 3) for i in range(L)]
    rng = np.random.default_rng(3141)
    a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
    words = programs.assemble(programs.poly_mul(N, mods))
    one = run_both(words, {0: a, L * ROWS: b}, 3 * L * ROWS, (2 * L * ROWS, L * N), launches=1, src0=0, src1=L * ROWS, rslt=2 * L * ROWS)
    monkeypatch.setenv("ALOHA_PRODUCT_FALLBACK", "1")
    five = run_both(words, {0: a, L * ROWS: b}, 3 * L * ROWS, (2 * L * ROWS, L * N), launches=5, repeat=3, src0=0, src1=L * ROWS, rslt=2 * L * ROWS)
    assert np.array_equal(one, five)
    with avp.VP(spm_rows=3 * L * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
        vp.write(SPM, 0, a); vp.write(SPM, L * ROWS, b); vp.load_isram(0, words)
        assert vp.run(0, src0=0, src1=L * ROWS, rslt=2 * L * ROWS) == 0
        vp.sync()
        assert vp.stats()["product_fallbacks"] == 1


@pytest.mark.gpu
def test_two_handles_running_64_limb_products_from_two_threads():
    """two handles on two streams, each running 64-limb products (a grid of 2560 workgroups whose later phases
    wait for the earlier ones) from its own thread, at the same time.  The admission rule lets ONE handle per device have such a
    launch in flight; the other runs the five-launch form meanwhile.  Every result equals the oracle, no status word is set."""
    import threading
    Lb = 64
    mods = [oracle.nth_prime(LOGN, i % 8) for i in range(Lb)]
    words = programs.assemble(programs.poly_mul(N, mods))
    rng = np.random.default_rng(3142)
    data, want = [], []
    for h in range(2):
        a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
        data.append((a, b))
        exp = []
        for i in (0, 17, 63):                                     # three limbs per handle through the oracle (a full 64-limb product is ~0.3 s each)
            fa, fb = oracle.ntt(a[i * N:(i + 1) * N], mods[i]), oracle.ntt(b[i * N:(i + 1) * N], mods[i])
            prod = np.array([(int(x) * int(y)) % mods[i] for x, y in zip(fa, fb)], dtype=np.uint64)
            exp.append((i, oracle.intt(prod, mods[i])))
        want.append(exp)
    vps = [avp.VP(spm_rows=3 * Lb * ROWS, ksk_rows=1, max_log_n=LOGN) for _ in range(2)]
    errors, firsts = [], [None, None]
    try:
        for vp, (a, b) in zip(vps, data):
            vp.write(SPM, 0, a); vp.write(SPM, Lb * ROWS, b); vp.load_isram(0, words)
        barrier = threading.Barrier(2)

        def work(h):
            try:
                vp = vps[h]
                barrier.wait()
                for rep in range(40):
                    assert vp.run(0, src0=0, src1=Lb * ROWS, rslt=2 * Lb * ROWS) == 0
                    if rep % 8 == 7:
                        vp.sync()                                 # (raises if a status word was set)
                        got = vp.read(SPM, 2 * Lb * ROWS, Lb * N)
                        if firsts[h] is None:
                            firsts[h] = got
                        assert np.array_equal(got, firsts[h]), f"handle {h}: result changed between repetitions"
                vp.sync()
            except Exception as e:                                # noqa: BLE001
                errors.append((h, e))

        ts = [threading.Thread(target=work, args=(h,)) for h in range(2)]
        for t in ts: t.start()
        for t in ts: t.join()
        assert not errors, errors
        for h in range(2):
            for i, exp in want[h]:
                assert np.array_equal(firsts[h][i * N:(i + 1) * N], exp), (h, i)
        fallbacks = [vp.stats()["product_fallbacks"] for vp in vps]
        launches = [vp.stats()["kernel_launches"] for vp in vps]
        print("product_fallbacks per handle:", fallbacks, "kernel launches:", launches)
        assert sum(fallbacks) > 0                                 # the two did meet: somebody took the five-launch form
    finally:
        for vp in vps:
            vp.close()


@pytest.mark.gpu
def test_product_batches_beyond_one_launch_and_mixed_moduli():
    """70 limbs: two launches (64 + 6), the second on the other counter bank; limbs of one prime and of different primes mixed."""
    L = 70
    mods = [oracle.nth_prime(LOGN, (i * 7) % 5) for i in range(L)]
    rng = np.random.default_rng(316)
    a = np.concatenate([rng.integers(0, q, N, dtype=np.uint64) for q in mods])
    b = np.concatenate([rng.integers(0, q, N, dtype=np.uint64) for q in mods])
    words = programs.assemble(programs.poly_mul(N, mods))
    with avp.VP(spm_rows=3 * L * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
        vp.write(SPM, 0, a); vp.write(SPM, L * ROWS, b)
        vp.load_isram(0, words)
        for rep in range(3):                                     # first run sequenced, then replays: the launch count alternates the banks
            k0 = vp.stats()["kernel_launches"]
            assert vp.run(0, src0=0, src1=L * ROWS, rslt=2 * L * ROWS) == 0
            vp.sync()
            assert vp.stats()["kernel_launches"] - k0 == 2
            got = vp.read(SPM, 2 * L * ROWS, L * N).reshape(L, N)
            for j in (0, 1, 63, 64, 69):
                q = mods[j]
                fa, fb = oracle.ntt(a[j * N:(j + 1) * N], q), oracle.ntt(b[j * N:(j + 1) * N], q)
                prod = np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64)
                assert np.array_equal(got[j], oracle.intt(prod, q)), (rep, j)
        # against the unfused execution of the same program, every limb
        with avp.VP(spm_rows=3 * L * ROWS, ksk_rows=1, max_log_n=LOGN, flags=avp.FLAG_NO_FUSION) as ref:
            ref.write(SPM, 0, a); ref.write(SPM, L * ROWS, b)
            ref.load_isram(0, words)
            assert ref.run(0, src0=0, src1=L * ROWS, rslt=2 * L * ROWS) == 0
            ref.sync()
            assert np.array_equal(ref.read(SPM, 2 * L * ROWS, L * N).reshape(L, N), got)


@pytest.mark.gpu
def test_product_matcher_refusals_still_compute_the_program():
    """Windows the matcher must NOT turn into a product job: a temporary read afterwards, the product stored as well, a
    destination that straddles an operand, another ring size.  The programs then run as before (more launches), and equal the
    oracle machine either way."""
    q = oracle.nth_prime(LOGN, 2)
    rng = np.random.default_rng(317)
    a, b = residues(rng, N, q), residues(rng, N, q)
    head = [f"vsetvl {64 * N:#x}", f"vsetq {q:#x}"]
    base = ["vle v0, src0+0x0", "vntt v2, v0", "vle v4, src1+0x0", "vntt v1, v4", "vfqmul.vv v0, v2, v1", "vintt v2, v0"]

    def check(lines, read_words, fused, **csr):
        words = isa.assemble("\n".join(head + lines + ["break"]))
        o = oracle.OracleVP(spm_rows=8 * ROWS, ksk_rows=1)
        with avp.VP(spm_rows=8 * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
            out = []
            for e in (vp, o):
                e.write(SPM, 0, a); e.write(SPM, ROWS, b)
                e.load_isram(0, words)
                k0 = e.stats()["kernel_launches"] if e is vp else 0
                assert e.run(0, **csr) == 0
                e.sync()
                if e is vp:
                    assert (vp.stats()["kernel_launches"] - k0 == 1) == fused
                out.append(e.read(SPM, 0, read_words))
            assert np.array_equal(out[0], out[1])

    csr = dict(src0=0, src1=ROWS, rslt=2 * ROWS)
    check(base + [f"vse v2, rslt+0x0"], 4 * N, True, **csr)                                              # the pattern itself
    check(base + [f"vse v2, rslt+0x0", f"vse v1, rslt+{8 * N:#x}"], 4 * N, False, **csr)                 # NTT(b) is read afterwards
    check(base[:5] + [f"vse v0, rslt+{8 * N:#x}", "vintt v2, v0", "vse v2, rslt+0x0"], 4 * N, False, **csr)   # the product is stored too
    check(base + [f"vse v2, src0+{4 * N:#x}"], 4 * N, False, **csr)                                      # destination straddles a and b
    check(base + [f"vse v2, rslt+0x0", "vle v6, rslt+0x0", "vfqadd.vv v3, v6, v1", f"vse v3, rslt+{8 * N:#x}"], 4 * N, False, **csr)


@pytest.mark.gpu
def test_product_followed_and_preceded_by_other_work():
    """The product job joins the ordinary flow of a program: operands produced by earlier instructions (flushed first),
    results consumed by later ones."""
    q = oracle.nth_prime(LOGN, 1)
    rng = np.random.default_rng(318)
    a, b = residues(rng, N, q), residues(rng, N, q)
    lines = [f"vsetvl {64 * N:#x}", f"vsetq {q:#x}",
             "vle v0, src0+0x0", "vfqadd.vs v2, v0, 0x5", f"vse v2, rslt+{8 * N:#x}",                   # a + 5 -> rslt limb 1
             f"vle v0, rslt+{8 * N:#x}", "vntt v2, v0", "vle v4, src1+0x0", "vntt v1, v4", "vfqmul.vv v0, v2, v1", "vintt v2, v0",
             "vse v2, rslt+0x0",                                                                         # (a + 5) * b -> rslt limb 0
             "vle v0, rslt+0x0", "vfqmul.vs v2, v0, 0x3", f"vse v2, rslt+{16 * N:#x}", "break"]          # 3 * that -> rslt limb 2
    words = isa.assemble("\n".join(lines))
    run_both(words, {0: a, ROWS: b}, 6 * ROWS, (2 * ROWS, 3 * N), launches=3, repeat=2, src0=0, src1=ROWS, rslt=2 * ROWS)


@pytest.mark.gpu
def test_product_tags_survive_launches_of_different_sizes():
    """The hand-off words carry a 2-bit tag per launch and a limb slot keeps the tag of the last launch that used it: launches
    of different sizes can leave all four tags behind, so that a product finds no free tag and the engine wipes its
    slots first; every result along the way (and after) equals the oracle machine.  (Sizes 1, 3, 1, 2, 1 leave tags 2, 3, 1, 0 in
    slots 0..3 under the engine's lowest-free-tag rule: the 4-limb product that follows is the shortest sequence that wipes.)"""
    rng = np.random.default_rng(319)
    mods = [oracle.nth_prime(LOGN, i) for i in range(4)]
    a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
    with avp.VP(spm_rows=12 * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
        o = oracle.OracleVP(spm_rows=12 * ROWS, ksk_rows=1)
        for e in (vp, o):
            e.write(SPM, 0, a); e.write(SPM, 4 * ROWS, b)
        pc = 0
        for L in (1, 3, 1, 2, 1, 4, 4, 2, 3, 1, 4):
            words = programs.assemble(programs.poly_mul(N, mods[:L]))
            for e in (vp, o):
                e.load_isram(pc, words)
                assert e.run(pc, src0=0, src1=4 * ROWS, rslt=8 * ROWS) == 0
            vp.sync()
            assert np.array_equal(vp.read(SPM, 8 * ROWS, L * N), o.read(SPM, 8 * ROWS, L * N)), L
            pc += 64                                             # (distinct programs: no recorded run is replayed)


def _random_product_program(rng, primes, nslots, segments):
    """Random straight-line program at N = 2^14 whose segments are product idioms -- random registers, load order, operand and
    destination slots (in place, squaring, shared operands), sometimes with a temporary read afterwards so that the matcher must
    refuse -- and plain element-wise work between them (operands produced and consumed around the products)."""
    lines = [f"vsetvl {64 * N:#x}", f"vsetq {primes[0]:#x}"]
    defined = set()
    slot = lambda: f"src0+{int(rng.integers(nslots)) * 8 * N:#x}"
    for _ in range(segments):
        kind = rng.choice(["prod", "prod", "prod", "ew", "setq", "intt"])
        if kind == "setq":
            lines.append(f"vsetq {int(rng.choice(primes)):#x}")
        elif kind == "ew":
            a, d = int(rng.integers(10)), int(rng.integers(10))
            lines += [f"vle v{a}, {slot()}", f"{rng.choice(['vfqadd.vs', 'vfqmul.vs'])} v{d}, v{a}, {int(rng.integers(1, 1 << 59)):#x}"]
            defined.update({a, d})
            if rng.integers(2):
                lines.append(f"vse v{d}, {slot()}")
        elif kind == "intt":                                  # a transform job of the ordinary kind next to the products
            lines += [f"vle v0, {slot()}", f"{rng.choice(['vntt', 'vintt'])} v2, v0", f"vse v2, {slot()}"]
            defined.discard(0); defined.add(2)
        else:
            t1 = int(rng.integers(10))
            t2 = int(rng.choice([r for r in range(10) if r % 2 != t1 % 2]))
            va = int(rng.choice([r for r in range(10) if r not in (t1, t2)]))
            vb = int(rng.choice([r for r in range(10) if r not in (t1, t2, va)]))
            t3 = int(rng.integers(10))
            t4 = int(rng.choice([r for r in range(10) if r != t3]))
            sa, sb = slot(), slot()
            if rng.integers(4) == 0:
                sb = sa                                        # a * a
            sd = rng.choice([slot(), sa, sb])                  # anywhere, or over an operand
            if rng.integers(2):
                seq = [f"vle v{va}, {sa}", f"vntt v{t1}, v{va}", f"vle v{vb}, {sb}", f"vntt v{t2}, v{vb}"]
            else:
                seq = [f"vle v{va}, {sa}", f"vle v{vb}, {sb}", f"vntt v{t1}, v{va}", f"vntt v{t2}, v{vb}"]
            seq += [f"vfqmul.vv v{t3}, v{t1}, v{t2}", f"vintt v{t4}, v{t3}", f"vse v{t4}, {sd}"]
            lines += seq
            for ln in seq:                                     # (vs1 of a transform is undefined afterwards)
                op, rest = ln.split(" ", 1)
                regs = [int(x[1:]) for x in rest.replace(",", " ").split() if x[0] == "v" and x[1:].isdigit()]
                if op in ("vntt", "vintt"):
                    defined.discard(regs[1]); defined.add(regs[0])
                elif op != "vse":
                    defined.add(regs[0])
            live = [r for r in (t1, t2, t4) if r in defined]
            if live and rng.integers(4) == 0:                  # a temporary is read afterwards: not a product job
                lines.append(f"vse v{int(rng.choice(live))}, {slot()}")
    for r in sorted(defined):
        lines.append(f"vse v{r}, src0+{(nslots + r) * 8 * N:#x}")
    return "\n".join(lines + ["break"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_programs_with_products_fused_unfused_oracle(seed):
    """Differential fuzz of the product matcher: fused engine == unfused engine == oracle machine on ALL memory, then the recorded
    run replayed twice on other data."""
    rng = np.random.default_rng(4000 + seed)
    nslots = 6
    primes = [oracle.nth_prime(LOGN, i) for i in range(3)]
    words = isa.assemble(_random_product_program(rng, primes, nslots, 12))
    assert len(words) < 1024
    total = (nslots + 10) * N
    o = oracle.OracleVP(spm_rows=(nslots + 10) * ROWS, ksk_rows=1)
    with avp.VP(spm_rows=(nslots + 10) * ROWS, ksk_rows=1, max_log_n=LOGN) as vp, \
            avp.VP(spm_rows=(nslots + 10) * ROWS, ksk_rows=1, max_log_n=LOGN, flags=avp.FLAG_NO_FUSION) as plain:
        for rep in range(3):
            mem = np.concatenate([residues(rng, N, primes[0]) for _ in range(nslots)])
            res = []
            for e in ((vp, plain, o) if rep == 0 else (vp, o)):
                e.write(SPM, 0, np.zeros(total, np.uint64)); e.write(SPM, 0, mem)
                if rep == 0:
                    e.load_isram(0, words)
                assert e.run(0, src0=0) == 0
                if e is not o:
                    e.sync()
                res.append(e.read(SPM, 0, total))
            for r in res[:-1]:
                assert np.array_equal(r, res[-1]), (seed, rep)


@pytest.mark.gpu
def test_products_of_two_engines_interleaved():
    """Two handles (two streams, two scratch buffers) launch their products alternately without waiting for each other: the
    waits inside one launch must not depend on what else runs on the chip."""
    L = 8
    mods = [oracle.nth_prime(LOGN, i) for i in range(L)]
    rng = np.random.default_rng(320)
    words = programs.assemble(programs.poly_mul(N, mods))
    data = []
    for _ in range(2):
        a = np.concatenate([residues(rng, N, q) for q in mods]); b = np.concatenate([residues(rng, N, q) for q in mods])
        o = oracle.OracleVP(spm_rows=3 * L * ROWS, ksk_rows=1)
        o.write(SPM, 0, a); o.write(SPM, L * ROWS, b); o.load_isram(0, words)
        assert o.run(0, src0=0, src1=L * ROWS, rslt=2 * L * ROWS) == 0
        data.append((a, b, o.read(SPM, 2 * L * ROWS, L * N)))
    with avp.VP(spm_rows=3 * L * ROWS, ksk_rows=1, max_log_n=LOGN) as v0, avp.VP(spm_rows=3 * L * ROWS, ksk_rows=1, max_log_n=LOGN) as v1:
        for vp, (a, b, _) in zip((v0, v1), data):
            vp.write(SPM, 0, a); vp.write(SPM, L * ROWS, b); vp.load_isram(0, words)
        for _ in range(50):
            for vp in (v0, v1):
                assert vp.run(0, src0=0, src1=L * ROWS, rslt=2 * L * ROWS) == 0
        for vp, (_, _, want) in zip((v0, v1), data):
            vp.sync()
            assert np.array_equal(vp.read(SPM, 2 * L * ROWS, L * N), want)
