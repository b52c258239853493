import sys, time, json
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from aloha_amd import vp as avp, programs
import oracle
SPM=0
def run(flags):
    logn, n, L = 14, 1 << 14, 8
    rows = n // 128
    mods = [avp.nth_prime(logn, i) for i in range(L)]
    rng = np.random.default_rng(3)
    a = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods]); b = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in mods])
    with avp.VP(spm_rows=4 * L * rows, ksk_rows=1, max_log_n=14, flags=flags) as vp:
        vp.write(SPM, 0, a.reshape(-1)); vp.write(SPM, L * rows, b.reshape(-1))
        vp.load_isram(0, programs.assemble(programs.poly_mul_batched(n, mods, scratch_row=3 * L * rows)))
        f = lambda: vp.run(0, src0=0, src1=L * rows, rslt=2 * L * rows)
        assert f() == 0
        got = vp.read(SPM, 2 * L * rows, n)
        fa, fb = oracle.ntt(a[0], mods[0]), oracle.ntt(b[0], mods[0])
        prod = np.array([(int(x) * int(y)) % mods[0] for x, y in zip(fa, fb)], dtype=np.uint64)
        assert np.array_equal(got, oracle.intt(prod, mods[0]))
        l0 = vp.stats()["kernel_launches"]
        f(); vp.sync()
        t0 = time.perf_counter()
        for _ in range(200): f()
        vp.sync()
        t = (time.perf_counter() - t0) / 200
        return t * 1e6, (vp.stats()["kernel_launches"] - l0) / 201
print("two-pass (default):", run(0))
print("single-pass N=2^14:", run(avp.FLAG_NTT_SINGLE_14))

def lat(logn):
    n = 1 << logn
    q = avp.nth_prime(logn, 0)
    with avp.VP(spm_rows=64 * n // 128, ksk_rows=1, max_log_n=logn) as vp:
        vp.ntt_rows(SPM, 0, 0, logn, [q]); vp.sync()
        t0 = time.perf_counter()
        for _ in range(300):
            vp.ntt_rows(SPM, 0, 0, logn, [q]); vp.sync()
        one = (time.perf_counter() - t0) / 300
        t0 = time.perf_counter()
        for _ in range(300):
            vp.ntt_rows(SPM, 0, 0, logn, [q])
        vp.sync()
        thr = (time.perf_counter() - t0) / 300
    return one * 1e6, thr * 1e6
for logn in (12, 13, 14, 16):
    print("single-limb NTT N=2^%d: %.1f us per synchronous call, %.1f us per call back to back" % ((logn,) + lat(logn)))
