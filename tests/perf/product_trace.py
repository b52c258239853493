"""Where the time of ONE product launch goes (csrc/ntt_product.hip): a variant build with -DALOHA_PROD_TRACE leaves 100 MHz
clock stamps of every workgroup's steps; this prints, per phase, when its workgroups started and passed each step, relative
to the first stamp of the launch.  Run on the GPU box:

    python -m aloha_amd.build --variant=trace:ALOHA_PROD_TRACE=1
    ALOHA_VP_LIB=aloha_amd/lib/libaloha_vp_trace.so python tests/perf/product_trace.py [limbs]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402  (checks the product before its timing is looked at)
from aloha_amd import programs  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
logn, n = 14, 1 << 14
rows = n // 128
mods = [avp.nth_prime(logn, i % 8) for i in range(L)]
with avp.VP(spm_rows=3 * L * rows, ksk_rows=1, max_log_n=14) as vp:
    vp.fill_synthetic(SPM, 0, logn, mods + mods, seed_base=0xA10A + 300000)
    vp.load_isram(0, programs.assemble(programs.poly_mul(n, mods)))
    for _ in range(20):
        assert vp.run(0, src0=0, src1=L * rows, rslt=2 * L * rows) == 0
    vp.sync()
    x, y = vp.read(SPM, 0, n), vp.read(SPM, L * rows, n)
    fx, fy = oracle.ntt(x, mods[0]), oracle.ntt(y, mods[0])
    prod = np.array([(int(u) * int(v)) % mods[0] for u, v in zip(fx, fy)], dtype=np.uint64)
    assert np.array_equal(vp.read(SPM, 2 * L * rows, n), oracle.intt(prod, mods[0]))
    nP1, nP2, nP3 = L * 16, L * 16, L * 8                        # workgroups of 256 threads per phase (ntt_product.hip)
    nwg = nP1 + nP2 + nP3
    buf = np.zeros(nwg * 16, dtype=np.uint64)
    fn = vp._L.aloha_vp_debug_product_trace
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    assert fn(vp._h, buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(nwg, 16).astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0                                            # 100 MHz -> microseconds
P1 = {0: "start", 4: "operand loaded", 1: "radix-16 done", 2: "stores issued"}
P2 = {0: "start", 2: "row loaded (all tags seen)", 3: "stages 4-7 done", 4: "exchange 1->2 done", 5: "stages 8-10 done",
      6: "exchange 2->3 done", 7: "stages 11-13 done", 8: "canonical, b's half in LDS", 9: "barrier passed", 10: "product done",
      11: "inverse 13-11 done", 12: "exchange 3->2 done", 13: "inverse 10-8 done", 14: "exchange + inverse 7-5 done", 15: "halves met, inverse 4 done"}
P3 = {0: "start", 4: "columns loaded (all tags seen)", 2: "radix-16 done", 3: "stores issued"}
for name, a, steps in (("P1", us[:nP1], P1), ("P2 (operand a, lower half)", us[nP1:nP1 + nP2], P2), ("P3", us[nP1 + nP2:], P3)):
    print(name)
    prev = None
    for slot, what in steps.items():
        col = a[:, slot]
        med = float(np.median(col))
        print(f"   {what:28s} min {col.min():7.2f}  median {med:7.2f}  max {col.max():7.2f} us" + (f"   (+{med - prev:5.2f})" if prev is not None else ""))
        prev = med
