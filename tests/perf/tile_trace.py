"""Where the time of ONE small N = 2^16 transform goes (csrc/ntt_tiles.hpp: TILE_STAMP / ROW_STAMP): a variant build with
-DALOHA_TILE_TRACE leaves 100 MHz clock stamps of the first thread of every workgroup -- column kernels in workgroup slots
[0, 2048) of the buffer, row kernels in [2048, 8192) -- and this prints, per kernel and per step, when the workgroups passed it,
relative to the first stamp of the transform.  Run on the GPU box:

    python -m aloha_amd.build --plain=tiletrace:ALOHA_TILE_TRACE=1
    ALOHA_VP_LIB=aloha_amd/lib/libaloha_vp_tiletrace.so python tests/perf/tile_trace.py [limbs] [vntt|vintt]

B limbs, each under its own prime: the shape of the key-switch's small transforms (24 limbs of d2, 8 special-prime accumulators)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aloha_amd import isa  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
op = sys.argv[2] if len(sys.argv) > 2 else "vntt"
logn, n = 16, 1 << 16
rows = n // 128
COL = {7: "kernel entry", 0: "job + prime read", 1: "data and early twiddles landed", 2: "first radix-16 done", 3: "exchange done",
       4: "second radix-16 done", 5: "stores issued", 6: "stores acknowledged"}
ROW = {7: "kernel entry", 0: "tile + job + prime read", 1: "data landed", 4: "both steps + transposes done", 5: "stores issued", 6: "stores acknowledged"}


def report(name, t, steps, t0):
    us = (t - t0) / 100.0
    print(f"{name}: {len(t)} workgroups, entries {us[:, 7].min():.2f} .. {us[:, 7].max():.2f} us, last store acknowledged {us[:, 6].max():.2f} us")
    prev = None
    for slot, what in steps.items():
        col = us[:, slot]
        med = float(np.median(col))
        print(f"   {what:32s} min {col.min():7.2f}  median {med:7.2f}  max {col.max():7.2f} us" + (f"   (+{med - prev:5.2f})" if prev is not None else ""))
        prev = med
    dur = us[:, 6] - us[:, 7]
    print(f"   one workgroup, entry -> stores acknowledged: min {dur.min():.2f}  median {float(np.median(dur)):.2f}  max {dur.max():.2f} us")


with avp.VP(spm_rows=2 * B * rows, ksk_rows=1, max_log_n=16) as vp:
    rng = np.random.default_rng(3)
    primes = [avp.nth_prime(logn, i) for i in range(B)]
    for i in range(B):
        vp.write(SPM, i * rows, rng.integers(0, primes[i], n, dtype=np.uint64))
    lines = [f"vsetvl {64 * n:#x}"]
    for i in range(B):
        lines += [f"vsetq {primes[i]:#x}", f"vle v0, src0+{i * 8 * n:#x}", f"{op} v2, v0", f"vse v2, rslt+{i * 8 * n:#x}"]
    vp.load_isram(0, isa.assemble("\n".join(lines + ["break"])))
    for _ in range(20):
        assert vp.run(0, src0=0, rslt=B * rows) == 0
    vp.sync()
    ncol, nrow = min(16 * B, 2048), min(64 * B, 6144)
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    fn = vp._L.aloha_vp_debug_tile_trace
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    assert fn(vp._h, buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(-1, 16).astype(np.int64)
col, row = t[:ncol], t[2048:2048 + nrow]
t0 = min(col[:, 7].min(), row[:, 7].min())
print(f"{op}, {B} limbs of N = 2^16, one prime each")
for name, a, steps in sorted((("column kernel", col, COL), ("row kernel", row, ROW)), key=lambda x: x[1][:, 7].min()):
    report(name, a, steps, t0)
print(f"transform, first entry -> last store acknowledged: {(max(col[:, 6].max(), row[:, 6].max()) - t0) / 100.0:.2f} us")
