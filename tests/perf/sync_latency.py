"""One small call and its synchronisation, as a caller of the C ABI sees it (BASELINE config 2: one limb at N = 2^12; the reference's
mul_plain at N = 2^13): microseconds per `run` + `sync`, median of 300.  (round 4: a polling wait in front of hipStreamSynchronize was measured with this and dropped: 18.2 against 16.9 us):

    python tests/perf/sync_latency.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aloha_amd import isa  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
res = {}
for logn in (12, 13, 16):
    n = 1 << logn
    q = avp.nth_prime(logn, 0)
    with avp.VP(spm_rows=4 * n // 128, ksk_rows=1, max_log_n=logn) as vp:
        vp.write(SPM, 0, np.random.default_rng(logn).integers(0, q, n, dtype=np.uint64))
        vp.load_isram(0, isa.assemble("\n".join([f"vsetvl {64 * n:#x}", f"vsetq {q:#x}", "vle v0, src0+0x0", "vntt v2, v0", "vse v2, rslt+0x0", "break"])))
        ts = []
        for i in range(330):
            t0 = time.perf_counter()
            vp.run(0, src0=0, rslt=n // 128)
            vp.sync()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[30:])
        res[f"vntt_n{n}"] = {"median_us": round(ts[len(ts) // 2] * 1e6, 2), "p10_us": round(ts[len(ts) // 10] * 1e6, 2), "p90_us": round(ts[9 * len(ts) // 10] * 1e6, 2)}
print(json.dumps(res))
