#!/usr/bin/env python3
"""How long the N = 2^16 transform takes on SMALL batches (B limbs, each under its own prime): the shapes the key-switch launches
between its big kernels (24 limbs of d2, 8 special-prime accumulators) and every launch of a rank's share at W = 8.  A batch of 4096
limbs runs at ~0.39 us per limb; the small ones are bound by the latency of one tile, which this probe measures per B:

    python tests/perf/small_batch_latency.py > gpurun_out/small_batch.json

The transform is run R times back to back on one stream (a recorded launch list replayed), so the figure is device time per
transform = both passes, without the host."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from aloha_amd import isa  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
LOGN, N = 16, 1 << 16
ROWS = N // 128


def main() -> int:
    batches = [int(x) for x in os.environ.get("BATCHES", "1,2,4,8,16,24,32,48,64,96,128,256").split(",")]
    reps = 200
    res = {}
    bmax = max(batches)
    with avp.VP(spm_rows=2 * bmax * ROWS, ksk_rows=1, max_log_n=LOGN) as vp:
        rng = np.random.default_rng(7)
        primes = [avp.nth_prime(LOGN, i) for i in range(bmax)]
        for i in range(bmax):
            vp.write(SPM, i * ROWS, rng.integers(0, primes[i], N, dtype=np.uint64))
        for op in ("vntt", "vintt"):
            for B in batches:
                lines = [f"vsetvl {64 * N:#x}"]
                for i in range(B):
                    lines += [f"vsetq {primes[i]:#x}", f"vle v0, src0+{i * 8 * N:#x}", f"{op} v2, v0", f"vse v2, rslt+{i * 8 * N:#x}"]
                vp.load_isram(0, isa.assemble("\n".join(lines + ["break"])))
                run = lambda: vp.run(0, src0=0, rslt=bmax * ROWS)
                assert run() == 0
                vp.sync()
                s0 = vp.stats()
                for _ in range(20):
                    run()
                vp.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    run()
                vp.sync()
                t = (time.perf_counter() - t0) / reps
                s1 = vp.stats()
                res[f"{op}_B{B}"] = dict(us=round(t * 1e6, 2), us_per_limb=round(t * 1e6 / B, 3),
                                         launches=(s1["kernel_launches"] - s0["kernel_launches"]) / (reps + 20))
                print(op, B, res[f"{op}_B{B}"], file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
