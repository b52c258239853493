"""Run-to-run spread of the one-launch product (csrc/ntt_product.hip): fresh engines, 8 and 64 limbs, time per launch over
back-to-back launches; every result checked.  python tests/perf/product_repeat.py [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402
from aloha_amd import programs  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

SPM = 0
logn, n = 14, 1 << 14
rows = n // 128
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for rnd in range(rounds):
    for L in (8, 64):
        mods = [avp.nth_prime(logn, i % 8) for i in range(L)]
        with avp.VP(spm_rows=3 * L * rows, ksk_rows=1, max_log_n=14) as vp:
            vp.fill_synthetic(SPM, 0, logn, mods + mods, seed_base=0xA10A + 300000 + rnd)
            vp.load_isram(0, programs.assemble(programs.poly_mul(n, mods)))
            run = lambda: vp.run(0, src0=0, src1=L * rows, rslt=2 * L * rows)
            assert run() == 0
            vp.sync()
            o = oracle.OracleVP(spm_rows=3 * L * rows, ksk_rows=1)              # every limb, through the oracle machine on the same program
            o.write(SPM, 0, vp.read(SPM, 0, 2 * L * n)); o.load_isram(0, programs.assemble(programs.poly_mul(n, mods)))
            assert o.run(0, src0=0, src1=L * rows, rslt=2 * L * rows) == 0
            assert np.array_equal(vp.read(SPM, 2 * L * rows, L * n), o.read(SPM, 2 * L * rows, L * n))
            o.close()
            ts = []
            for blk in range(5):
                t0 = time.perf_counter()
                for _ in range(40):
                    run()
                vp.sync()
                ts.append((time.perf_counter() - t0) / 40 * 1e6)
            print(f"round {rnd}  limbs {L:3d}  us per launch, 5 blocks of 40: " + " ".join(f"{t:8.1f}" for t in ts), flush=True)
