#!/usr/bin/env python3
"""Soak of the two differential fuzz tests on the GPU box, seeds beyond the ones the suite runs:

    python tools/soak_fuzz.py <first seed> <count> [product|sequencer|both]

tests/test_product.py::test_random_programs_with_products_fused_unfused_oracle (the product matcher at N = 2^14) and
tests/test_gpu_parity.py::test_random_programs_fused_unfused_oracle (the sequencer at N <= 2^13): fused engine == unfused
engine == oracle machine on all memory, recorded runs replayed.  Prints one line per 10 seeds and the failures."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tg  # noqa: E402
import test_product as tp  # noqa: E402
from aloha_amd import vp as avp  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "both"
bad = []
t0 = time.time()
engine = avp.VP(spm_rows=1 << 16, ksk_rows=1 << 16, max_log_n=16)
for i, seed in enumerate(range(first, first + count)):
    for name, fn in (("product", lambda s: tp.test_random_programs_with_products_fused_unfused_oracle(s)),
                     ("sequencer", lambda s: tg.test_random_programs_fused_unfused_oracle.__wrapped__(engine, s)
                      if hasattr(tg.test_random_programs_fused_unfused_oracle, "__wrapped__") else tg.test_random_programs_fused_unfused_oracle(engine, s))):
        if which not in ("both", name):
            continue
        try:
            fn(seed)
        except Exception as e:                                   # noqa: BLE001
            bad.append((name, seed, repr(e)[:200]))
    if i % 10 == 9:
        print(f"{i + 1} seeds, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
st = engine.stats()
print("shared engine:", {k: st[k] for k in ("flow_runs", "kernel_launches", "product_fallbacks") if k in st})
engine.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
