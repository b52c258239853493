#!/usr/bin/env python3
"""The reference's four ROM programs at N = 8192 on the golden vectors (bench.py's `reference_programs_n8192` leg by itself),
with the dataflow sequencer (default) and, ALOHA_NO_FLOW=1, the in-order one.  Prints one JSON line per mode."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    import numpy as np  # noqa: F401
    import bench

    def timeit(fn, sync, reps):
        fn(); sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(max(1, reps // 5)):
                fn()
            sync()
            ts.append((time.perf_counter() - t0) / max(1, reps // 5))
        return sorted(ts)[2]
    r = bench.reference_programs(0, timeit)
    print(json.dumps({"mode": "in-order (ALOHA_NO_FLOW=1)" if os.environ.get("ALOHA_NO_FLOW") else "dataflow (default)",
                      **{k: ({kk: round(vv, 2) if isinstance(vv, float) else vv for kk, vv in v.items()} if isinstance(v, dict) else v)
                         for k, v in r.items() if k != "note"}}))
else:
    for env in ({}, {"ALOHA_NO_FLOW": "1"}):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)
