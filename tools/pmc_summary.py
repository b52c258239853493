#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc CSVs written by tools/collect_pmc.sh into one JSON summary:
per-kernel averages per launch of every counter, and the HBM bytes per bench step with the
guide's gfx950 correction (FETCH_SIZE counts 64 B per 128-B request on wide streaming reads:
doubled; WRITE_SIZE exact; both in KiB -- MI355X_MICROARCH.md, HBM section)."""
from __future__ import annotations

import csv
import glob
import json
import os
import sys
from collections import defaultdict

LIMBS, N = 4096, 1 << 16          # bench.py default batch


def short(name: str) -> str:
    for k in ("ntt_col_kernel", "ntt_row_kernel"):
        if k in name:
            return k
    return ""


def main() -> int:
    root = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows.append((k, int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    # the bench also launches a few single-limb transforms (its correctness guard): keep the full-batch launches only
    full = defaultdict(int)
    for k, g, _, _ in rows:
        full[k] = max(full[k], g)
    acc = defaultdict(lambda: defaultdict(list))
    for k, g, c, v in rows:
        if g == full[k]:
            acc[k][c].append(v)
    out = {"note": "rocprofv3 --pmc, one counter group per run, bench.py --steps 3 --warmup 2 "
                   f"({LIMBS} limbs, N=2^16); averages per launch",
           "commit": os.environ.get("ALOHA_COMMIT", "")}      # the tree the counters were collected on (tools/collect_pmc.sh)
    for k, cs in sorted(acc.items()):
        for c, vals in sorted(cs.items()):
            out[f"{k}.{c}"] = sum(vals) / len(vals)
    try:
        fetch = sum(out[f"{k}.FETCH_SIZE"] for k in ("ntt_col_kernel", "ntt_row_kernel"))
        write = sum(out[f"{k}.WRITE_SIZE"] for k in ("ntt_col_kernel", "ntt_row_kernel"))
        hbm = (2 * fetch + write) * 1024
        alg = 16 * N * LIMBS
        out["derived"] = {"algorithmic_bytes_per_step": alg, "hbm_bytes_per_step_corrected": hbm, "ratio": hbm / alg}
    except KeyError as e:
        out["derived"] = {"error": f"missing counter {e}"}
    print(json.dumps(out, indent=1, sort_keys=True))
    return 0


if __name__ == "__main__":
    sys.exit(main())
