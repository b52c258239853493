#!/usr/bin/env python3
"""Timeline of the LAST repetition of an operation from a rocprofv3 --kernel-trace CSV:

    python tools/kernel_timeline.py <..._kernel_trace.csv> <reps>

The trace of `reps` identical repetitions (after whatever warm-up came first) is cut into its last
len/reps... no: the last repetition is found as the last K dispatches, K = (dispatches after the
warm-up marker) / reps, where the marker is the dispatch count modulo reps.  Prints per dispatch:
start offset, duration, gap to the previous kernel's end, grid, kernel name; then the sums."""
import csv
import sys


def short(name: str) -> str:
    for p in ("aloha::k::(anonymous namespace)::", "aloha::k::", "void "):
        name = name.replace(p, "")
    return name[:70]


def main() -> int:
    path, reps = sys.argv[1], int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # dispatches per repetition, when known
    rows = [r for r in csv.DictReader(open(path)) if not r["Kernel_Name"].startswith("__amd_rocclr")]   # (host copies of the set-up / read-out)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if not per:
        # the repetitions are the periodic tail of the trace: find the smallest period p such that the last
        # reps * p kernel names repeat with period p
        names = [r["Kernel_Name"] for r in rows]
        per = 0
        for p in range(1, len(names) // reps + 1):
            tail = names[len(names) - reps * p:]
            if all(tail[i] == tail[i % p] for i in range(len(tail))):
                per = p
                break
        if not per:
            print("no periodic tail found", file=sys.stderr)
            return 1
    last = rows[len(rows) - per:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end, busy, by = None, 0, {}
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y")))
        nm = short(r["Kernel_Name"])
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {grid:>12s} wg {r['Workgroup_Size_X']:>4s}  {nm}")
        busy += e - s
        by[nm] = by.get(nm, 0) + (e - s)
        prev_end = e if prev_end is None else max(prev_end, e)
    span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"dispatches {per}  span {span:.1f} us  kernel time {busy / 1e3:.1f} us  gaps {span - busy / 1e3:.1f} us")
    for nm, t in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"  {t / 1e3:8.1f} us  {nm}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
