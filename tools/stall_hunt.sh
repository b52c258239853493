#!/bin/bash
# Are the rare multi-millisecond stalls of tests/perf/product_repeat.py inside a kernel or between kernels?  rocprofv3 kernel trace
# of the script; prints the longest kernels and the longest gaps.  Usage: bash tools/stall_hunt.sh <tag> [rounds]
TAG=${1:-stall}; ROUNDS=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/tests/perf/product_repeat.py $ROUNDS > $O/repeat.txt 2> $O/prof.log
python3 - "$(find $O/prof -name 't_kernel_trace.csv' | head -1)" > $O/longest.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60], int(r["Grid_Size_X"])) for r in rows)
print("longest kernels (us):")
for d in durs[-8:]:
    print("  %10.1f  grid %8d  %s" % (d[0], d[2], d[1]))
gaps = []
for a, b in zip(rows, rows[1:]):
    if "poly_product" in a["Kernel_Name"] and "poly_product" in b["Kernel_Name"]:
        gaps.append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, int(a["Grid_Size_X"])))
gaps.sort()
print("longest gaps between consecutive product launches (us):")
for g in gaps[-8:]:
    print("  %10.1f  grid %8d" % g)
print("product launches:", sum("poly_product" in r["Kernel_Name"] for r in rows))
PY
rm -rf $O/prof
grep -v amdgpu $O/repeat.txt; cat $O/longest.txt
