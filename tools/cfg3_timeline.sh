#!/bin/bash
# Config 3 on the GPU box: the perf figures (tests/perf/bench_configs.py config3) and a rocprofv3 kernel trace of the same run.
# Usage: bash tools/cfg3_timeline.sh <tag>
TAG=${1:-cfg3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/tests/perf/bench_configs.py config3 > $O/config3.json 2> $O/config3.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/tests/perf/bench_configs.py config3 > $O/config3_prof.json 2> $O/prof.log
cp $(find $O/prof -name 't_kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 - "$(find $O/prof -name 't_kernel_trace.csv' | head -1)" > $O/timeline_tail.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not r["Kernel_Name"].startswith("__amd_rocclr")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def show(rs):
    prev = None
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].replace("aloha::k::(anonymous namespace)::", "").replace("void ", "")[:60]
        print(f"dur {(e - s) / 1e3:7.2f} us  gap {((s - prev) / 1e3 if prev else 0):6.2f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} x {r['Workgroup_Size_X']:>4s}  {nm}")
        prev = e
prod = [i for i, r in enumerate(rows) if "poly_product" in r["Kernel_Name"]]
if prod:
    print("-- last 6 product launches"); show(rows[prod[-6]:prod[-1] + 1])
print("-- last 10 dispatches"); show(rows[-10:])
PY
rm -rf $O/prof
cat $O/config3.json; cat $O/timeline_tail.txt; python3 $R/tools/kernel_stats.py $O/kernel_stats.csv | head -12
