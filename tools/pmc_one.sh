#!/bin/bash
# One PMC group on bench.py for a library variant: bash tools/pmc_one.sh "<counters>" [variant]   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -n "$2" ] && export ALOHA_VP_LIB=$R/aloha_amd/lib/libaloha_vp_$2.so
OUT=$R/gpurun_out/pmc_one_${2:-main}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT.log 2>&1
cd $R && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in acc.items():
    if "ntt_" not in k: continue
    for c, vals in cs.items():
        g = max(v[0] for v in vals); big = [v[1] for v in vals if v[0] == g]
        print(k, c, sum(big) / len(big))
PY
