#!/bin/bash
# PMC counters of the key-switch kernels (GPU box): one counter group per run, PMC only (no trace domains), as
# MI355X_MICROARCH.md prescribes.  Per kernel name: the average per launch over the largest grid seen.
# Usage: bash tools/pmc_ks.sh <tag> [bench_keyswitch_sharded.py args...]     -> gpurun_out/<tag>_pmc.txt
TAG=${1:-ks}; shift
ARGS=${@:---L 24 --alpha 4 --op multiply --reps 4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 $R/tools/bench_keyswitch_sharded.py $ARGS > $OUT/$name.log 2>&1
done
cd $R && python3 - "$OUT" > $R/gpurun_out/${TAG}_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aloha::k::(anonymous namespace)::", "").replace("void ", "")[:40]
        acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in sorted(acc.items()):
    if "ntt_" not in k and "dot_" not in k and "sum_" not in k and "ew_" not in k: continue
    line = []
    for c, vals in sorted(cs.items()):
        g = max(v[0] for v in vals); big = [v[1] for v in vals if v[0] == g]
        line.append(f"{c}={sum(big) / len(big):.4g}")
    print(k, " ".join(line))
PY
rm -rf $OUT
cat $R/gpurun_out/${TAG}_pmc.txt
