#!/bin/bash
# PMC counters of any python command on the GPU box, one counter group per run (PMC only, no trace domains), per kernel name the
# average per launch over the largest grid seen.  Usage: bash tools/pmc_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>_pmc.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_pmc
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 $SCRIPT "$@" > $OUT/$name.log 2>&1
done
cd $R && python3 - "$OUT" > $R/gpurun_out/${TAG}_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aloha::k::(anonymous namespace)::", "").replace("void ", "")[:48]
        acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in sorted(acc.items()):
    if "__amd" in k: continue
    print(k)
    for c, vals in sorted(cs.items()):
        g = max(v[0] for v in vals); big = [v[1] for v in vals if v[0] == g]
        print(f"    {c:24s} {sum(big) / len(big):14.5g}   (grid {g}, {len(big)} launches)")
PY
rm -rf $OUT
cat $R/gpurun_out/${TAG}_pmc.txt
