#!/bin/bash
# Round 4, VERDICT item 4: the row tile's LDS image, ALOHA_ROW_BMAP = 1 (conflict-free on all four patterns, the default build)
# against = 0 (rounds 1-3; lib/libaloha_vp_bmap0.so from `python -m aloha_amd.build --plain=bmap0:ALOHA_ROW_BMAP=0`), on one box:
# headline bench interleaved, the key-switch configs, and the LDS counters of both.   Usage: bash tools/r04_bmap_ab.sh <tag>
TAG=${1:-r04_bmap}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B0=$R/aloha_amd/lib/libaloha_vp_bmap0.so
{
  echo "== headline (bench.py --steps 300 --warmup 30), interleaved"
  for rep in 1 2 3; do
    for v in main bmap0; do
      lib=""; [ "$v" != "main" ] && lib=$B0
      ALOHA_VP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v:', round(d['value']), 'NTT/s', round(d['ms_per_step'],4), 'ms/step  frac', round(d['roofline']['frac'],4))"
      ALOHA_VP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 --inverse 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v inverse:', round(d['value']), 'NTT/s', round(d['ms_per_step'],4), 'ms/step')"
    done
  done
  echo "== key-switch configs (tools/bench_keyswitch_sharded.py --reps 20), interleaved"
  for rep in 1 2; do
    for v in main bmap0; do
      lib=""; [ "$v" != "main" ] && lib=$B0
      for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
        r=$(ALOHA_VP_LIB=$lib timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --reps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
        echo "$v | $cfg: $r"
      done
    done
  done
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in main bmap0; do
  lib=""; [ "$v" != "main" ] && lib=$B0
  export ALOHA_VP_LIB=$lib
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_$v -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-settle --no-configs > $O/pmc_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcks_$v -- python3 $R/tools/bench_keyswitch_sharded.py --L 24 --alpha 4 --op multiply --reps 4 > $O/pmcks_$v.log 2>&1
done
unset ALOHA_VP_LIB
cd $R && python3 - $O > $O/lds_counters.txt <<'PY'
import csv, glob, sys, collections
for v in ("main", "bmap0"):
    for kind in ("pmc", "pmcks"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{sys.argv[1]}/{kind}_{v}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("aloha::k::(anonymous namespace)::", "").replace("void ", "")[:44]
                acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for k, cs in sorted(acc.items()):
            if "ntt_" not in k: continue
            line = []
            for c, vals in sorted(cs.items()):
                g = max(x[0] for x in vals); big = [x[1] for x in vals if x[0] == g]
                line.append(f"{c}={sum(big) / len(big):.4g}")
            print(f"{v:5s} {kind:5s} {k:44s}", " ".join(line))
PY
rm -rf $O/pmc_main $O/pmc_bmap0 $O/pmcks_main $O/pmcks_bmap0
cat $O/lds_counters.txt
