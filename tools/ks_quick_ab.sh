#!/bin/bash
# quick interleaved A/B of the two key-switch configs over (library variant, environment) pairs on one box:
#   bash tools/ks_quick_ab.sh <tag> "name|VAR=1" ...     name = main or a variant of aloha_amd/lib; VAR=1 may be empty ("main|")
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG.txt
: > $O
cd $R
for rep in 1 2 3; do
  for spec in "$@"; do
    v=${spec%%|*}; e=${spec#*|}; [ -z "$e" ] && e="A=1"
    lib=""; [ "$v" != "main" ] && lib=$R/aloha_amd/lib/libaloha_vp_$v.so
    for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
      r=$(env ALOHA_VP_LIB=$lib $e timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --rom 16384 --reps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
      echo "$spec | $cfg: $r" >> $O
    done
  done
done
cat $O
