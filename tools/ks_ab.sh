#!/bin/bash
# single-GPU key-switch timings, A/B over an environment switch: tools/ks_ab.sh <tag> VAR=1
tag=$1; shift
out=gpurun_out/${tag}_ks_ab.txt
: > $out
for env in "A=1" "$@"; do
  for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate" "--L 24 --alpha 0 --op multiply" "--L 32 --alpha 0 --op rotate"; do
    r=$(env $env timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --reps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
    echo "$env | $cfg: $r" | tee -a $out
  done
done
