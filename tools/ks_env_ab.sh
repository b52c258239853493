#!/bin/bash
# A/B of an environment switch on the two key-switch configs, interleaved, on one box, with the kernel timeline of the last repetition:
#   bash tools/ks_env_ab.sh <tag> VAR=1 [VAR2=1 ...]      -> gpurun_out/<tag>.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG.txt
: > $O
cd $R
for rep in 1 2 3; do
  for env in "A=1" "$@"; do
    for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
      r=$(env $env timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --rom 16384 --reps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
      echo "$env | $cfg: $r" >> $O
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for env in "A=1" "$@"; do
  for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
    d=$R/gpurun_out/${TAG}_prof
    rm -rf $d
    env $env timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 $R/tools/bench_keyswitch_sharded.py $cfg --rom 16384 --reps 10 > /dev/null 2>&1
    echo "== $env | $cfg" >> $O
    python3 $R/tools/kernel_timeline.py $(find $d -name 't_kernel_trace.csv' | head -1) 10 | head -14 >> $O
    rm -rf $d
  done
done
cat $O
