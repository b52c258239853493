#!/bin/bash
# A/B of library variants on one box, interleaved: headline forward / inverse (bench.py) and the two key-switch configs
# (tools/bench_keyswitch_sharded.py; --rom 16384 as in bench.py's configs leg).  "main" = the default library, other names =
# aloha_amd/lib/libaloha_vp_<name>.so (python -m aloha_amd.build --plain=name:MACRO=..).   Usage: bash tools/ab_r04.sh <tag> [--no-headline] name...
TAG=$1; shift
HEAD=1; if [ "$1" == "--no-headline" ]; then HEAD=0; shift; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG.txt
cd $R
: > $O
for rep in 1 2 3; do
  for v in "$@"; do
    lib=""; [ "$v" != "main" ] && lib=$R/aloha_amd/lib/libaloha_vp_$v.so
    if [ $HEAD == 1 ]; then
      ALOHA_VP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v: fwd', round(d['value']), 'NTT/s', round(d['ms_per_step'],4), 'ms/step  frac', round(d['roofline']['frac'],4))" >> $O
      ALOHA_VP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 --inverse 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v: inv', round(d['value']), 'NTT/s', round(d['ms_per_step'],4), 'ms/step')" >> $O
    fi
    for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
      r=$(ALOHA_VP_LIB=$lib timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --rom 16384 --reps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
      echo "$v | $cfg: $r" >> $O
    done
  done
done
cat $O
