#!/bin/bash
# A rank's share of a key-switch spread over W GPUs (exchanges stubbed) under environment switches, interleaved on one box:
#   bash tools/w8_share_ab.sh <W> "VAR=1" ...       ("A=1" = defaults)
W=$1; shift
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
for e in "$@"; do
  for cfg in "--L 24 --alpha 4 --op multiply" "--L 32 --alpha 4 --op rotate"; do
  r=$(env $e timeout -k 10 200 python tools/bench_keyswitch_sharded.py $cfg --stub-world $W --reps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  launches %.0f' % (d['ms'], d['launches_per_op']))")
  echo "W=$W $e | $cfg: $r"
  done
done
done
