#!/bin/bash
# Per-kernel times of one key-switch config under two settings of an environment switch (rocprofv3 --kernel-trace --stats).
# Usage: bash tools/ks_kernel_ab.sh <tag> <VAR> -- <bench_keyswitch_sharded.py args...>
TAG=$1; VAR=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python3 $R/tools/bench_keyswitch_sharded.py --reps 10 "$@" > $O/$v.json 2> $O/$v.log
  python3 $R/tools/kernel_timeline.py $(find $O/$v -name 't_kernel_trace.csv' | head -1) 10 > $O/${v}_timeline.txt
  rm -rf $O/$v
  echo "== $VAR $v"; tail -n 9 $O/${v}_timeline.txt
done
