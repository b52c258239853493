#!/bin/bash
# Kernel timelines of the key-switch configs on the GPU box (rocprofv3 --kernel-trace): for each of
#   hybrid alpha = 4, L = 24 HMult   |  hybrid alpha = 4, L = 32 rotate  |  W = 8 rank-0 share of the first
# the per-dispatch timeline of the LAST repetition and the per-kernel sums.  Usage: bash tools/ks_timeline.sh <tag>
set -e
TAG=${1:-ks}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, reps, args...
  local name=$1 reps=$2; shift 2
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 $R/tools/bench_keyswitch_sharded.py --reps $reps "$@" \
    > $O/$name.json 2> $O/$name.log
  local csv=$(find $O/$name -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/kernel_timeline.py $csv $reps > $O/${name}_timeline.txt
  cp $(find $O/$name -name 't_kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
  rm -rf $O/$name
}
run hmult_L24_a4 10 --L 24 --alpha 4 --op multiply
run rotate_L32_a4 10 --L 32 --alpha 4 --op rotate
run rotate_L32_a4_rom16k 10 --L 32 --alpha 4 --op rotate --rom 16384
run hmult_L24_a4_w8_rank0 10 --L 24 --alpha 4 --op multiply --stub-world 8
tail -n 12 $O/hmult_L24_a4_timeline.txt
cat $O/hmult_L24_a4.json
