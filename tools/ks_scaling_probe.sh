#!/bin/bash
# Per-rank share of the sharded key-switch on ONE GPU (exchanges stubbed): how the time of rank 0 shrinks as the
# limbs are spread over W ranks -- launch-bound or not.  Output: gpurun_out/ks_probe.jsonl
set -e
mkdir -p gpurun_out
out=gpurun_out/ks_probe.jsonl
: > $out
for W in 1 2 4 8; do
  for op in multiply rotate; do
    timeout -k 10 120 python tools/bench_keyswitch_sharded.py --L 24 --alpha 4 --op $op --stub-world $W --reps 20 >> $out
  done
done
timeout -k 10 120 python tools/bench_keyswitch_sharded.py --L 24 --alpha 4 --op multiply --stub-world 8 --batch 4 --reps 20 >> $out
timeout -k 10 120 python tools/bench_keyswitch_sharded.py --L 32 --alpha 4 --op rotate --stub-world 8 --reps 20 >> $out
timeout -k 10 120 python tools/bench_keyswitch_sharded.py --L 24 --alpha 0 --op multiply --stub-world 8 --reps 20 >> $out
cat $out
