#!/bin/bash
export ALOHA_NTT_FUSED=256   # (the fused forms are off by default; "two launches" lines override with 0)
tag=${1:-r02}
out=gpurun_out/${tag}_fused_ab.txt
: > $out
run() {  # label, env...
  label=$1; shift
  r=$(env "$@" ALOHA_NTT_FUSED_STATS=1 timeout -k 10 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>gpurun_out/${tag}_err.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3f M NTT/s  frac %.3f  kernels/step %d' % (d['ms_per_step'], d['value']/1e6, d['roofline']['frac'], d['config']['kernels_per_step']))")
  st=$(grep "aloha fused" gpurun_out/${tag}_err.tmp | tail -1 | sed 's/.*positions, //')
  echo "$label: $r   [$st]" | tee -a $out
}
run "two launches" ALOHA_NTT_FUSED=0
for d in 2 4 8 16; do run "dispatch delay $d" ALOHA_NTT_FUSED_DELAY=$d; done
run "persistent delay 8" ALOHA_NTT_FUSED_DELAY=8 ALOHA_NTT_FUSED_MODE=persistent
for v in abl1 abl2; do
  lib=$PWD/aloha_amd/lib/libaloha_vp_$v.so
  for d in 4 8; do run "dispatch $v delay $d" ALOHA_VP_LIB=$lib ALOHA_NTT_FUSED_DELAY=$d; done
  run "two launches $v" ALOHA_VP_LIB=$lib ALOHA_NTT_FUSED=0
done
rm -f gpurun_out/${tag}_err.tmp
