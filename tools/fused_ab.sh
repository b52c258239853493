#!/bin/bash
# A/B of the fused single-launch forward transform against the two-launch form, plus ablation builds
# (abl1 = no butterflies: memory + scheduling skeleton; abl2 = no global loads/stores: arithmetic + LDS skeleton).
# usage: tools/fused_ab.sh <tag> ; needs lib variants built by `python -m aloha_amd.build --variant=...`
export ALOHA_NTT_FUSED=256   # (the fused forms are off by default; "two launches" lines override with 0)
tag=${1:-r02}
out=gpurun_out/${tag}_fused_ab.txt
: > $out
run() {  # label, env...
  label=$1; shift
  r=$(env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>>gpurun_out/${tag}_fused_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3f M NTT/s  frac %.3f  kernels/step %d' % (d['ms_per_step'], d['value']/1e6, d['roofline']['frac'], d['config']['kernels_per_step']))")
  echo "$label: $r" | tee -a $out
}
run "fused default" A=1
run "two launches" ALOHA_NTT_FUSED=0
run "fused delay 2" ALOHA_NTT_FUSED_DELAY=2
run "fused delay 8" ALOHA_NTT_FUSED_DELAY=8
run "fused 2 wg/cu" ALOHA_NTT_FUSED_WGS=2
for v in w4 abl1 abl2; do
  lib=$PWD/aloha_amd/lib/libaloha_vp_$v.so
  [ -f $lib ] || continue
  run "fused $v" ALOHA_VP_LIB=$lib
  run "two launches $v" ALOHA_VP_LIB=$lib ALOHA_NTT_FUSED=0
done
ALOHA_NTT_FUSED_STATS=1 timeout -k 10 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "aloha fused" | tail -3 | tee -a $out
