#!/bin/bash
# bench.py A/B over library variants built with `python -m aloha_amd.build --variant=name:MACRO=1`: tools/ab_variants.sh <tag> name...
tag=$1; shift
out=gpurun_out/${tag}_variants.txt
: > $out
run() {
  label=$1; shift
  r=$(env "$@" timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3f M NTT/s  frac %.4f' % (d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))")
  echo "$label: $r" | tee -a $out
}
run "main" A=1
for v in "$@"; do run "$v" ALOHA_VP_LIB=$PWD/aloha_amd/lib/libaloha_vp_$v.so; done
run "main again" A=1
