#!/bin/bash
# A/B timing of library variants built with `python -m aloha_amd.build --variant=name:MACRO=..` (GPU box).
# Usage: bash tools/ab_libs.sh [--inverse] name1 name2 ...   ("main" = the default library)
EXTRA=""
if [ "$1" == "--inverse" ]; then EXTRA="--inverse"; shift; fi
for v in "$@"; do
  lib=""; [ "$v" != "main" ] && lib="$PWD/aloha_amd/lib/libaloha_vp_$v.so"
  for rep in 1 2; do
    ALOHA_VP_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 30 $EXTRA 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v:', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
  done
done
