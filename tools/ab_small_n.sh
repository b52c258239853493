#!/bin/bash
# A/B of library variants on the single-pass sizes (N = 2^12, 2^13, 2^14), one box, interleaved.  Usage: bash tools/ab_small_n.sh <tag> name...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG.txt
cd $R
: > $O
for rep in 1 2 3; do
  for v in "$@"; do
    lib=""; [ "$v" != "main" ] && lib=$R/aloha_amd/lib/libaloha_vp_$v.so
    for ln in 12 13 14; do
      r=$(ALOHA_VP_LIB=$lib PRIMES=1 timeout -k 10 120 python tests/perf/bench_small_n.py $ln 2>/dev/null | tr '\n' ' ')
      echo "$v | N=2^$ln: $r" >> $O
    done
  done
done
cat $O
