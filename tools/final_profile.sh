#!/bin/bash
# Round-end evidence for the kernels of this commit (run on the GPU box): default bench line, rocprofv3 kernel
# stats of the same command, per-kernel roofline table, inverse side metric.  Copies go to profiles/ by hand.
# Usage: bash tools/final_profile.sh <tag>
set -e
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o bench -- python3 $R/bench.py --no-cpu-baseline \
  > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_prof.log
cp $R/gpurun_out/${TAG}_prof/bench_kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
# the configs leg launches the same column/row kernels on smaller batches, which drags their AVERAGE in the stats above; the headline
# alone (what roofline.achieved is quoted on) is this second trace
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_headline -o bench -- python3 $R/bench.py --no-cpu-baseline --no-configs \
  > $R/gpurun_out/${TAG}_bench_headline_under_rocprof.json 2>> $R/gpurun_out/${TAG}_prof.log
cp $R/gpurun_out/${TAG}_prof_headline/bench_kernel_stats.csv $R/gpurun_out/${TAG}_bench_headline_kernel_stats.csv
timeout -k 10 200 python3 $R/bench.py --inverse --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_inverse.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout -k 10 200 python3 $R/tests/perf/bench_kernels.py > $R/gpurun_out/${TAG}_kernels.json 2> $R/gpurun_out/${TAG}_kernels.err
cd $R && python3 tools/kernel_stats.py gpurun_out/${TAG}_bench_headline_kernel_stats.csv | head -4
cut -c1-300 gpurun_out/${TAG}_bench.json
