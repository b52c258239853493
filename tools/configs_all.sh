#!/bin/bash
# Every BASELINE config on this box: GPU numbers (tests/perf/bench_configs.py) with the CPU baseline beside them
# (tests/perf/bench_configs_cpu.py: the oracle on this host's cores) -> gpurun_out/<tag>_configs_all.json
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout -k 10 600 python tests/perf/bench_configs.py > gpurun_out/${tag}_configs_gpu.json 2> gpurun_out/${tag}_configs_gpu.err || echo "gpu configs failed"
timeout -k 10 900 python tests/perf/bench_configs_cpu.py > gpurun_out/${tag}_configs_cpu.json 2> gpurun_out/${tag}_configs_cpu.err || echo "cpu configs failed"
python - <<PY
import json
g = json.load(open("gpurun_out/${tag}_configs_gpu.json")); c = json.load(open("gpurun_out/${tag}_configs_cpu.json"))
out = {"note": "GPU: one MI355X, tests/perf/bench_configs.py.  cpu_baseline: the oracle (plain-C port of the reference's arithmetic and ISA machine; "
               "kind 'port') on this box's host cores, one thread and all cores (T machines side by side), tests/perf/bench_configs_cpu.py",
       "host_cores_used": c.get("host_cores_used")}
for k in sorted(set(g) | set(c) - {"host_cores_used"}):
    out[k] = {"gpu": g.get(k), "cpu_baseline": c.get(k)}
json.dump(out, open("gpurun_out/${tag}_configs_all.json", "w"), indent=1)
print(json.dumps({k: (v["gpu"] is not None, v["cpu_baseline"] is not None) for k, v in out.items() if isinstance(v, dict)}))
PY
