#!/bin/bash
# Quick perf iteration on the GPU box: bench lines only (each line carries parity_checked).
set -o pipefail
mkdir -p gpurun_out
for wl in ${WORKLOADS:-cat_1080p dog_4k wolf_1080p}; do
  timeout -k 10 300 python bench.py --workload $wl ${BENCH_ARGS:---cpu-seconds 1 --steps 30} > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err || { tail -20 gpurun_out/bench_$wl.err; exit 1; }
  python - <<PY
import json
r=json.load(open("gpurun_out/bench_$wl.json"))
print("$wl", r["value"], "MP/s", r["roofline"]["achieved"], "GB/s frac", r["roofline"]["frac"], "us/launch", r["roofline"]["us_per_launch"], "parity", r.get("parity_checked"), r.get("parity_stats", ""), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
done
