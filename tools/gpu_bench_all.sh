#!/bin/bash
# Bench lines for the DESIGN.md measurement table: every workload family, bounded CPU baseline each.
set -o pipefail
mkdir -p gpurun_out/all
: > gpurun_out/all/summary.txt
for wl in ${WORKLOADS:-cat_1080p cat_4k dog_1080p dog_4k wolf_1080p honeybee_1080p honeybee_4k mantis_1080p mantis_4k reindeer_1080p rat_uv_1080p goldfish_1080p damselfish_1080p anableps_1080p anchovy_1080p guppy_1080p morpho_1080p heliconius_1080p pieris_1080p hummingbird_1080p kestrel_1080p jumping_spider_1080p dragonfly_1080p reindeer_4k kestrel_4k}; do
  timeout -k 10 300 python bench.py --workload $wl ${BENCH_ARGS:---cpu-seconds 3 --steps 10 --warmup 2 --no-e2e} > gpurun_out/all/$wl.json 2> gpurun_out/all/$wl.err || { tail -5 gpurun_out/all/$wl.err; echo "$wl FAILED" >> gpurun_out/all/summary.txt; continue; }
  python - <<PY >> gpurun_out/all/summary.txt
import json
r=json.load(open("gpurun_out/all/$wl.json"))
print("$wl", r["value"], "MP/s", r["config"]["fps"], "fps", r["roofline"]["achieved"], r["roofline"]["unit"], "frac", r["roofline"]["frac"], "parity", r.get("parity_checked"), r.get("parity_stats", ""), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
  tail -1 gpurun_out/all/summary.txt
done
