#!/bin/bash
# MST++ route: unit tests of the fused kernels, the large-frame parity tests (errors printed), then the 1080p / 4K workloads with
# the kernels named in $AB switched off and on (same box, same process order).  Usage: AB="AVX_MST_NO_FFN_FUSED" tools/gpu_mst_ab.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_mstpp.py -m gpu -q -x -rP -p no:cacheprovider > gpurun_out/mst_tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|part [0-9]|attention matrices|Error|error" gpurun_out/mst_tests.log | tail -30
for wl in honeybee_mst_1080p honeybee_mst_4k; do
  for var in ${AB:-AVX_MST_NO_FFN_FUSED}; do
    env $var=${ABVAL:-1} timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/ab_${wl}_off_${var}.json 2>/dev/null || echo "bench failed ($wl, $var off)"
  done
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/ab_${wl}_on.json 2>/dev/null || echo "bench failed ($wl on)"
  python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_${wl}_*.json")):
    try:
        r=json.load(open(f)); print(f.split("/")[-1], "ms/step", r["ms_per_step"], "MP/s", r["value"], "fps", r["config"]["fps"], "parity", r.get("parity_checked"))
    except Exception as e: print(f, "unreadable", e)
PY
done
