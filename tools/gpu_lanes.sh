#!/bin/bash
# frame lanes of the honeybee tail: parity tests, then throughput by lane count
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "honeybee or uv or bee" > gpurun_out/pytest_lanes.log 2>&1 || { tail -20 gpurun_out/pytest_lanes.log; exit 1; }
tail -1 gpurun_out/pytest_lanes.log
for wl in honeybee_1080p honeybee_4k; do
  for L in 1 2 4 8; do
    AVX_UV_LANES=$L timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-e2e > gpurun_out/lanes.json 2> gpurun_out/lanes.err || { tail -5 gpurun_out/lanes.err; exit 1; }
    python -c "
import json; r=json.load(open('gpurun_out/lanes.json')); print('lanes=$L', '$wl', r['value'], 'MP/s', r['roofline']['us_per_launch'], 'us/step', r.get('parity_checked'))"
  done
done
