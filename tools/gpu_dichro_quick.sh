#!/bin/bash
# quick loop for the dichromat kernels: parity tests, then bench lines of the main species
set -o pipefail
timeout -k 10 400 python -m pytest tests/test_dichromat_gpu.py tests/test_geometry_gpu.py -m gpu -x -q > gpurun_out/pytest_dichro.log 2>&1; tail -3 gpurun_out/pytest_dichro.log
for wl in ${WLS:-cat_1080p cat_4k dog_1080p wolf_1080p lion_1080p squirrel_1080p}; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-e2e > gpurun_out/bq_$wl.json 2> gpurun_out/bq_$wl.err
  python -c "
import json; r=json.load(open('gpurun_out/bq_$wl.json')); print('$wl', r['value'], 'MP/s', r['roofline']['us_per_launch'], 'us', 'parity', r.get('parity_checked'))"
done
