#!/bin/bash
# same box, same build: bench lines with the seeded launch geometries, then with the tuner measuring them afresh
set -o pipefail
WLS=${WLS:-dog_1080p dog_4k wolf_1080p lion_1080p squirrel_1080p}
for mode in seed tune seed; do
  for wl in $WLS; do
    if [ $mode = tune ]; then export AVX_MARCH_NOSEED=1 AVX_TUNE_LOG=1; else unset AVX_MARCH_NOSEED AVX_TUNE_LOG; fi
    timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-e2e > gpurun_out/sc_${mode}_$wl.json 2> gpurun_out/sc_${mode}_$wl.err || exit 1
    python -c "
import json; r=json.load(open('gpurun_out/sc_${mode}_$wl.json')); print('$mode', '$wl', r['value'], 'MP/s', r['roofline']['us_per_launch'], 'us')"
    grep "avx tune" gpurun_out/sc_${mode}_$wl.err || true
  done
done
