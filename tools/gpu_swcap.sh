#!/bin/bash
# A/B: full-width strips vs strips narrowed so the producer's decode needs one pass less (AVX_MARCH_SWCAP), geometry re-tuned
set -o pipefail
AVX_MARCH_SWCAP=112 timeout -k 10 400 python -m pytest tests/test_dichromat_gpu.py -m gpu -x -q > gpurun_out/pytest_swcap.log 2>&1 || { tail -20 gpurun_out/pytest_swcap.log; exit 1; }
tail -1 gpurun_out/pytest_swcap.log
run() {  # workload cap
  AVX_MARCH_SWCAP=$2 AVX_MARCH_NG=64 AVX_MARCH_NOSEED=1 AVX_TUNE_LOG=1 timeout -k 10 200 python bench.py --workload $1 --no-cpu-baseline --no-e2e > gpurun_out/swcap.json 2> gpurun_out/swcap.err || { tail -5 gpurun_out/swcap.err; exit 1; }
  python -c "
import json; r=json.load(open('gpurun_out/swcap.json')); print('cap=$2', '$1', r['value'], 'MP/s', r['roofline']['us_per_launch'], 'us', r.get('parity_checked'))"
  grep "avx tune" gpurun_out/swcap.err | tr '\n' ' '; echo
}
run wolf_1080p 0 && run wolf_1080p 112 && run wolf_1080p 96 && run lion_1080p 0 && run lion_1080p 112 && run squirrel_1080p 0 && run squirrel_1080p 240 && run squirrel_1080p 128 && run dog_1080p 0 && run dog_1080p 96 && run dog_4k 0 && run dog_4k 96
