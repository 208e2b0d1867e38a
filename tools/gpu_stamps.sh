#!/bin/bash
# in-kernel phase stamps of the marching kernel, producer wave then a compute wave (stamp instantiations: R = 6, 14, cat)
set -o pipefail
for wl in ${WLS:-wolf_1080p dog_1080p cat_1080p}; do
  for ab in 0 64; do
    AVX_STAMPS=1 AVX_ABLATE=$ab timeout -k 10 200 python bench.py --workload $wl --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e 2>&1 >/dev/null | grep "avx march stamps" | tail -1 | sed "s/^/$wl ablate=$ab /"
  done
done
