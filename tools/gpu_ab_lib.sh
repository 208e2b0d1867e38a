#!/bin/bash
# same-box A/B of two builds: animal-vision_amd/libavx_head.so (the previous build, copied there by hand) vs libavx.so
set -o pipefail
L=animal-vision_amd
cp $L/libavx.so $L/libavx_new.so
for round in 1 2; do
  for which in head new; do
    cp $L/libavx_$which.so $L/libavx.so
    for wl in ${WLS:-cat_1080p dog_1080p wolf_1080p lion_1080p squirrel_1080p}; do
      ${ENVV:-env} timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-e2e > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; cp $L/libavx_new.so $L/libavx.so; exit 1; }
      python -c "
import json; r=json.load(open('gpurun_out/ab.json')); print('$which', '$wl', r['value'], 'MP/s', r['roofline']['us_per_launch'], 'us')"
    done
  done
done
cp $L/libavx_new.so $L/libavx.so
