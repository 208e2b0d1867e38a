#!/bin/bash
# Every bench workload once (device-resident, no CPU baseline / PCIe leg): one line per workload into gpurun_out/bench_all.txt
# usage: bash tools/gpu_bench_all.sh [part]   part 1: dichromats, honeybee, spectral, mantis; part 2: the 14 plane-program species at 1080p and 4K
out=gpurun_out/bench_all_$1.txt
: > $out
if [ "$1" = "1" ]; then
  W="cat_1080p cat_4k dog_1080p dog_4k wolf_1080p lion_1080p squirrel_1080p sheep_1080p honeybee_1080p honeybee_4k honeybee_mst_1080p honeybee_mst_4k spectral_4k_12x31 spectral_4k_10x81 spectral_1080p_12x31 mantis_1080p mantis_4k"
else
  W=""
  for m in reindeer rat_uv goldfish damselfish anableps anchovy guppy morpho heliconius pieris hummingbird kestrel jumping_spider dragonfly; do W="$W ${m}_1080p ${m}_4k"; done
fi
for w in $W; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-e2e > gpurun_out/_one.json 2>gpurun_out/_one.err \
    && python -c "import json; d=json.loads(open('gpurun_out/_one.json').readlines()[-1]); r=d['roofline']; print('%-24s %10.1f MP/s  %9.4f ms/step  %2d frames/step  frac %.4f  parity %s' % ('$w', d['value'], d['ms_per_step'], d['config']['frames_per_step_per_gpu'], r['frac'], d['parity_checked']))" | tee -a $out \
    || { echo "$w FAILED" | tee -a $out; tail -2 gpurun_out/_one.err; }
done
