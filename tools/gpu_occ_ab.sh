#!/bin/bash
# Same-box A/B of the persistent MST++ kernels' occupancy share (AVX_MST_OCC_* percent of the resident-workgroup cap) x frame lanes.
# usage: bash tools/gpu_occ_ab.sh "<lanes>:<ffn>:<tail>:<conv>" ...   (100 = full cap)
mkdir -p gpurun_out
for cfg in "$@"; do
  IFS=: read L F T C <<< "$cfg"
  AVX_BENCH_MST_LANES=$L AVX_MST_OCC_FFN=$F AVX_MST_OCC_TAIL=$T AVX_MST_OCC_CONV=$C timeout -k 10 200 python bench.py --workload honeybee_mst_4k --no-cpu-baseline --no-e2e --no-legs --steps ${STEPS:-10} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lanes=$L ffn=$F tail=$T conv=$C', d['ms_per_step'], d['config']['fps'], d['parity_checked'])" | tee -a gpurun_out/occ_ab.txt
done
