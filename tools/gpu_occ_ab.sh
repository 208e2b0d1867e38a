#!/bin/bash
# Same-box A/B of the number of frame lanes of the headline bench (AVX_BENCH_MST_LANES).  usage: bash tools/gpu_occ_ab.sh <lanes> [<lanes> ...]
# (Round 3 also used it with an experimental per-kernel-class cap on the persistent kernels' resident workgroups -- AVX_MST_OCC_FFN / _TAIL / _CONV, percent -- to let two
#  lanes co-reside on each CU; that measured slower (83.9 vs 82.6 ms per four-frame step at 50 %) and the knob was not kept in the library: DESIGN 4.3.)
mkdir -p gpurun_out
for L in "$@"; do
  AVX_BENCH_MST_LANES=$L timeout -k 10 200 python bench.py --workload honeybee_mst_4k --no-cpu-baseline --no-e2e --no-legs --steps ${STEPS:-10} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lanes=$L', d['ms_per_step'], d['config']['fps'], d['parity_checked'])" | tee -a gpurun_out/lanes_ab.txt
done
