#!/bin/bash
# quick A/B loop for the MST++ route: unit tests, bench line, steady-state per-kernel breakdown
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_mstpp.py -m gpu -x -q > gpurun_out/pytest_mst.log 2>&1; tail -3 gpurun_out/pytest_mst.log
timeout -k 10 200 python bench.py --workload ${WL:-honeybee_mst_1080p} --no-cpu-baseline > gpurun_out/b_mst.json 2> gpurun_out/b_mst.err; cut -c1-200 gpurun_out/b_mst.json
bash tools/gpu_prof_mst.sh > gpurun_out/prof_mst_out.txt 2>&1; head -${N:-16} gpurun_out/prof_mst_steady.txt | cut -c1-170
