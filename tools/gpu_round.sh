#!/bin/bash
# One GPU-box round: parity tests -> smoke -> bench -> rocprofv3 kernel trace.  Logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc" | tee -a gpurun_out/progress.log
[ $rc -ne 0 ] && exit $rc
echo "== smoke" | tee -a gpurun_out/progress.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
for wl in ${WORKLOADS:-cat_1080p dog_1080p dog_4k}; do
  echo "== bench $wl" | tee -a gpurun_out/progress.log
  timeout -k 10 400 python bench.py --workload $wl > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err || { tail -20 gpurun_out/bench_$wl.err; exit 1; }
  cat gpurun_out/bench_$wl.json
done
echo "== rocprofv3" | tee -a gpurun_out/progress.log
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 1; }
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
echo "== done" | tee -a gpurun_out/progress.log
