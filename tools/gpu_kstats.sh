#!/bin/bash
# Per-kernel time table of one bench workload: rocprofv3 --kernel-trace --stats, top 32 kernels -> gpurun_out/kstats_<workload>.txt
# usage: bash tools/gpu_kstats.sh <workload> [steps]
WL=${1:-honeybee_mst_4k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -o k -- python bench.py --workload $WL --no-cpu-baseline --no-e2e --no-legs --steps ${2:-6} > gpurun_out/kstats.log 2>&1 || { tail -5 gpurun_out/kstats.log; exit 1; }
python - > gpurun_out/kstats_$WL.txt <<PY
import csv, glob
f = glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:32]:
    print("%-100s %6s %10.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf gpurun_out/kstats
cat gpurun_out/kstats_$WL.txt
