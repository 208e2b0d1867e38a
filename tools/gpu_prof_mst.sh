#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mst
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mst -- python bench.py --workload ${WL:-honeybee_mst_1080p} --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/prof_mst.json 2> gpurun_out/prof_mst.err || { tail -5 gpurun_out/prof_mst.err; exit 1; }
f=$(find gpurun_out/prof_mst -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6, "kernels", len(rows))
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls={r["Calls"]:>6}  avg_us={float(r["AverageNs"])/1e3:9.1f}  {r["Name"][:110]}')
PY
