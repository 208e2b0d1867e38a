#!/bin/bash
# GPU-busy fraction of a workload's steady state: sum of kernel durations / wall span of the last half of the dispatches.
set -o pipefail
export TMPDIR=/tmp
WL=${WL:-reindeer_1080p}
rm -rf gpurun_out/busy
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/busy -- python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > gpurun_out/busy.json 2> gpurun_out/busy.err || { tail -3 gpurun_out/busy.err; exit 1; }
f=$(find gpurun_out/busy -name "*kernel_trace.csv" | head -1)
python - "$f" "$WL" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[len(rows) // 2:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
r = json.load(open("gpurun_out/busy.json"))
print(sys.argv[2], "dispatches", len(tail), "kernel time %.2f ms" % (busy / 1e6), "wall span %.2f ms" % (span / 1e6), "GPU busy %.1f%%" % (100.0 * busy / span), "| bench", r["value"], "MP/s")
PY
