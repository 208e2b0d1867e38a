#!/bin/bash
# Per-kernel breakdown of the MST++ honeybee route in STEADY STATE: the first frames include MIOpen's solver search, so the
# summary is computed from the kernel trace of the last quarter of the dispatches only.
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mst
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mst -- python bench.py --workload ${WL:-honeybee_mst_1080p} --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/prof_mst.json 2> gpurun_out/prof_mst.err || { tail -5 gpurun_out/prof_mst.err; exit 1; }
f=$(find gpurun_out/prof_mst -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/prof_mst_steady.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[len(rows) * 3 // 4:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
frames = max(1, sum(v[0] for k, v in agg.items() if "k_honeybee" in k or "honeybee_finish" in k) or 1)
tot = sum(v[1] for v in agg.values())
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print(f"steady-state window: {len(tail)} dispatches, kernel time {tot/1e6:.2f} ms, wall span {span/1e6:.2f} ms, distinct kernels {len(agg)}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]/tot*100:5.1f}%  calls={v[0]:5d}  avg_us={v[1]/v[0]/1e3:9.1f}  tot_ms={v[1]/1e6:8.2f}  {k[:120]}")
PY
