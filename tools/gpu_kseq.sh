#!/bin/bash
# Ordered kernel sequence of ONE steady-state frame of a bench workload (one lane): name, duration -> gpurun_out/kseq_<workload>.txt
# usage: bash tools/gpu_kseq.sh <workload> [substring of the frame's first kernel; default conv_in_u8; no match: the last 60 launches]
WL=${1:-honeybee_mst_4k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kseq
AVX_BENCH_MST_LANES=1 AVX_BENCH_UV_LANES=1 AVX_MANTIS_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kseq -o k -- python bench.py --workload $WL --no-cpu-baseline --no-e2e --no-legs --steps 3 --warmup 2 > gpurun_out/kseq.log 2>&1 || { tail -5 gpurun_out/kseq.log; exit 1; }
python - > gpurun_out/kseq_$WL.txt <<PY
import csv, glob
f = glob.glob("gpurun_out/kseq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marker = "$2" or "conv_in_u8"
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# the last marks belong to bench.py's parity check (two forward passes since round 3) and follow the dominant-kernel loop: take a frame from the middle of the timed region
a, b = (marks[len(marks) // 2], marks[len(marks) // 2 + 1]) if len(marks) >= 4 else (max(0, len(rows) - 60), len(rows))
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
    prev_end = max(prev_end, e); tot += e - s
print("frame span %.1f us, kernel time %.1f us, %d launches" % ((prev_end - t0) / 1e3, tot / 1e3, b - a))
PY
rm -rf gpurun_out/kseq
tail -1 gpurun_out/kseq_$WL.txt
