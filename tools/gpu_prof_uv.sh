#!/bin/bash
# rocprofv3 kernel trace of one UV species workload: per-kernel time breakdown of a species plan.
set -o pipefail
WL=${1:-reindeer_1080p}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$WL
rm -rf $OUT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $OUT.log 2>&1 || { tail -20 $OUT.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>5}  avg_us {float(r["AverageNs"])/1e3:9.1f}  {r["Name"][:90]}')
PY
tail -2 $OUT.log
