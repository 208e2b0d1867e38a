#!/bin/bash
# LDS counters of one workload's kernels, noise vs structured content (table lookups conflict on noise)
set -o pipefail
export TMPDIR=/tmp
WL=${WL:-sheep_1080p}
for fr in noise structured; do
  rm -rf gpurun_out/pmcl
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcl -- python bench.py --workload $WL --frames $fr --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e > gpurun_out/pmcl.out 2> gpurun_out/pmcl.err || { tail -5 gpurun_out/pmcl.err; exit 1; }
  python - $fr <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "rocclr" in k: continue
    print(sys.argv[1], k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
done
