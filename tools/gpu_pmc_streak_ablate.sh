#!/bin/bash
# LDS counters of the streak kernel with phases ablated (AVX_DIAG=1 + AVX_ABLATE bits: 1 decode + across-channel pass, 2 first along-row
# pass, 4 second along-row pass, 8 quantiser); outputs are wrong, only the counters matter
set -o pipefail
export TMPDIR=/tmp
for ab in ${ABS:-0 1 2 4 8}; do
  rm -rf gpurun_out/pmcs
  AVX_DIAG=1 AVX_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcs -- python bench.py --workload ${WL:-sheep_1080p} --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e > gpurun_out/pmcs.out 2> gpurun_out/pmcs.err || { tail -5 gpurun_out/pmcs.err; exit 1; }
  python - $ab <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "streak_kernel<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate", sys.argv[1], " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg.items())))
PY
done
