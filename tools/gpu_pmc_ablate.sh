#!/bin/bash
# LDS counters of the marching kernel's diagnostic (STAMP) instantiation with phases ablated (AVX_ABLATE bits:
# 1 decode, 2 row pass, 4 quantiser + OUT writes, 8 quantiser lookups only, 32 stores)
set -o pipefail
export TMPDIR=/tmp
WL=${WL:-cat_1080p}
for ab in ${ABS:-0 1 2 4 8 32}; do
  rm -rf gpurun_out/pmca
  AVX_STAMPS=1 AVX_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmca -- python bench.py --workload $WL --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e > gpurun_out/pmca.out 2> gpurun_out/pmca.err || { tail -5 gpurun_out/pmca.err; exit 1; }
  python - $ab <<'PY'
import csv, glob, collections, sys, re
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmca/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if re.search(r"_march_kernel<\w+, \d+, false, [^>]*, true, (true|false)>\(", k):  # main (not DARK) kernel, STAMP = true
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate", sys.argv[1], " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg.items())))
PY
done
