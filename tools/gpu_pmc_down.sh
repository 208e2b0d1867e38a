#!/bin/bash
# PMC passes over tools/experiments/time_down.py (the encoder's strided conv alone): LDS, MFMA and wait counters per kernel -> gpurun_out/pmc_down.txt
export TMPDIR=/tmp
: > gpurun_out/pmc_down.txt
i=0
for group in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcd
  timeout -k 10 200 rocprofv3 --pmc $group --output-format csv -d gpurun_out/pmcd -- python tools/experiments/time_down.py > gpurun_out/pmcd.out 2> gpurun_out/pmcd.err || { tail -5 gpurun_out/pmcd.err; exit 1; }
  python - >> gpurun_out/pmc_down.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "down4x4" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][22:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
done
rm -rf gpurun_out/pmcd
cat gpurun_out/pmc_down.txt
