#!/bin/bash
# VALU / LDS / wait counters per kernel of one bench workload (one lane) -> stdout; usage: bash tools/gpu_pmc_wl.sh <workload>
export TMPDIR=/tmp
WL=${1:-honeybee_4k}
rm -rf gpurun_out/pmcw
AVX_BENCH_MST_LANES=1 AVX_BENCH_UV_LANES=1 AVX_MANTIS_LANES=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcw -- python bench.py --workload $WL --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e --no-legs > gpurun_out/pmcw.out 2> gpurun_out/pmcw.err || { tail -3 gpurun_out/pmcw.err; }
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if gui <= 0: continue
    print(f"{k[:40]:40s} n={len(d['GRBM_GUI_ACTIVE']):4d} cycles {gui:9.3g} VALU insts {m.get('SQ_INSTS_VALU',0):9.3g} busy {m.get('SQ_INSTS_VALU',0)*4/1024/gui:5.2f} LDS busy {m.get('SQ_LDS_IDX_ACTIVE',0)/256/gui:5.2f} (conflict {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):4.2f}) wait {m.get('SQ_WAIT_INST_ANY',0)/max(1,m.get('SQ_WAVE_CYCLES',1)):4.2f}")
PY
rm -rf gpurun_out/pmcw
