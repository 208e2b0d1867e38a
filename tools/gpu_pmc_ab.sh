#!/bin/bash
# VALU instruction counts and busy cycles per kernel for two builds of the library on one box (one lane): libavx_<tag>.so vs the current one
export TMPDIR=/tmp
WL=${1:-honeybee_mst_4k}; TAG=${2:-A0}
L=animal-vision_amd
cp $L/libavx.so /tmp/libavx_cur.so
for v in $TAG cur; do
  if [ $v = cur ]; then cp /tmp/libavx_cur.so $L/libavx.so; else cp $L/libavx_$v.so $L/libavx.so; fi
  rm -rf gpurun_out/pmcab
  AVX_BENCH_MST_LANES=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/pmcab -- python bench.py --workload $WL --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e --no-legs > gpurun_out/pmcab.out 2> gpurun_out/pmcab.err || { tail -3 gpurun_out/pmcab.err; }
  python - $v <<'PY'
import csv, glob, collections, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_mst_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if "ffn" in k or "tail" in k:
        print(sys.argv[1], k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
done
cp /tmp/libavx_cur.so $L/libavx.so
rm -rf gpurun_out/pmcab
