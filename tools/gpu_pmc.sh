#!/bin/bash
# PMC counter passes for one workload (separate runs per counter group; --pmc only, no trace domains).
set -o pipefail
export TMPDIR=/tmp
WL=${WL:-wolf_1080p}
mkdir -p gpurun_out/pmc
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf gpurun_out/pmc/p$i
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d gpurun_out/pmc/p$i -- python bench.py --workload $WL --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e > gpurun_out/pmc/p$i.out 2> gpurun_out/pmc/p$i.err || { tail -5 gpurun_out/pmc/p$i.err; }
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
GROUPS
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # steady state only: the first call of a geometry times candidate launch shapes (dichromat_march.hip), so keep the
    # dispatches of the last 3 steps (3 main + 3 fix-up launches), identified by their dispatch ids
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "_kernel<" in r["Kernel_Name"]})
    keep = set(ids[-6:])
    for r in rows:
        k = r["Kernel_Name"]
        if "_kernel<" in k and int(r["Dispatch_Id"]) not in keep:
            continue
        mm = re.search(r"_kernel<\w+, \d+, (true|false),", k)  # third template argument = DARK
        k = ("main" if mm.group(1) == "false" else "dark") if mm else k[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json, os
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
m = agg.get("main", {})
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    fetch_kb = sum(m["FETCH_SIZE"]) / len(m["FETCH_SIZE"]); write_kb = sum(m["WRITE_SIZE"]) / len(m["WRITE_SIZE"])
    out = {"workload": os.environ.get("WL", "wolf_1080p"), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
           "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
           "counters": {c: sum(v) / len(v) for c, v in m.items()}}
    json.dump(out, open("gpurun_out/pmc/summary_%s.json" % out["workload"], "w"), indent=1)
    print("traffic bytes/launch:", out["hbm_bytes_per_launch"])
PY
