#!/bin/bash
# PMC passes over the MST++ route (WL=...: any workload; kernels of the library are reported by name) (separate runs per counter group; --pmc only, no trace domains): per-kernel means.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_mst
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf gpurun_out/pmc_mst/p$i
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d gpurun_out/pmc_mst/p$i -- python bench.py --workload ${WL:-honeybee_mst_1080p} --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-e2e > gpurun_out/pmc_mst/p$i.out 2> gpurun_out/pmc_mst/p$i.err || { tail -3 gpurun_out/pmc_mst/p$i.err; }
done <<GROUPS
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS
FETCH_SIZE
WRITE_SIZE
GROUPS
python - <<'PY' | tee gpurun_out/pmc_mst/summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mst/p*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # steady state: keep the last third of the dispatches of each kernel
    byk = collections.defaultdict(list)
    for r in rows:
        byk[r["Kernel_Name"]].append(r)
    for k, rs in byk.items():
        if not any(t in k for t in ("k_mst_", "k_dwconv", "k_ew", "k_sel_pass", "k_plane_blur", "streak", "k_spectral", "k_bee", "dichromat", "k_mantis", "k_up_", "k_barcode", "k_stack")):
            continue
        ids = sorted({int(r["Dispatch_Id"]) for r in rs})
        keep = set(ids[len(ids) * 2 // 3:])
        mm = re.search(r"(k_\w+(<[^>]*>)?|dichromat_\w+)", k)
        name = mm.group(1) if mm else k[:40]
        for r in rs:
            if int(r["Dispatch_Id"]) in keep:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
# HBM traffic of ONE frame of the route: FETCH_SIZE / WRITE_SIZE (KB) summed over every dispatch between two consecutive k_map_encode
# launches (one per frame); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 tallies 128-B reads as 64 B)
tot = {}
for f in glob.glob("gpurun_out/pmc_mst/p*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
    if not rows:
        continue
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the last kernel of a frame: k_map_encode (plane schedule of the honeybee tail) or k_bee_tile<.., .., 1, ..> (its map + encode stage)
    marks = [i for i, r in enumerate(rows) if "k_map_encode" in r["Kernel_Name"] or re.search(r"k_bee_tile<\d, \d, 1, ", r["Kernel_Name"])]
    if len(marks) >= 2:
        # a steady frame: the window after the warm-up frame's mark (the LAST window also holds bench.py's own 20 extra launches of the
        # dominant kernel and the parity check's forward pass)
        lo, hi = (marks[1], marks[2]) if len(marks) >= 3 else (marks[-2], marks[-1])
        win = rows[lo + 1: hi + 1]
        tot[rows[0]["Counter_Name"]] = sum(float(r["Counter_Value"]) for r in win)
        per = collections.defaultdict(float)
        for r in win:
            mm = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            per[mm.group(1) if mm else r["Kernel_Name"][:40]] += float(r["Counter_Value"])
        print("== per-frame", rows[0]["Counter_Name"], "KB by kernel:", ", ".join(f"{k}={v:.0f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]))
if len(tot) == 2:
    import json
    b = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
    print(f"== HBM traffic per frame: FETCH_SIZE {tot['FETCH_SIZE']:.0f} KB (x2), WRITE_SIZE {tot['WRITE_SIZE']:.0f} KB -> {b/1e9:.3f} GB")
    json.dump({"FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"], "hbm_bytes_per_frame": b}, open("gpurun_out/pmc_mst/traffic.json", "w"))
for k in sorted(agg):
    d = agg[k]
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print("==", k)
    print("   " + "  ".join(f"{c}={m[c]:.4g}" for c in sorted(m)))
    if "SQ_BUSY_CYCLES" in m and "SQ_ACTIVE_INST_VALU" in m:
        print(f"   VALU active / wave-cycles-per-SIMD: {m['SQ_ACTIVE_INST_VALU'] / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
rm -rf gpurun_out/pmc_mst/p[0-9]*/  # the raw counter CSVs are hundreds of MB: only summary.txt and traffic.json travel back
