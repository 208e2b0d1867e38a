#!/bin/bash
# Same-box A/B of builds of the library: animal-vision_amd/libavx_<tag>.so (built from other trees) against the current libavx.so (tag "cur"),
# alternating, on one bench workload.  usage: bash tools/gpu_ab_lib.sh <workload> <rounds> <steps> <tag> [<tag> ...]
WL=${1:-honeybee_mst_4k}; R=${2:-3}; S=${3:-12}; shift 3
L=animal-vision_amd
cp $L/libavx.so /tmp/libavx_cur.so
for r in $(seq $R); do
  for v in "$@" cur; do
    if [ $v = cur ]; then cp /tmp/libavx_cur.so $L/libavx.so; else cp $L/libavx_$v.so $L/libavx.so; fi
    timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-e2e --no-legs --steps $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], d['parity_checked'])"
  done
done
cp /tmp/libavx_cur.so $L/libavx.so
