#!/bin/bash
# Same-box A/B of library builds on several bench workloads: animal-vision_amd/libavx_<tag>.so against the current libavx.so ("cur"), alternating.
# usage: bash tools/gpu_ab_wl.sh "<wl> <wl> ..." <rounds> <tag> [<tag> ...]
WLS=$1; R=${2:-2}; shift 2
L=animal-vision_amd
cp $L/libavx.so /tmp/libavx_cur.so
for r in $(seq $R); do
  for v in "$@" cur; do
    if [ $v = cur ]; then cp /tmp/libavx_cur.so $L/libavx.so; else cp $L/libavx_$v.so $L/libavx.so; fi
    for wl in $WLS; do
      timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'], d['parity_checked'])"
    done
  done
done
cp /tmp/libavx_cur.so $L/libavx.so
