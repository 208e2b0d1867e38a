#!/bin/bash
# Error of the float16 forward pass against the reference's float32 outputs (tests/test_mstpp.py's goldens) for builds of the library: animal-vision_amd/libavx_<tag>.so and the current one
L=animal-vision_amd
cp $L/libavx.so /tmp/libavx_cur.so
for v in "$@" cur; do
  if [ $v = cur ]; then cp /tmp/libavx_cur.so $L/libavx.so; else cp $L/libavx_$v.so $L/libavx.so; fi
  echo "== $v"
  timeout -k 10 300 python -m pytest tests/test_mstpp.py -m gpu -q -s -k "large_frames_vs_reference or forward_on_gpu" 2>&1 | grep -E "part|attention|passed|failed"
done
cp /tmp/libavx_cur.so $L/libavx.so
