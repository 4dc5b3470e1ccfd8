#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline8.txt
for spec in "768 2304 768 19" "3072 2304 768 19" "6144 2304 768 19" "12288 2304 768 19" "24576 2304 768 19"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline8.txt
done
cat $OUT/timeline8.txt
