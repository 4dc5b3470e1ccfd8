#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests7.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests7.log
if [ $rc -ge 124 ]; then exit $rc; fi

: > $OUT/timeline7.txt
for spec in "12288 2304 768 19" "12288 3072 768 18 gelu" "12288 3072 768 18"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline7.txt
done
cat $OUT/timeline7.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 --tiles 3,7,13,18,19,20 > $OUT/vs_vendor7.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok\|amdgpu.ids" $OUT/vs_vendor7.txt | tail -8
