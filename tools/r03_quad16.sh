#!/bin/bash
# Round-3 session: chunk stores as four turns (one real, three dropped) a few MFMAs apart.
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests18.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests18.log
if [ $rc -ne 0 ]; then grep -E "^E" $OUT/tests18.log | head -8; exit $rc; fi
: > $OUT/timeline18.txt
for spec in "12288 2304 768 19" "12288 3072 768 20" "12288 3072 768 20 gelu" "12288 2304 768 19"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline18.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
cat $OUT/timeline18.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 --tiles 6,13,19,20 > $OUT/vs_vendor18.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok\|amdgpu.ids" $OUT/vs_vendor18.txt | tail -6
