#!/bin/bash
# Round-3 session: the 192 x 288 stream form with whole-row stores through an LDS staging buffer (ROWS).
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests14.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests14.log
if [ $rc -ne 0 ]; then grep -E "^E" $OUT/tests14.log | head -8; exit $rc; fi
: > $OUT/timeline14.txt
for spec in "12288 2304 768 19" "12288 2304 768 19" "6144 2304 768 19"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline14.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
cat $OUT/timeline14.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes qkv --tiles 6,13,19 > $OUT/vs_vendor14.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok\|amdgpu.ids" $OUT/vs_vendor14.txt | tail -4
