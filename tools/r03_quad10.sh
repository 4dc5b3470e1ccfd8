#!/bin/bash
# Round-3 session: the 128-byte store segments of the 192 x 256 stream form (tile 20).
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests10.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests10.log
if [ $rc -ne 0 ]; then exit $rc; fi
: > $OUT/timeline10.txt
for spec in "12288 3072 768 20 gelu" "12288 3072 768 20" "12288 3072 768 18" "49152 4096 1024 20 gelu"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline10.txt
done
cat $OUT/timeline10.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes fc1 --tiles 7,13,18,20 > $OUT/vs_vendor10.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok\|amdgpu.ids" $OUT/vs_vendor10.txt | tail -5
