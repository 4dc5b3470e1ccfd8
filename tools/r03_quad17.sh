#!/bin/bash
# Round-3 session: the column permutation of the 256 x 192 form (tile 18), beside tile 20 on the ViT-H qkv shape.
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests19.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests19.log
if [ $rc -ne 0 ]; then grep -E "^E" $OUT/tests19.log | head -8; exit $rc; fi
: > $OUT/timeline19.txt
for spec in "55296 3840 1280 18" "55296 3840 1280 20" "12288 3072 768 18" "12288 3072 768 20"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -2 >> $OUT/timeline19.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
cat $OUT/timeline19.txt
