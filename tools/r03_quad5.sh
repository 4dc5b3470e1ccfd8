#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests5.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests5.log
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/r03_quad4.sh > /dev/null 2>&1; cat $OUT/timeline4.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 --tiles 3,6,7,13,16,17,18,19 > $OUT/vs_vendor5.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok\|amdgpu.ids" $OUT/vs_vendor5.txt | tail -12
