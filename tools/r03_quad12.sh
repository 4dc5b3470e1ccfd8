#!/bin/bash
# Round-3 session: the stream's fill -- K-tiles 0, 1 first and 2, 3 behind K-tile 0's arrival (lab_tl) against all four at once (lab_fillall).
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests12.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests12.log
if [ $rc -ne 0 ]; then exit $rc; fi
: > $OUT/timeline12.txt
for rep in 1 2; do
for lib in lab_tl lab_fillall; do
for spec in "12288 2304 768 19" "12288 3072 768 20 gelu"; do
  echo "== $lib" >> $OUT/timeline12.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so $spec 2>&1 | grep -v amdgpu.ids | head -2 >> $OUT/timeline12.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done; done; done
cat $OUT/timeline12.txt
