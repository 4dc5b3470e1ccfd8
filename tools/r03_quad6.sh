#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline6.txt
for lib in lab_bare lab_nopiece lab_nopiece2; do
for spec in "12288 2304 768 17"; do
  echo "== $lib" >> $OUT/timeline6.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so $spec 2>&1 | grep -v amdgpu.ids | head -3 >> $OUT/timeline6.txt
  rc=$?; if [ $rc -ge 124 ]; then echo "timeline $spec killed"; exit $rc; fi
done; done
cat $OUT/timeline6.txt
