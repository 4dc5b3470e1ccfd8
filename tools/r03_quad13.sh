#!/bin/bash
# Round-3 session: what a dropped (out-of-range) chunk store costs: every store out of range (lab_oob) against the shipped form (lab_tl).
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline13.txt
for lib in lab_tl lab_oob lab_tl lab_oob; do
  echo "== $lib" >> $OUT/timeline13.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so 12288 2304 768 19 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline13.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
cat $OUT/timeline13.txt
