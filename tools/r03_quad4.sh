#!/bin/bash
# Round-3 session: what the DMA pieces cost in the four-wave K-loop (lab builds), and the stream's epilogue share.
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline4.txt
for lib in lab_qb0 lab_nodma; do
for spec in "12288 2304 768 17" "12288 3072 768 16"; do
  echo "== $lib" >> $OUT/timeline4.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so $spec 2>&1 | grep -v amdgpu.ids | head -3 >> $OUT/timeline4.txt
  rc=$?; if [ $rc -ge 124 ]; then echo "timeline $spec killed"; exit $rc; fi
done; done
for spec in "12288 2304 768 19" "12288 3072 768 18 gelu" "12288 3072 768 18"; do
  echo "== lab_tl" >> $OUT/timeline4.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/timeline4.txt
done
cat $OUT/timeline4.txt
