#!/bin/bash
# Round-3 session: the three DMA-issue mechanisms of the four-wave GEMM forms (lab builds lab_qb0/1/2), timelines only.
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad" > $OUT/tests2.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests2.log
if [ $rc -ge 124 ]; then exit $rc; fi
: > $OUT/timeline2.txt
for v in 0 1 2; do
for spec in "12288 2304 768 17" "12288 3072 768 16"; do
  echo "== PP_QUAD_BURST=$v" >> $OUT/timeline2.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_qb$v.so $spec 2>&1 | grep -v amdgpu.ids | head -3 >> $OUT/timeline2.txt
  rc=$?; if [ $rc -ge 124 ]; then echo "timeline $spec killed"; exit $rc; fi
done; done
cat $OUT/timeline2.txt
