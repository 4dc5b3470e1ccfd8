#!/bin/bash
# Round-3 session for the four-wave GEMM forms (tiles 15 - 17): correctness first, then timelines and the vendor yardstick.
# A step that is killed at its limit ends the session (no further GPU step after a hang).
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "quad or headmajor" > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
: > $OUT/timeline.txt
for spec in "12288 2304 768 17" "12288 2304 768 15" "12288 3072 768 16 gelu" "12288 3072 768 16" "12288 2304 768 3"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec >> $OUT/timeline.txt 2>&1
  rc=$?; if [ $rc -ge 124 ]; then echo "timeline $spec killed"; exit $rc; fi
done
cat $OUT/timeline.txt
timeout -k 10 300 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 --tiles 3,6,7,13,15,16,17 > $OUT/vs_vendor.txt 2>&1
rc=$?; echo "vendor rc=$rc"; grep -v "^ok" $OUT/vs_vendor.txt | tail -12
exit $rc
