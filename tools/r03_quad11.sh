#!/bin/bash
# Round-3 session: does a DMA piece cost less when it reads 8 whole lines instead of 16 half lines?  (per-launch lab form)
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline11.txt
for lib in lab_tl lab_contig; do
  echo "== $lib" >> $OUT/timeline11.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so 12288 2304 768 17 2>&1 | grep -v amdgpu.ids | head -3 >> $OUT/timeline11.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
cat $OUT/timeline11.txt
