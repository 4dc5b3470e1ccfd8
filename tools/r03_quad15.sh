#!/bin/bash
# Round-3 session: over how many K-tile iterations of the next tile the chunk stores are spread (NSI = 6 / 12 / 18).
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline16.txt
for rep in 1 2; do
for lib in lab_tl lab_nt lab_sc; do
for spec in "12288 2304 768 19" "12288 3072 768 20"; do
  echo "== $lib" >> $OUT/timeline16.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so $spec 2>&1 | grep -v amdgpu.ids | sed -n '1,2p;4p' >> $OUT/timeline16.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done; done; done
cat $OUT/timeline16.txt
