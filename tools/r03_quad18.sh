#!/bin/bash
# Round-3 session: chunk stores by turns -- each wave executes the store statement of its turn, the other three skipped by a branch (lab_turns) against the shipped form (lab_tl).
OUT=gpurun_out/r03q; mkdir -p $OUT
: > $OUT/timeline20.txt
for rep in 1 2; do
for lib in lab_tl lab_turns; do
for spec in "12288 2304 768 19" "12288 3072 768 20"; do
  echo "== $lib" >> $OUT/timeline20.txt
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib $lib.so $spec 2>&1 | grep -v amdgpu.ids | sed -n '1,2p;4p' >> $OUT/timeline20.txt
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done; done; done
grep -E "^==|lifetime|persistent" $OUT/timeline20.txt
