#!/bin/bash
# Round-3 first probe: vendor yardstick, in-kernel clock / what-if-L2-resident operands, baseline bench line.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
echo "== vendor kernel names"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vendor_trace -- python3 tools/gemm_vs_vendor.py --rounds 1 --inner 3 --tiles 6 --shapes qkv,proj,fc1,fc2,aux_conv0,deconv_parity > $O/vendor_trace.log 2>&1 || { tail -5 $O/vendor_trace.log; exit 1; }
find $O/vendor_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vendor_kernel_stats.csv
cut -c1-200 $O/vendor_kernel_stats.csv | head -40
rm -rf $O/vendor_trace
echo "== timeline / clock"
for t in 6 3; do
  for spec in "12288 2304 768 $t" "12288 2304 768 $t lda0 ldw0" "12288 768 768 $t resid" "12288 3072 768 $t gelu" "12288 3072 768 $t gelu lda0 ldw0" "12288 768 3072 $t resid" "12288 768 3072 $t resid lda0 ldw0"; do
    timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec >> $O/timeline.txt 2>> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
  done
done
cat $O/timeline.txt
echo "== bench"
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
echo "== isolate the faulting yardstick shape (M 49152, N 256, K 1024, batch 4): vendor alone, then pp tile by tile"
timeout -k 10 120 python3 tools/gemm_vs_vendor.py --shapes deconv1_parity --tiles 99 --rounds 1 --inner 2 > $O/iso_vendor.txt 2> $O/iso_vendor.err || { echo "vendor-only run failed"; tail -3 $O/iso_vendor.err; exit 1; }
for t in 2 3 4 5 6 7 9 10; do
  timeout -k 10 120 python3 tools/gemm_vs_vendor.py --shapes deconv1_parity --tiles $t --no-vendor --rounds 1 --inner 2 > $O/iso_t$t.txt 2> $O/iso_t$t.err || { echo "tile $t failed"; tail -3 $O/iso_t$t.err; exit 1; }
  echo "tile $t ok"
done
