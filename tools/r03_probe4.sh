#!/bin/bash
# Round-3 probe 4: full GPU suite after the decode / metrics / head / persistent-GEMM changes, then measurements.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== decode + metrics + head-variant + tile 13 tests first (new code)"
timeout -k 10 900 python3 -m pytest tests/test_decode_gpu.py tests/test_metrics.py tests/test_ops_gpu.py -k "not test_gemm_plain_bias_tails" -x -q -m gpu > $O/tests_new.log 2>&1 || { tail -40 $O/tests_new.log; exit 1; }
tail -3 $O/tests_new.log
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -k "head_variants or captured_graph or head_fp32" -x -q -m gpu > $O/tests_new2.log 2>&1 || { tail -40 $O/tests_new2.log; exit 1; }
tail -3 $O/tests_new2.log
echo "== timeline tile 13"
for spec in "12288 2304 768 13" "12288 3072 768 13 gelu"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec >> $O/timeline.txt 2>> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
done
cat $O/timeline.txt
echo "== vendor table"
timeout -k 10 400 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 > $O/gemm_vs_vendor.txt 2> $O/gemm_vs_vendor.err || { tail -5 $O/gemm_vs_vendor.err; exit 1; }
cat $O/gemm_vs_vendor.txt
echo "== decode A/B"
timeout -k 10 300 python3 tools/decode_ab.py > $O/decode_ab.txt 2> $O/decode_ab.err || { tail -5 $O/decode_ab.err; exit 1; }
cat $O/decode_ab.txt
timeout -k 10 300 python3 tools/decode_real.py > $O/decode_real.txt 2> $O/decode_real.err || { tail -5 $O/decode_real.err; exit 1; }
cat $O/decode_real.txt
echo "== bench"
timeout -k 10 700 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -15 $O/bench.err; exit 1; }
tail -8 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03d/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step","gemm_tiles_autotuned")}); print(d["roofline"]); print(d["roofline_decode"]); print(d["roofline_decode_at_scale"]); print(d["cpu_baseline"])
PY
echo "== 2-rank rehearsal through the self-launcher (gloo, one GPU)"
PP_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-decode-scale > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err || { tail -15 $O/bench_rehearsal2.err; exit 1; }
cut -c1-420 $O/bench_rehearsal2.json
echo "== rest of the GPU suite"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_decode_gpu.py --deselect tests/test_metrics.py > $O/tests_rest.log 2>&1 || { tail -40 $O/tests_rest.log; exit 1; }
tail -3 $O/tests_rest.log
