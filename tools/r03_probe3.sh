#!/bin/bash
# Round-3 probe 3: the rebuilt persistent GEMM (tile 13): correctness, timeline, vendor table; metric / graph tests.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
echo "== tests (tile 13/14, metrics, graph survival, parallel)"
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -k "experimental_forms" tests/test_metrics.py tests/test_model_gpu.py::test_captured_graph_survives_many_other_batch_sizes -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
echo "== timeline tile 13 vs 3"
for spec in "12288 2304 768 13" "12288 2304 768 3" "12288 3072 768 13 gelu" "12288 3072 768 3 gelu"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec >> $O/timeline.txt 2>> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
done
cat $O/timeline.txt
echo "== vendor table"
timeout -k 10 400 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1,proj,fc2 > $O/gemm_vs_vendor.txt 2> $O/gemm_vs_vendor.err || { tail -5 $O/gemm_vs_vendor.err; exit 1; }
cat $O/gemm_vs_vendor.txt
echo "== bench"
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step","gemm_tiles_autotuned")}); print(d["roofline"]); print(d["cpu_baseline"])
PY
echo "== 2-rank rehearsal through the self-launcher (gloo, one GPU)"
PP_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-decode-scale > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err || { tail -5 $O/bench_rehearsal2.err; exit 1; }
cut -c1-400 $O/bench_rehearsal2.json
