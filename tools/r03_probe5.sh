#!/bin/bash
# Round-3 probe 5: revised one-launch decode, persistent GEMM epilogue v2, head-major attention; then suite + bench.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== decode tests + decode A/B"
timeout -k 10 300 python3 -m pytest tests/test_decode_gpu.py -x -v --timeout=60 --timeout-method=thread -m gpu > $O/tests_decode.log 2>&1 || { tail -40 $O/tests_decode.log; exit 1; }
tail -2 $O/tests_decode.log
timeout -k 10 300 python3 tools/decode_ab.py > $O/decode_ab.txt 2> $O/decode_ab.err || { tail -5 $O/decode_ab.err; exit 1; }
cat $O/decode_ab.txt
timeout -k 10 300 python3 tools/decode_real.py > $O/decode_real.txt 2> $O/decode_real.err || { tail -5 $O/decode_real.err; exit 1; }
cat $O/decode_real.txt
echo "== ops tests (tile 13, head-major)"
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py -k "experimental_forms or headmajor or attention" -x -q -m gpu > $O/tests_ops.log 2>&1 || { tail -40 $O/tests_ops.log; exit 1; }
tail -2 $O/tests_ops.log
echo "== timeline tile 13"
for spec in "12288 2304 768 13" "12288 3072 768 13 gelu"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec >> $O/timeline.txt 2>> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
done
grep -E "^M=|per workgroup|clock|persistent" $O/timeline.txt
echo "== vendor table"
timeout -k 10 400 python3 tools/gemm_vs_vendor.py --shapes qkv,fc1 > $O/gemm_vs_vendor.txt 2> $O/gemm_vs_vendor.err || { tail -5 $O/gemm_vs_vendor.err; exit 1; }
cat $O/gemm_vs_vendor.txt
echo "== bench"
timeout -k 10 700 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -15 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03i/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step","gemm_tiles_autotuned")}); print(d["roofline"]); print(d["roofline_decode"]); print(d["roofline_decode_at_scale"]); print(d["cpu_baseline"]); print(d["parity"])
PY
echo "== ViT-H (head-major attention) bench"
timeout -k 10 700 python3 bench.py --config vit_h_wholebody --no-cpu-baseline --no-parity --steps 5 --warmup 2 > $O/bench_vit_h.json 2> $O/bench_vit_h.err || { tail -15 $O/bench_vit_h.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03i/bench_vit_h.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step","attention")}); print(d["roofline"])
PY
echo "== model-level tests touched by the decode dispatch"
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -k "head_fp32 or full_size or inference" -x -q -m gpu > $O/tests_model.log 2>&1 || { tail -40 $O/tests_model.log; exit 1; }
tail -2 $O/tests_model.log
