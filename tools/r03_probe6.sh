#!/bin/bash
# Round-3 probe 6: decode as shipped (all-pixel for small batches; wave + self-resetting list kernel above), full suite.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python3 -m pytest tests/test_decode_gpu.py -x -q --timeout=60 --timeout-method=thread -m gpu > $O/tests_decode.log 2>&1 || { tail -30 $O/tests_decode.log; exit 1; }
tail -1 $O/tests_decode.log
timeout -k 10 300 python3 tools/decode_ab.py > $O/decode_ab.txt 2> $O/decode_ab.err || { tail -5 $O/decode_ab.err; exit 1; }
cat $O/decode_ab.txt
timeout -k 10 300 python3 tools/decode_real.py vit_b 64 > $O/decode_real.txt 2> $O/decode_real.err || { tail -5 $O/decode_real.err; exit 1; }
timeout -k 10 300 python3 tools/decode_real.py vit_b 256 >> $O/decode_real.txt 2>> $O/decode_real.err || { tail -5 $O/decode_real.err; exit 1; }
cat $O/decode_real.txt
echo "== bench"
timeout -k 10 700 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -15 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03j/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step")}); print(d["roofline_decode"]); print(d["roofline_decode_at_scale"])
PY
timeout -k 10 700 python3 bench.py --config vit_h_wholebody --no-cpu-baseline --no-parity --steps 5 --warmup 2 > $O/bench_vit_h.json 2> $O/bench_vit_h.err || { tail -15 $O/bench_vit_h.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03j/bench_vit_h.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms","kernel_ms_per_step","attention")}); print(d["roofline_decode_at_scale"])
PY
echo "== full GPU suite"
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --deselect tests/test_decode_gpu.py > $O/tests_rest.log 2>&1 || { tail -40 $O/tests_rest.log; exit 1; }
tail -3 $O/tests_rest.log
