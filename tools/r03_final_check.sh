#!/bin/bash
# What the driver runs at round end, rehearsed: smoke(), the full GPU suite, the default bench line, the 2-rank rehearsal.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1 || { tail -40 $O/tests_gpu.log; exit 1; }
tail -2 $O/tests_gpu.log
timeout -k 10 700 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -15 $O/bench.err; exit 1; }
cut -c1-900 $O/bench.json
PP_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-decode-scale > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err || { tail -15 $O/bench_rehearsal2.err; exit 1; }
cut -c1-420 $O/bench_rehearsal2.json
