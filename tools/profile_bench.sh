#!/bin/bash
# Collect the rocprofv3 kernel trace of the bench command on the GPU box and
# summarise it.  Usage (through gpurun): bash tools/profile_bench.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# un-profiled pass first: tunes the GEMM tiles and saves the winners, so the traced run launches only the step's kernels
python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-decode-scale --tune-cache "$OUT/tune.json" "$@" > "$OUT/tune_stdout.txt" 2>&1
# --single-stream: the kernels' own durations (the aux-branch overlap of the product path stretches whichever
# kernels run concurrently); this is also how bench.py measures roofline.avg_launch_us
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-decode-scale --single-stream --tune-cache "$OUT/tune.json" "$@" > "$OUT/bench_stdout.txt" 2>&1
tail -1 "$OUT/bench_stdout.txt"
python3 "$ROOT/tools/summarize_trace.py" "$OUT" "$ROOT/gpurun_out/${TAG}_kernel_summary.md"
