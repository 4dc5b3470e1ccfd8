#!/bin/bash
# rocprofv3 kernel trace of the front-end micro-benchmark.  Usage (through gpurun): bash tools/profile_frontend.sh <tag>
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_frontend_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
  python3 "$ROOT/tools/frontend_bench.py" > "$OUT/stdout.txt" 2>&1
tail -1 "$OUT/stdout.txt"
python3 "$ROOT/tools/summarize_trace.py" "$OUT" "$ROOT/gpurun_out/${TAG}_frontend_kernel_summary.md" | head -12
