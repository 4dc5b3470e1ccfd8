#!/bin/bash
# HBM traffic of the hot kernels from the PMC counters, each counter in its own pass
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass), kernel-trace only.
# Usage (through gpurun): bash tools/profile_pmc.sh <tag>
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TUNE=$ROOT/gpurun_out/pmc_${TAG}_tune.json
python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-parity --no-decode-scale --tune-cache "$TUNE" > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$C
  rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity --no-decode-scale --tune-cache "$TUNE" > "$OUT/bench_stdout.txt" 2>&1
  echo "pass $C done"
done
python3 "$ROOT/tools/summarize_pmc.py" "$ROOT/gpurun_out" "$TAG" "$ROOT/gpurun_out/${TAG}_pmc_traffic.json"
