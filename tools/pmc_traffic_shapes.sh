#!/bin/bash
# HBM-side traffic (FETCH_SIZE x2, WRITE_SIZE; separate passes) of the four ViT-B bs64 GEMM shapes, one shape per run.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
# (name M N K tile [resid]): the tiles the round-3 tuner picks, plus the runner-up of the wide-N shapes
for SHAPE in "qkv_t19 12288 2304 768 19" "qkv_t13 12288 2304 768 13" "proj_t6 12288 768 768 6 resid" "fc1_t20 12288 3072 768 20" "fc1_t18 12288 3072 768 18" "fc1_t7 12288 3072 768 7" "fc2_t6 12288 768 3072 6 resid"; do
  set -- $SHAPE
  NAME=$1; shift
  M=$1; N=$2; K=$3; T=$4; R=${5:-}
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/pmcs_${NAME}_$C; rm -rf "$OUT"; mkdir -p "$OUT"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/gemm_one.py" $M $N $K $T 6 $R > "$OUT/stdout.txt" 2>&1
  done
  python3 - "$ROOT/gpurun_out" "$NAME" $M $N $K "$R" <<'PY'
import csv, glob, sys
root, name, M, N, K, R = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
def avg(c):
    v = []
    for f in glob.glob(f"{root}/pmcs_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    v = v[1:] or v          # drop the cold first launch
    return sum(v) / len(v) * 1024
fetch, write = avg("FETCH_SIZE") * 2, avg("WRITE_SIZE")
alg_r = M * K * 2 + N * K * 2 + (M * N * 4 if R else 0)
alg_w = M * N * (4 if R else 2)
print(f"{name:7s} fetched {fetch / 1e6:7.1f} MB (algorithmic {alg_r / 1e6:6.1f})   written {write / 1e6:6.1f} MB (algorithmic {alg_w / 1e6:6.1f})")
PY
done
