#!/bin/bash
# PMC counter passes over one GEMM shape.  Usage: bash tools/pmc_gemm.sh M N K tile
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"; do
  OUT=$ROOT/gpurun_out/pmcg_$i; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/gemm_one.py" "$@" 5 > "$OUT/stdout.txt" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/stdout.txt"; }
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} avg/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
  i=$((i+1))
done
