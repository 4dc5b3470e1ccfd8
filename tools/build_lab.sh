#!/bin/bash
# Fast experiment build of the GEMM only (bf16 tiles 3 and 10, with in-kernel stamps):
#   bash tools/build_lab.sh <name> [extra hipcc flags]  ->  probpose_pytorch_amd/lib/exp/<name>.so
# Never shipped, never loaded by the package; tools/gemm_stamps.py / gemm_sweep.py take it with --lib.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/probpose_pytorch_amd/lib/exp"
cd "$ROOT/probpose_pytorch_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -DPP_GEMM_LAB "$@" pp_gemm.hip pp_gemm_quad.hip pp_capi.hip -o "../lib/exp/$NAME.so"
echo "built lib/exp/$NAME.so"
