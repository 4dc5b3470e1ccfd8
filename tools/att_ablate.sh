#!/bin/bash
# Diagnostic: builds the attention kernel with one ingredient removed at a time (lib/diag, never shipped).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/probpose_pytorch_amd/lib/diag"
cd "$ROOT/probpose_pytorch_amd/csrc"
for a in 0 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -DPP_ATT_ABL=$a pp_attention.hip pp_ops.hip pp_capi.hip \
    -o ../lib/diag/libpp_att_abl$a.so 2>&1 | grep -v warning || true
done
