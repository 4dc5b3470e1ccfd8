#!/bin/bash
# Build the diagnostic (never shipped) GEMM library with in-kernel phase stamps.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/probpose_pytorch_amd/lib/diag"
cd "$ROOT/probpose_pytorch_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -DPP_GEMM_STAMPS pp_gemm.hip pp_capi.hip -o ../lib/diag/libpp_gemm_stamps.so
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -DPP_ATT_STAMPS pp_attention.hip pp_ops.hip pp_capi.hip \
  -o ../lib/diag/libpp_att_stamps.so
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DPP_DEC_STAMPS pp_decode.hip pp_capi.hip \
  -o ../lib/diag/libpp_dec_stamps.so
