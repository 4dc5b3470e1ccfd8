#!/bin/bash
# Build an experimental variant of the whole library for A/B runs (never shipped, never loaded by the
# package): bash tools/build_exp.sh <name> <extra hipcc flags...>  ->  probpose_pytorch_amd/lib/exp/<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/probpose_pytorch_amd/lib/exp"
cd "$ROOT/probpose_pytorch_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  "$@" pp_*.hip -o "../lib/exp/$NAME.so"
echo "built lib/exp/$NAME.so"
