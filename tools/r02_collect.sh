#!/bin/bash
# Round-2 artefact collection (through gpurun): bench line, rocprofv3 trace, PMC traffic, the other BASELINE configurations.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
timeout -k 10 300 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
bash tools/profile_bench.sh r02 > $O/r02_profile_bench.log 2>&1
bash tools/profile_pmc.sh r02 > $O/r02_profile_pmc.log 2>&1
timeout -k 10 280 python bench.py --config vit_l --no-cpu-baseline --no-parity > $O/r02_bench_vit_l_bf16.json 2> $O/e1.err
timeout -k 10 280 python bench.py --config vit_l --dtype fp8 --no-cpu-baseline --no-parity > $O/r02_bench_vit_l_fp8.json 2> $O/e2.err
timeout -k 10 280 python bench.py --config vit_h_wholebody --no-cpu-baseline --no-parity > $O/r02_bench_vit_h_wholebody.json 2> $O/e3.err
timeout -k 10 200 python bench.py --config vit_s --no-cpu-baseline --no-parity > $O/r02_bench_vit_s.json 2> $O/e4.err
timeout -k 10 200 python bench.py --dtype fp32 --no-cpu-baseline --no-parity > $O/r02_bench_vit_b_fp32.json 2> $O/e5.err
timeout -k 10 150 python tools/decode_ab.py > $O/r02_decode_ab.txt 2>&1
timeout -k 10 100 python tools/att_sweep.py > $O/r02_att_sweep.txt 2>&1
timeout -k 10 100 python tools/decode_real.py > $O/r02_decode_real.txt 2>&1
for f in $O/r02_bench.json $O/r02_bench_vit_*.json; do
  python -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['attention']['achieved_tflops'], d['decode_ms'], (d['roofline_decode_at_scale'] or {}).get('frac'))" $f
done
