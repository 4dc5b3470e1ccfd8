#!/bin/bash
# Round-3 artefact collection (through gpurun): bench line, rocprofv3 trace, PMC traffic, per-shape vendor table with PMC
# bytes, small kernels (trace + PMC), the other BASELINE configurations.  Everything lands in gpurun_out/r03/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
step() { echo "== $1"; }
step "decode tests (work list: quick, bounded)"
timeout -k 10 300 python3 -m pytest tests/test_decode_gpu.py -x -q --timeout=60 --timeout-method=thread -m gpu > $O/tests_decode.log 2>&1 || { tail -30 $O/tests_decode.log; exit 1; }
tail -1 $O/tests_decode.log
step "bench (the headline line)"
timeout -k 10 600 python3 bench.py > $O/r03_bench.json 2> $O/r03_bench.err || { tail -5 $O/r03_bench.err; exit 1; }
step "kernel trace"
bash tools/profile_bench.sh r03 > $O/profile_bench.log 2>&1 || { tail -5 $O/profile_bench.log; exit 1; }
cp gpurun_out/r03_kernel_summary.md $O/ 2>/dev/null; find gpurun_out/prof_r03 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_rocprofv3_kernel_stats.csv
step "PMC traffic of the step"
bash tools/profile_pmc.sh r03 > $O/profile_pmc.log 2>&1 || { tail -5 $O/profile_pmc.log; exit 1; }
cp gpurun_out/r03_pmc_traffic.json $O/ 2>/dev/null
step "vendor yardstick (all shapes)"
timeout -k 10 500 python3 tools/gemm_vs_vendor.py > $O/r03_gemm_vs_vendor.txt 2> $O/gemm_vs_vendor.err || { tail -5 $O/gemm_vs_vendor.err; exit 1; }
cat $O/r03_gemm_vs_vendor.txt
step "PMC bytes per GEMM shape (tile the tuner picks)"
bash tools/pmc_traffic_shapes.sh > $O/r03_pmc_gemm_shapes.txt 2> $O/pmc_shapes.err || { tail -5 $O/pmc_shapes.err; exit 1; }
cat $O/r03_pmc_gemm_shapes.txt
step "small kernels: times, then PMC"
timeout -k 10 300 python3 tools/small_kernels_bench.py > $O/r03_small_kernels.txt 2> $O/small.err || { tail -5 $O/small.err; exit 1; }
cat $O/r03_small_kernels.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/small_trace -- python3 $ROOT/tools/small_kernels_bench.py --iters 5 > $ROOT/$O/small_trace.log 2>&1 ) || { tail -5 $O/small_trace.log; exit 1; }
find $O/small_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_small_kernels_stats.csv; rm -rf $O/small_trace
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$O/small_pmc_$C -- python3 $ROOT/tools/small_kernels_bench.py --once > $ROOT/$O/small_pmc_$C.log 2>&1 ) || { tail -5 $O/small_pmc_$C.log; exit 1; }
done
python3 - <<'PY' > gpurun_out/r03/r03_small_kernels_pmc.txt
import csv, glob, collections
root = "gpurun_out/r03"
res = collections.OrderedDict()
for c, mult in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    for f in glob.glob(f"{root}/small_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            if not any(k in n for k in ("sparsemax", "dark_decode", "heatmap_argmax", "pck_counts")) or r["Counter_Name"] != c:
                continue
            res.setdefault(n, {}).setdefault(c, []).append(float(r["Counter_Value"]) * mult)
print("# HBM bytes per launch from rocprofv3 --pmc (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), launches in the order of tools/small_kernels_bench.py --once")
for n, d in res.items():
    print(n[-70:], " fetch MB:", [round(v / 1e6, 2) for v in d.get("FETCH_SIZE", [])], " write MB:", [round(v / 1e6, 2) for v in d.get("WRITE_SIZE", [])])
PY
cat $O/r03_small_kernels_pmc.txt; rm -rf $O/small_pmc_FETCH_SIZE $O/small_pmc_WRITE_SIZE
step "decode tables"
timeout -k 10 200 python3 tools/decode_ab.py > $O/r03_decode_ab.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/decode_real.py vit_b 64 > $O/r03_decode_model_heatmaps.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/decode_real.py vit_b 256 >> $O/r03_decode_model_heatmaps.txt 2>&1 || exit 1
cat $O/r03_decode_ab.txt $O/r03_decode_model_heatmaps.txt
step "other BASELINE configurations"
timeout -k 10 400 python3 bench.py --config vit_l --no-cpu-baseline > $O/r03_bench_vit_l_bf16.json 2> $O/e1.err || { tail -3 $O/e1.err; exit 1; }
timeout -k 10 400 python3 bench.py --config vit_l --dtype fp8 --no-cpu-baseline > $O/r03_bench_vit_l_fp8.json 2> $O/e2.err || { tail -3 $O/e2.err; exit 1; }
timeout -k 10 400 python3 bench.py --config vit_h_wholebody --no-cpu-baseline --no-parity > $O/r03_bench_vit_h_wholebody.json 2> $O/e3.err || { tail -3 $O/e3.err; exit 1; }
timeout -k 10 300 python3 bench.py --config vit_s --no-cpu-baseline --no-parity > $O/r03_bench_vit_s.json 2> $O/e4.err || { tail -3 $O/e4.err; exit 1; }
timeout -k 10 300 python3 bench.py --dtype fp32 --no-cpu-baseline --no-parity > $O/r03_bench_vit_b_fp32.json 2> $O/e5.err || { tail -3 $O/e5.err; exit 1; }
for f in $O/r03_bench.json $O/r03_bench_vit_*.json; do
  python3 -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['attention']['achieved_tflops'], d['decode_ms'], (d['roofline_decode_at_scale'] or {}).get('frac'))" $f
done
