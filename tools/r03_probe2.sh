#!/bin/bash
# Round-3 probe 2: what holds the clock down in the GEMM K-loop (fabric traffic vs operand entropy), vendor kernels per shape.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
for rep in 1 2; do
for spec in "12288 768 768 3" "12288 768 768 3 batch8" "12288 768 768 3 batch8 l2res" "12288 768 768 3 samerows" "12288 768 768 3 lda0 ldw0" \
            "12288 2304 768 3" "12288 2304 768 3 samerows" "12288 2304 768 6" "12288 2304 768 6 samerows" "12288 2304 768 3 batch8" "12288 2304 768 3 batch8 l2res"; do
  timeout -k 10 120 python3 tools/gemm_timeline.py --lib lab_tl.so $spec 2>> $O/power.err | grep -E "^M=|clock" >> $O/power.txt || { tail -5 $O/power.err; exit 1; }
done
done
cat $O/power.txt
echo "== vendor kernels per shape"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/vt -- python3 tools/gemm_vs_vendor.py --rounds 1 --inner 3 --tiles 6 > $O/vt.log 2>&1 || { tail -5 $O/vt.log; exit 1; }
find $O/vt -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/vendor_kernel_trace.csv
rm -rf $O/vt
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r03b/vendor_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not (n.startswith("Cijk") or n.startswith("Custom_Cijk")): continue
    key = (n, r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size",""), r.get("VGPR_Count",""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    print(f"{len(v):3d} calls  avg {sum(v)/len(v)/1e3:8.1f} us  grid {k[1]:>8s} wg {k[2]:>4s} lds {k[3]:>6s} vgpr {k[4]:>4s}  {k[0][:170]}")
PY
