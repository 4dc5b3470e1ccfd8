#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a small
markdown table (per kernel, and per kernel x grid for the GEMM shapes)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, out):
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        print("no kernel trace found under", d)
        return 1
    rows = []
    for t in traces:
        with open(t) as f:
            rows += list(csv.DictReader(f))
    per = defaultdict(lambda: [0, 0.0])
    per_grid = defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r.get("Kernel_Name", "?")
        short = name.split("(")[0][-70:]
        dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3  # us
        per[short][0] += 1
        per[short][1] += dur
        grid = (r.get("Grid_Size_X") or r.get("Grid_Size", "?"), r.get("Grid_Size_Y", ""))
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size", "?")
        per_grid[(short, grid, wg, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))][0] += 1
        per_grid[(short, grid, wg, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))][1] += dur
    total = sum(v[1] for v in per.values())
    lines = [f"# rocprofv3 kernel trace summary ({os.path.basename(d)})", "",
             f"total kernel time {total / 1e3:.3f} ms over {len(rows)} dispatches", "",
             "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    is_gemm = lambda k: any(t in k for t in ("gemm_kernel", "gemm_persist_kernel", "gemm_duo_kernel", "gemm_quad_stream_kernel"))
    gn = sum(n for k, (n, t) in per.items() if is_gemm(k))
    gt = sum(t for k, (n, t) in per.items() if is_gemm(k))
    if gn:   # the figure bench.py's roofline.avg_launch_us is compared with
        lines.append(f"| **`pp::gemm_kernel` / `gemm_persist_kernel` / `gemm_duo_kernel` / `gemm_quad_stream_kernel`, all instantiations** | {gn} | {gt:.1f} | **{gt / gn:.2f}** | {100 * gt / total:.1f} |")
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{k}` | {n} | {t:.1f} | {t / n:.2f} | {100 * t / total:.1f} |")
    lines += ["", "## per launch geometry", "", "| kernel | grid | wg | vgpr | lds | calls | avg us |", "|---|---|---|---|---|---|---|"]
    for (k, grid, wg, vg, lds), (n, t) in sorted(per_grid.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append(f"| `{k[-40:]}` | {grid[0]}x{grid[1]} | {wg} | {vg} | {lds} | {n} | {t / n:.2f} |")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
