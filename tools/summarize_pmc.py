#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-B/lane
streaming stores (narrower stores are uncalibrated; reported as read)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                name = r.get("Kernel_Name", "?").split("(")[0]
                per[name][0] += 1
                per[name][1] += float(r["Counter_Value"])
    return per


def main(root, tag, out):
    fetch = collect(os.path.join(root, f"pmc_{tag}_FETCH_SIZE"), "FETCH_SIZE")
    write = collect(os.path.join(root, f"pmc_{tag}_WRITE_SIZE"), "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        res[k] = {
            "launches": max(nf, nw),
            "fetch_bytes_per_launch": (vf / nf * 1024 * 2) if nf else None,   # x2: gfx950 FETCH_SIZE correction
            "write_bytes_per_launch": (vw / nw * 1024) if nw else None,
        }
        f, w = res[k]["fetch_bytes_per_launch"], res[k]["write_bytes_per_launch"]
        res[k]["hbm_bytes_per_launch"] = (f or 0) + (w or 0)
    with open(out, "w") as fh:
        json.dump({"tag": tag, "note": "FETCH_SIZE KiB x2 (gfx950 correction), WRITE_SIZE KiB; separate --pmc passes; "
                   "bench.py --steps 2 --warmup 1 --no-graph, ViT-B bs64", "kernels": res}, fh, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k[-60:]:60s} n={v['launches']:4d} fetch/launch={v['fetch_bytes_per_launch'] or 0:12.0f} "
              f"write/launch={v['write_bytes_per_launch'] or 0:12.0f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
