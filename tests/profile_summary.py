"""Condense rocprofv3 outputs into the small summaries kept under profiles/.

    python tests/profile_summary.py stats  <kernel_stats.csv> <out.csv>          # per-kernel totals (as rocprofv3 --stats wrote them)
    python tests/profile_summary.py pmc    <counter_collection.csv> <out.json>   # FETCH_SIZE per launch of every kernel

FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per 128-B request of a wide streaming read
(MI355X_MICROARCH.md "HBM"): bytes = 2 x 1024 x FETCH_SIZE.
"""
import collections
import csv
import json
import sys


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(src, dst):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"unit": "bytes per launch = 2 * 1024 * FETCH_SIZE[KB] (gfx950 wide-read correction)", "source_sha": bench.source_sha(), "kernels": {}}
    mv = [0, 0.0]
    for k, (n, kb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(2 * 1024 * kb / n)}
        if "matvec_q4k_kernel" in k:
            mv[0] += n
            mv[1] += kb
    if mv[0]:
        out["matvec_q4k_kernel"] = {"launches": mv[0], "fetch_bytes_per_launch": round(2 * 1024 * mv[1] / mv[0])}
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
