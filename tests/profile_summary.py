"""Condense rocprofv3 outputs into the small summaries kept under profiles/.

    python tests/profile_summary.py stats  <kernel_stats.csv> <out.csv>          # per-kernel totals (as rocprofv3 --stats wrote them)
    python tests/profile_summary.py pmc    <counter_collection.csv> <out.json>   # FETCH_SIZE per launch of every kernel
    python tests/profile_summary.py counters <out.txt> <counter_collection.csv> [more.csv ...]   # any counters: per kernel, per launch averages

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


def counters(dst, *srcs):
    """per kernel (top 22 by total wave cycles): average of every collected counter per launch, plus the derived average vector-memory latency
    SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD (cycles an issued vector-memory read stays outstanding) where both were collected"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for src in srcs:
        for r in csv.DictReader(open(src)):
            k = r["Kernel_Name"].split("(")[0]
            c = agg[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    names = sorted({c for k in agg for c in agg[k]})
    order = sorted(agg, key=lambda k: -(agg[k]["SQ_WAVE_CYCLES"][1] if "SQ_WAVE_CYCLES" in agg[k] else max(v[0] for v in agg[k].values())))   # by total wave cycles
    with open(dst, "w") as f:
        f.write("# per launch averages; counters collected in separate rocprofv3 --pmc passes of the same command (eager launches)\n")
        for k in order[:22]:
            n = max(v[0] for v in agg[k].values())
            f.write(f"{k}  launches {n}\n")
            for c in names:
                if c in agg[k]:
                    f.write(f"    {c:28s} {agg[k][c][1] / agg[k][c][0]:16.1f}\n")
            a = agg[k]
            if "SQ_INST_LEVEL_VMEM" in a and "SQ_INSTS_VMEM_RD" in a and a["SQ_INSTS_VMEM_RD"][1] > 0:
                f.write(f"    {'-> cycles per vmem read':28s} {a['SQ_INST_LEVEL_VMEM'][1] / a['SQ_INST_LEVEL_VMEM'][0] / (a['SQ_INSTS_VMEM_RD'][1] / a['SQ_INSTS_VMEM_RD'][0]):16.1f}\n")
            if "SQ_WAIT_INST_ANY" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"][1] > 0:
                f.write(f"    {'-> waiting / wave cycles':28s} {a['SQ_WAIT_INST_ANY'][1] / a['SQ_WAIT_INST_ANY'][0] / (a['SQ_WAVE_CYCLES'][1] / a['SQ_WAVE_CYCLES'][0]):16.3f}\n")
            if "SQ_ACTIVE_INST_VALU" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"][1] > 0:
                f.write(f"    {'-> VALU active / wave cycles':28s} {a['SQ_ACTIVE_INST_VALU'][1] / a['SQ_ACTIVE_INST_VALU'][0] / (a['SQ_WAVE_CYCLES'][1] / a['SQ_WAVE_CYCLES'][0]):16.3f}\n")


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], *sys.argv[3:])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
