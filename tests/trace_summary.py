"""Summarise a rocprofv3 kernel trace CSV per (kernel, grid): python tests/trace_summary.py <kernel_trace.csv> [frames]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
tot = collections.defaultdict(float)
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[(name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Workgroup_Size_X'])].append(d)
    tot[name] += d
print("== per kernel (ms per frame, launches per frame)")
cnt = collections.Counter(r['Kernel_Name'].split('(')[0] for r in rows)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:40s} {v / 1e6 / frames:8.3f} ms  {cnt[k] / frames:8.1f}")
print("total %.3f ms/frame, %.0f launches/frame" % (sum(tot.values()) / 1e6 / frames, len(rows) / frames))
print("== per (kernel, workgroups, threads): n/frame, median us, total ms/frame")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v.sort()
    print(f"{k[0]:32s} wg={k[1]:6d} thr={k[2]:>4s} n={len(v) / frames:7.1f} med={v[len(v) // 2] / 1e3:8.1f} min={v[0] / 1e3:7.1f} tot={sum(v) / 1e6 / frames:7.3f}")
