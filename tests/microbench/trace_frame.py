"""Last frame of a rocprofv3 kernel trace (csv) of tests/microbench/mimi_only.py: start, gap to the previous kernel's end, duration, grid, name.  python trace_frame.py trace.csv frames_traced"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // int(sys.argv[2])
fr = rows[-per:]
t0 = int(fr[0]["Start_Timestamp"]); prev_end = t0; tot = 0
for r in fr:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    print(f"{(s - t0) / 1000:8.1f} gap {(s - prev_end) / 1000:5.1f} dur {(e - s) / 1000:5.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size')):>7} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size')):>4} {r['Kernel_Name'][:70]}")
    prev_end = e; tot += e - s
print(f"frame span {(prev_end - t0) / 1000:.1f} us, kernel time {tot / 1000:.1f} us, {per} launches")
