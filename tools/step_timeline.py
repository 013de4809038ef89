"""Per-launch timeline of ONE training step out of a rocprofv3 kernel trace (the last complete span between
two demf_adamw launches): start offset, duration, grid, kernel.  usage: step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = [m for i, m in enumerate(marks) if i + 1 == len(marks) or marks[i + 1] != m + 1]   # last of a run
a, b = ends[-4] + 1, ends[-3] + 1      # (the last spans are the eager kernel-timer pass of bench.py)
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    g = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>2} {g:>16}  {r['Kernel_Name'][:140]}")
print(f"launches {b - a}  busy {tot / 1e3:.1f} us  span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
