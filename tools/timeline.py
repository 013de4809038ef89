import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
cands = [r for r in rows if pat in r["Kernel_Name"]]
c = sorted(cands, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))[-1]  # the slowest instance
s, e = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
print("slowest %s: %.1f us, queue %s, grid %s" % (pat, (e - s) / 1e3, c["Queue_Id"], c.get("Grid_Size_X", c.get("Grid_Size"))))
for r in rows:
    rs, re = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if re > s - 50000 and rs < e + 50000 and (re - rs > 20000 or r is c):
        print("  q%s  start %+9.1f us  dur %8.1f us  %s" % (r["Queue_Id"], (rs - s) / 1e3, (re - rs) / 1e3, r["Kernel_Name"][:70]))
