import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_reg_kernel<1024, 20>" in r["Kernel_Name"]]
i = marks[-8]
f = rows[i]
s, e = int(f["Start_Timestamp"]), int(f["End_Timestamp"])
inside = [r for r in rows if s <= int(r["Start_Timestamp"]) < e and r is not f]
print("FPS1 duration %.1f us; kernels starting inside it: %d, their total time %.1f us" % ((e - s) / 1e3, len(inside), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in inside) / 1e3))
for r in inside[:12]:
    print("   +%.1f us %.1f us %s" % ((int(r["Start_Timestamp"]) - s) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80]))
