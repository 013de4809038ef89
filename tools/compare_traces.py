import csv, sys, collections
def load(path, K, SKIP):
    rows = list(csv.DictReader(open(path))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "fps_reg_kernel<1024, 20>" in r["Kernel_Name"]]
    sel = rows[marks[-K - 1 - SKIP]:marks[-1 - SKIP]]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"][:80]]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
    q = collections.defaultdict(int)
    for r in sel: q[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    return {k: (v[0] / K / 1e3, v[1] / K) for k, v in agg.items()}, {k: v / K / 1e6 for k, v in q.items()}, span / K / 1e6
a, qa, sa = load(sys.argv[1], 6, 6); b, qb, sb = load(sys.argv[2], 6, 6)
print("span/step: %.2f -> %.2f ms; per-queue busy ms:" % (sa, sb), qa, qb)
rows = []
for k in set(a) | set(b):
    ta, tb = a.get(k, (0, 0))[0], b.get(k, (0, 0))[0]
    rows.append((tb - ta, ta, tb, k))
rows.sort(reverse=True)
print("kernels whose per-step time grew most (us/step):")
for d, ta, tb, k in rows[:14]: print("  %+8.1f  %8.1f -> %8.1f  %s" % (d, ta, tb, k))
print("total delta %.1f us" % sum(r[0] for r in rows))
