"""Per-step summary of a rocprofv3 --kernel-trace CSV: takes the last K steps (delimited by the
20000-point FPS launch), reports span, busy, idle and the top kernels by time per step."""
import csv, sys, collections
path, K = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
SKIP = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # steps to drop at the end (eager post-pass)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_reg_kernel<1024, 20>" in r["Kernel_Name"] or "fps_prune_kernel<20" in r["Kernel_Name"] or "fps_pair_kernel<20" in r["Kernel_Name"]]
sel = rows[marks[-K - 1 - SKIP]:marks[-1 - SKIP]]
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
busy, cur_end = 0, 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r["Kernel_Name"][:90]]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
print(f"steps {K}: span/step {span/K/1e6:.2f} ms, busy {busy/K/1e6:.2f} ms, idle {(span-busy)/K/1e6:.2f} ms, launches/step {len(sel)/K:.0f}")
small = sum(v[0] for v in agg.values() if v[0] / v[1] < 10000)
print(f"kernels with avg < 10 us: {small/K/1e6:.2f} ms/step")
for name, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"  {t/K/1e6:7.3f} ms/step  x{n/K:6.1f}  avg {t/n/1e3:8.1f} us  {name}")

import os
if os.environ.get("DETAIL"):
    # per launch-shape breakdown of one kernel family (grid size + duration)
    sub = os.environ["DETAIL"]
    det = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        if sub in r["Kernel_Name"]:
            key = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            d = det[key]; d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
    print(f"--- {sub}: per grid shape")
    for k, (t, n) in sorted(det.items(), key=lambda kv: -kv[1][0]):
        print(f"  {t/K/1e3:8.1f} us/step  x{n/K:5.1f}  avg {t/n/1e3:7.1f} us  grid {k}")

if os.environ.get("TOPN"):
    n = int(os.environ["TOPN"])
    one = rows[marks[-2 - SKIP]:marks[-1 - SKIP]]
    t0 = int(one[0]["Start_Timestamp"])
    print(f"--- the {n} longest launches of one step (start offset us, duration us, name)")
    for r in sorted(one, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:n]:
        print(f"  @{(int(r['Start_Timestamp'])-t0)/1e3:8.1f}  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  {r['Kernel_Name'][:110]}")

if os.environ.get("LIBSEQ"):
    # every library (non-demf) launch of one step, in order, with the demf kernel in front of it for context
    one = rows[marks[-2 - SKIP]:marks[-1 - SKIP]]
    t0 = int(one[0]["Start_Timestamp"])
    print("--- library launches of one step (offset us, duration us, grid, name | previous demf kernel)")
    prev = "-"
    for r in one:
        n = r["Kernel_Name"]
        if "demf::" in n:
            prev = n.split("demf::")[1][:40]
            continue
        print("  @%8.1f %7.1f us  grid %-9s %-70s | %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")),
              n[:70], prev))
