"""Per-kernel digest of ONE profiled bench run as JSON: for every kernel instantiation its time per step
(rocprofv3 --kernel-trace, the last K graph-replayed steps), launches per step and - when the two PMC
passes of the same command are given - its HBM bytes per step (FETCH_SIZE x 2 + WRITE_SIZE, see
bench.py pmc_per_launch), achieved GB/s and the fraction of the 8 TB/s roof.  bench.py reads the newest
committed profiles/r*_top_kernels.json for ``roofline.share_of_step`` and ``roofline_top5``.

usage: python tools/top_kernels.py <kernel_trace.csv> [pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv] [K=5] [SKIP=7]
"""
import collections
import csv
import json
import sys

HBM_PEAK_GBS = 8000.0
MARK = "fps_reg_kernel<1024, 20>"       # one launch per step (the 20 000 -> 2 048 FPS chain)


def load_pmc(path):
    out = {}
    with open(path) as fh:
        for row in csv.reader(fh):
            if len(row) >= 4 and row[0] != "kernel":
                out[row[0]] = (int(row[1]), float(row[2]))      # calls, mean KiB per launch
    return out


def main():
    args = sys.argv[1:]
    trace = args[0]
    pmc = [a for a in args[1:] if a.endswith(".csv")]
    nums = [int(a) for a in args[1:] if a.isdigit()]
    K = nums[0] if nums else 5
    SKIP = nums[1] if len(nums) > 1 else 7
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"] or "fps_prune_kernel<20" in r["Kernel_Name"] or "fps_pair_kernel<20" in r["Kernel_Name"]]
    if len(marks) < K + 1 + SKIP:
        SKIP = max(0, len(marks) - K - 1)
    sel = rows[marks[-K - 1 - SKIP]:marks[-1 - SKIP]]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    fetch = load_pmc(pmc[0]) if len(pmc) >= 2 else {}
    write = load_pmc(pmc[1]) if len(pmc) >= 2 else {}
    kern = []
    for name, (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        e = dict(kernel=name[:160], us_per_step=ns / K / 1e3, launches_per_step=n / K,
                 avg_launch_us=ns / n / 1e3,
                 # coordinate-only pre-pass of the NEXT batch: runs on the side stream underneath the step
                 side_stream=any(t in name for t in ("fps_", "ball_query", "bq_grid", "invert_index",
                                                     "three_nn")))
        key = name[:110]
        if key in fetch and key in write:
            per_launch = (2.0 * fetch[key][1] + write[key][1]) * 1024.0
            e["hbm_bytes_per_step"] = per_launch * n / K
            e["gbps"] = per_launch * n / (ns * 1e-9) / 1e9 if ns else None
            e["frac_of_hbm_peak"] = e["gbps"] / HBM_PEAK_GBS if e["gbps"] is not None else None
        kern.append(e)
    total_us = sum(e["us_per_step"] for e in kern)
    out = dict(source=trace.split("/")[-1], steps=K, span_us_per_step=span / K / 1e3,
               kernel_us_per_step=total_us, launches_per_step=len(sel) / K,
               library_launches_per_step=sum(e["launches_per_step"] for e in kern
                                             if "demf::" not in e["kernel"]),
               small_kernel_us_per_step=sum(e["us_per_step"] for e in kern if e["avg_launch_us"] < 10.0),
               kernels=kern[:60])
    tot = fetch.get("__TOTAL_PER_STEP__"), write.get("__TOTAL_PER_STEP__")
    if tot[0] and tot[1]:
        out["hbm_bytes_per_step"] = (2.0 * tot[0][1] + tot[1][1]) * 1024.0
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
