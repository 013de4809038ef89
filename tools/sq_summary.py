"""Per-kernel means of the SQ counters collected by tools/sq_counters.sh (one row per kernel, one column per counter,
mean per launch) for the step's longest kernels, plus the ratios the MI355X guide reads them by: parked / issue-stalled /
issuing shares of the wave cycles, MFMA-busy share, LDS bank-conflict share."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
KEEP = ("mlp_bwd_fused_kernel", "mlp_bwd_pool_kernel", "mlp_fwd_pc_kernel", "mlp_fwd_res_kernel", "mlp_dw_group_kernel",
        "group_first_bwd_k", "fps_pair_kernel")
rows = [(k, v) for k, v in agg.items() if any(t in k for t in KEEP)]
rows.sort(key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])))
cnt = sorted({c for _, v in rows for c in v})
print("counters (mean per launch): " + ", ".join(cnt))
for k, v in rows:
    m = {c: sum(v[c]) / len(v[c]) for c in v}
    name = k.replace("void demf::", "").split("(")[0][:64]
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    print("\n%s  (launches seen: %d)" % (name, max(len(x) for x in v.values())))
    print("  " + "  ".join("%s=%.3g" % (c, m[c]) for c in cnt if c in m))
    if wc:
        print("  of the wave cycles: parked (s_waitcnt / barrier) %.2f, issue-stalled %.2f (of which LDS issue %.2f), issuing %.2f"
              % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc,
                 m.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print("  LDS: bank-conflict cycles / active cycles %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]))
    if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("  MFMA-busy cycles / SQ busy cycles %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]))
