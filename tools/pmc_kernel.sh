# Pipe / wait counters of the kernels matching $KPAT in one command (separate --pmc passes, no tracing):
#   KPAT=mlp_bwd_fused bash tools/pmc_kernel.sh python tools/bwd_fused_micro.py SA1.L3
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pm; (cd $R && rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o p -- "$@" > /dev/null 2>&1)
  f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
  KPAT=${KPAT:-mlp_} python - "$f" <<'PY'
import csv,sys,collections,os
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if os.environ["KPAT"] in r['Kernel_Name']:
        agg[(r['Kernel_Name'][6:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(agg.items()): print(f"  {c:32s} {sum(v)/len(v):16.0f}  x{len(v)}  {k}")
PY
done
