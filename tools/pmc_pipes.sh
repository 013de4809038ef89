export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  DEMF_FWD_LDS=$mode python $GRAFT_REPO_ROOT/tools/sa1_l3_micro.py 20
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
    rm -rf /tmp/pm; DEMF_FWD_LDS=$mode rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/sa1_l3_micro.py 4 > /dev/null 2>&1
    f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
    python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'mlp_' in r['Kernel_Name']: agg[(r['Kernel_Name'][12:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(agg.items()): print(f"  {c:32s} {sum(v)/len(v):16.0f}   {k}")
PY
  done
done
