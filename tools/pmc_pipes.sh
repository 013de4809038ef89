# Pipe-utilisation counters of the dominant kernel alone (tools/sa1_l3_micro.py) per compute mode:
# bash tools/pmc_pipes.sh   (on the GPU box; separate --pmc passes, no tracing options)
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-f32_native f32x3 bf16}; do
  DEMF_MODE=$mode python $GRAFT_REPO_ROOT/tools/sa1_l3_micro.py 20
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" ${EXTRA_GROUPS:+"$EXTRA_GROUPS"}; do
    rm -rf /tmp/pm; DEMF_MODE=$mode rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/sa1_l3_micro.py 4 > /dev/null 2>&1
    f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
    python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'mlp_' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][12:64], r['Counter_Name'])].append(float(r['Counter_Value']))
        if 'End_Timestamp' in r: agg[(r['Kernel_Name'][12:64], 'duration_ns')].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for (k,c),v in sorted(agg.items()): print(f"  {c:32s} {sum(v)/len(v):16.0f}   {k}")
PY
  done
done
