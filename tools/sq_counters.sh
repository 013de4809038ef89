#!/bin/bash
# SQ counters of the step's big kernels ("or report SQ counters"): separate --pmc passes of a short
# bench run, per-kernel means -> gpurun_out/<tag>_sq.txt.   usage (GPU box, repo root): bash tools/sq_counters.sh r06_final
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${tag}_sq
mkdir -p $out
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/sq_${tag}_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/sq_${tag}_$i -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc-live > $out/pass$i.log 2>&1
  f=$(find /tmp/sq_${tag}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
done
cd $R
python tools/sq_summary.py $out/pass*.csv > gpurun_out/${tag}_sq.txt 2>&1
cat gpurun_out/${tag}_sq.txt
