#!/bin/bash
# usage: bash tools/prof.sh <tag> [bench args...]   (on the GPU box, from the repo root)
# rocprofv3 kernel-trace + stats of bench.py -> gpurun_out/<tag>/{kernel_stats.csv,steps_summary.txt}
# PMC=1: additionally two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), summarised per kernel.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc-live "$@" > $out/prof.log 2>&1
st=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
tr=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
[ -n "$st" ] && cp $st $out/kernel_stats.csv
[ -n "$tr" ] && LIBSEQ=1 TOPN=${TOPN:-45} DETAIL=${DETAIL:-demf::gemm_kernel} python $R/tools/trace_summary.py $tr 5 40 7 > $out/steps_summary.txt 2>&1
[ -n "$tr" ] && python $R/tools/step_timeline.py $tr > $out/step_timeline.txt 2>&1
tail -1 $out/prof.log | head -c 400 > $out/bench_under_profiler.txt
if [ -n "$PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$tag_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc-live "$@" > $out/pmc_$c.log 2>&1
    f=$(find /tmp/pmc_${tag}_$c -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f $c > $out/pmc_$c.csv
  done
fi
pf=""; [ -n "$PMC" ] && pf="$out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv"
[ -n "$tr" ] && python $R/tools/top_kernels.py $tr $pf 5 7 > $out/top_kernels.json 2>> $out/prof.log
cd $R
head -45 $out/steps_summary.txt; grep -A40 -e '^--- ' $out/steps_summary.txt | head -60
