python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -3
python bench.py 2>&1 | tail -1 | cut -c1-260
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k -o k --output-format csv -- python bench.py --steps 20 --warmup 5 > gpurun_out/bench_k.log 2>&1
f=$(find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/stats_k.csv
t=$(find gpurun_out/prof_k -name "*kernel_trace.csv" | head -1); cp $t gpurun_out/trace_k.csv
rm -rf gpurun_out/prof_k
