export PYTHONPATH=.
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>&1 | tail -1 > gpurun_out/bench_l.json; cut -c1-200 gpurun_out/bench_l.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l -o l --output-format csv -- python bench.py > gpurun_out/bench_l_prof.log 2>&1
f=$(find gpurun_out/prof_l -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/stats_l.csv
rm -rf gpurun_out/prof_l
