export PYTHONPATH=.
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_l.json; cut -c1-240 gpurun_out/bench_l.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
