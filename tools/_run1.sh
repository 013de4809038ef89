export PYTHONPATH=.
python -m pytest tests/test_gpu_ops.py -x -q -k "invert or group or gather" 2>&1 | tail -3
python tools/inv_list_stats.py 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 40 2>&1 | tail -1 | cut -c100-240; done
