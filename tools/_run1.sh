python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -3
python bench.py 2>&1 | tail -1 | cut -c1-260
