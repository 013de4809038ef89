python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -5
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
