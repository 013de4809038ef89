export PYTHONPATH=.
python -m pytest tests/test_gpu_mlp.py -x -q 2>&1 | tail -8
