export PYTHONPATH=.
python -m pytest tests/test_gpu_mlp.py tests/test_gpu_model.py -x -q 2>&1 | tail -5
for i in 1 2; do python bench.py --steps 40 2>&1 | tail -1 | cut -c100-240; done
