export PYTHONPATH=.
python -m pytest tests/test_gpu_mlp.py tests/test_gpu_model.py tests/test_abi.py -x -q 2>&1 | tail -5
for i in 1 2; do python bench.py --steps 40 2>&1 | tail -1 | cut -c100-240; DEMF_NO_RED_FUSE=1 python bench.py --steps 40 2>&1 | tail -1 | cut -c100-240; done
