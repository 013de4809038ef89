export PYTHONPATH=.
DEMF_STATIC_TILES=1 DEMF_GEO_AT_FWD=1 DEMF_NO_RED_FUSE=1 DEMF_NO_FIRST_FUSE=1 DEMF_SHARE_DEVICE=1 DEMF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-260
