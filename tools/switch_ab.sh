#!/bin/bash
# Same-box A/B of the round's kernel-form switches on the resident step (bench.py --resident): one line per setting.
# usage (GPU box, repo root): bash tools/switch_ab.sh > gpurun_out/<tag>_switch_ab.txt
run() { python bench.py --resident --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-62s %.3f ms  (repeats %.3f %.3f)' % ('$1', d['ms_per_step'], *d['repeat_ms_per_step']))"; }
run "defaults"
DEMF_F16_TERMS=0 run "DEMF_F16_TERMS=0 (three bf16 terms everywhere)"
DEMF_F16_TERMS_BWD=0 run "DEMF_F16_TERMS_BWD=0 (two fp16 terms in the forward kernels only)"
DEMF_GF_RECOMPUTE=0 DEMF_GF_WACC=0 DEMF_GF_FIN=0 run "group_first: y read back, in-place atomics, finalize launch"
DEMF_ACC_REPL=0 run "DEMF_ACC_REPL=0 (no replicated accumulators)"
DEMF_INVERT_SPLIT=2 run "DEMF_INVERT_SPLIT=2 (SA2 inverse lists: five whole-GPU launches)"
DEMF_FWD_TILE=0 DEMF_DX_TILE=0 run "DEMF_FWD_TILE=0 DEMF_DX_TILE=0 (few-row layers on mlp_gemm_kernel)"
DEMF_GRAPH_UPDATE=0 run "DEMF_GRAPH_UPDATE=0 (update eager behind the graph)"
DEMF_SKIP_GEO=1 run "DEMF_SKIP_GEO=1 (no pre-pass underneath: the step alone)"
run "defaults (again)"
