#!/bin/bash
# end-of-round artefacts on the GPU box: kernel stats + step digests (f32 with the PMC passes, bf16), the full bench line,
# the bf16 line and the resident step with / without the pre-pass -> gpurun_out/<tag>_*; tools/collect_final.sh copies
# them to profiles/.   usage: bash tools/final_run.sh r04_final
tag=${1:-r04_final}
export PYTHONPATH=.
PMC=1 bash tools/prof.sh ${tag}_f32 > /dev/null 2>&1
bash tools/prof.sh ${tag}_bf16 --dtype bf16 > /dev/null 2>&1
head -3 gpurun_out/${tag}_f32/steps_summary.txt; head -3 gpurun_out/${tag}_bf16/steps_summary.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_f32/bench_f32.json
python bench.py --dtype bf16 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/${tag}_bf16/bench_bf16.json
python bench.py --resident --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/${tag}_f32/bench_f32_resident.json
DEMF_SKIP_GEO=1 python bench.py --resident --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/${tag}_f32/bench_f32_skipgeo.json
for f in gpurun_out/${tag}_f32/bench_f32.json gpurun_out/${tag}_bf16/bench_bf16.json gpurun_out/${tag}_f32/bench_f32_resident.json gpurun_out/${tag}_f32/bench_f32_skipgeo.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['repeat_ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('secondary'))"; done
