#!/bin/bash
# end-of-round artefacts on the GPU box: kernel stats + step digests (f32 with the PMC passes, bf16), the full bench line,
# the bf16 line and the step without the pre-pass -> gpurun_out/r03_final_*; tools/collect_final.sh copies them to profiles/
export PYTHONPATH=.
PMC=1 bash tools/prof.sh r03_final_f32 > /dev/null 2>&1
bash tools/prof.sh r03_final_bf16 --dtype bf16 > /dev/null 2>&1
head -3 gpurun_out/r03_final_f32/steps_summary.txt; head -3 gpurun_out/r03_final_bf16/steps_summary.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_final_f32/bench_f32.json
python bench.py --dtype bf16 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/r03_final_bf16/bench_bf16.json
DEMF_SKIP_GEO=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/r03_final_f32/bench_f32_skipgeo.json
for f in gpurun_out/r03_final_f32/bench_f32.json gpurun_out/r03_final_bf16/bench_bf16.json gpurun_out/r03_final_f32/bench_f32_skipgeo.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['repeat_ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('secondary'))"; done
