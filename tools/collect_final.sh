#!/bin/bash
# copies the artefacts of tools/final_run.sh (gpurun_out/<tag>_*) into profiles/ under the round's names
tag=${1:-r04_final}
for m in f32 bf16; do
  cp gpurun_out/${tag}_$m/kernel_stats.csv profiles/${tag}_${m}_kernel_stats.csv
  cp gpurun_out/${tag}_$m/steps_summary.txt profiles/${tag}_${m}_steps_summary.txt
  cp gpurun_out/${tag}_$m/top_kernels.json profiles/${tag}_${m}_top_kernels.json
done
cp gpurun_out/${tag}_f32/top_kernels.json profiles/${tag}_top_kernels.json
cp gpurun_out/${tag}_f32/pmc_FETCH_SIZE.csv profiles/${tag}_pmc_FETCH_SIZE.csv
cp gpurun_out/${tag}_f32/pmc_WRITE_SIZE.csv profiles/${tag}_pmc_WRITE_SIZE.csv
cp gpurun_out/${tag}_f32/bench_f32.json profiles/${tag}_bench_f32.json
cp gpurun_out/${tag}_bf16/bench_bf16.json profiles/${tag}_bench_bf16.json
cp gpurun_out/${tag}_f32/bench_f32_resident.json profiles/${tag}_bench_f32_resident.json
cp gpurun_out/${tag}_f32/bench_f32_skipgeo.json profiles/${tag}_bench_f32_skipgeo.json
cp gpurun_out/${tag}_f32/step_timeline.txt profiles/${tag}_f32_step_timeline.txt
head -2 profiles/${tag}_pmc_FETCH_SIZE.csv profiles/${tag}_pmc_WRITE_SIZE.csv
