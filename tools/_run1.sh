export PYTHONPATH=.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in 1024 2048 4096; do
DEMF_SPARSE_DW_GRID=$g rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l -o l --output-format csv -- python bench.py --steps 6 --warmup 2 > gpurun_out/bench_l_prof.log 2>&1
python - <<PY
import csv,glob,collections
t=glob.glob('gpurun_out/prof_l/**/*kernel_trace.csv', recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    if 'sparse_dw_gram' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:34], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    print("grid $g", f"x{len(v):4d} avg {sum(v)/len(v):8.1f} us min {min(v):8.1f}  {k}")
PY
rm -rf gpurun_out/prof_l
done
