export PYTHONPATH=.
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
echo "fwd order $(DEMF_SKIP_GEO=1 python bench.py --steps 40 2>&1 | tail -1 | ms)"
echo "reversed  $(DEMF_SKIP_GEO=1 DEMF_BNRED_REV=1 python bench.py --steps 40 2>&1 | tail -1 | ms)"
done
