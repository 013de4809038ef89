export PYTHONPATH=.
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in 16 64 128 256; do echo "rows $r  $(DEMF_COLSUM_ROWS=$r python bench.py --steps 40 2>&1 | tail -1 | ms)"; done
echo "rows 16  $(DEMF_COLSUM_ROWS=16 python bench.py --steps 40 2>&1 | tail -1 | ms)"
