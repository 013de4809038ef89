export PYTHONPATH=.
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
echo "default      $(python bench.py --steps 60 2>&1 | tail -1 | ms)"
echo "GEO_AT_FWD   $(DEMF_GEO_AT_FWD=1 python bench.py --steps 60 2>&1 | tail -1 | ms)"
echo "DW_DYN       $(DEMF_DW_DYN=1 python bench.py --steps 60 2>&1 | tail -1 | ms)"
echo "STATIC_TILES $(DEMF_STATIC_TILES=1 python bench.py --steps 60 2>&1 | tail -1 | ms)"
done
