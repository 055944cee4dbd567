cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for v in ""; do
  GH_RASTER_LIB=$v timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items() if 'render' in k})"
done
