cd $GRAFT_REPO_ROOT
for v in "" tools/abl/v3_NOSCAN.so tools/abl/v3_NONOP.so tools/abl/v3_NORCP.so tools/abl/v3_NOEXP.so tools/abl/v3_NOSTATE.so tools/abl/v3_NOADD.so; do
  GH_RASTER_LIB=$v timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items() if 'render' in k})"
done
