# timing-only ablations of the backward kernel (results are wrong by construction): which part of a trip costs what
cd $GRAFT_REPO_ROOT
cp guassianhand_amd/libgh_raster.so /tmp/keep.so
for v in G H I J; do
  cp tools/abl/$v.so guassianhand_amd/libgh_raster.so
  timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k: round(v['ms'],3) for k,v in d['stages'].items()})"
done
cp /tmp/keep.so guassianhand_amd/libgh_raster.so
