# timing of variant builds (tools/abl/*.so) on the default bench workload: stage times per variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp guassianhand_amd/libgh_raster.so /tmp/keep.so
for rep in 1 2; do
for f in tools/abl/*.so; do
  v=$(basename $f .so)
  cp $f guassianhand_amd/libgh_raster.so
  timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items()})" | tee -a gpurun_out/abl.log
done
done
cp /tmp/keep.so guassianhand_amd/libgh_raster.so
