# timing of variant builds (tools/abl/*.so, built by tools/abl_build.sh) on the default bench workload: stage times per variant.
# The variants are loaded through GH_RASTER_LIB; the in-tree library is never touched.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for f in tools/abl/*.so; do
  v=$(basename $f .so)
  GH_RASTER_LIB=$GRAFT_REPO_ROOT/$f timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items()})" | tee -a gpurun_out/abl.log
done
done
