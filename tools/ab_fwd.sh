# same-box A/B of two builds of the library: tools/abl/base.so against the in-tree one (kernel stats of the default bench)
for r in 1 2; do
  echo "== base";  GH_RASTER_LIB=$GRAFT_REPO_ROOT/tools/abl/base.so bash tools/kstats.sh 2>&1 | grep -E "render_fwd|render_bwd|ranges|preprocess_bwd|record_sum"
  echo "== new";   bash tools/kstats.sh 2>&1 | grep -E "render_fwd|render_bwd|ranges|preprocess_bwd|record_sum"
done
