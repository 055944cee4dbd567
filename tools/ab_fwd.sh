# same-box A/B of two builds of the library: tools/abl/<name>.so (default base1) against the in-tree one (kernel stats of the default bench)
B=${1:-base1}
for r in 1 2 3; do
  echo "== $B";  GH_RASTER_LIB=$GRAFT_REPO_ROOT/tools/abl/$B.so bash tools/kstats.sh 2>&1 | grep -E "render_fwd|render_bwd"
  echo "== new";   bash tools/kstats.sh 2>&1 | grep -E "render_fwd|render_bwd"
done
