# same-box A/B of library builds on the moving-geometry loop (tools/moving_kstats.sh): usage: bash tools/ab_moving.sh <views> <refresh_every> name1 name2 ...  ('tree' = in-tree)
V=$1; RF=$2; shift 2
for v in "$@"; do
  echo "#### $v (views $V, refresh_every $RF)"
  if [ "$v" = tree ]; then bash tools/moving_kstats.sh $V two_hands $RF 2>&1 | grep -E "==|render_fwd|all gh"
  else GH_RASTER_LIB=$GRAFT_REPO_ROOT/tools/abl/$v.so bash tools/moving_kstats.sh $V two_hands $RF 2>&1 | grep -E "==|render_fwd|all gh"; fi
done
