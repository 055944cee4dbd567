# same-box A/B of library variants (tools/abl/<name>.so, tools/abl_build.sh): per-kernel averages of the bench step.
# usage: bash tools/ab_k.sh '<kernel regex>' '<bench args>' name1 name2 ...   ('tree' = the in-tree library)
RX=$1; ARGS=$2; shift 2
for r in 1 2; do
  for v in "$@"; do
    echo "== $v [$ARGS] round $r"
    if [ "$v" = tree ]; then bash tools/kstats.sh $ARGS 2>&1 | grep -E "$RX"
    else GH_RASTER_LIB=$GRAFT_REPO_ROOT/tools/abl/$v.so bash tools/kstats.sh $ARGS 2>&1 | grep -E "$RX"; fi
    python3 -c "import json; d=json.load(open('gpurun_out/kstats/bench.json')); print('   bench', round(d['value']), d['ms_per_step'], {k: round(v['ms'],4) for k,v in d.get('stages',{}).items()})"
  done
done
