# same-box A/B of a whole source tree (tools/abl/<name>/ = `git archive` of another commit, built there) against the working
# tree: bench lines (graph replay) alternating, 2 rounds. usage: bash tools/ab_tree.sh <name> '<bench args>'
NAME=$1; ARGS=$2
for r in 1 2; do
  for t in $NAME tree; do
    if [ "$t" = tree ]; then D=$GRAFT_REPO_ROOT; else D=$GRAFT_REPO_ROOT/tools/abl/$t; fi
    (cd $D && python3 bench.py --steps 50 --warmup 20 --repeats 5 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$t [$ARGS] round $r:', round(d['value']), 'renders/s', d['ms_per_step'], 'ms; windows', c.get('repeats'), 'step median', (c.get('step_ms') or {}).get('median'), {k: round(v['ms'],4) for k,v in d.get('stages',{}).items()})")
  done
done
