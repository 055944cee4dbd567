# Sweep of the fine-grained forward's knobs on the in-tree library: bench line + stage times.
# usage: bash tools/fine_sweep.sh '<bench args>' 'K:MIN[:ORDER[:CLASSES]] ...'   (GH_FWD_FINE_K tiles at the head of the launch order walked in the fine
# form if their list holds at least GH_FWD_FINE_MIN entries; ORDER 0 = launch order by list length, 1 (default) = by the previous forward's measurements, ranked inside the projection
# kernel; CLASSES 0 = the backward's work list in one piece, 1 (default) = in regions: DESIGN §5)
ARGS=$1; shift
for r in 1 2; do
for km in $@; do
  IFS=: read K MIN ORD CLS <<< "$km"
  export GH_FWD_FINE_K=$K GH_FWD_FINE_MIN=$MIN GH_FWD_HEAVY_ORDER=${ORD:-1} GH_BWD_CLASSES=${CLS:-1}
  python3 bench.py --steps 50 --warmup 20 --repeats 3 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('K:MIN:ORDER $km', round(d['value']), 'renders/s', round(d['ms_per_step'],4), 'ms; median', round(c['repeats']['ms_per_step_median'],4), {k: round(v['ms'],4) for k,v in d.get('stages',{}).items()})"
done
done
