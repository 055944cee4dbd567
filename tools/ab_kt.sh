# same-box A/B of whole source trees: per-kernel rocprofv3 averages of the bench step + the un-profiled bench line, alternating.
# usage: bash tools/ab_kt.sh '<kernel regex>' '<bench args>' tree1 tree2 ...   (a tree = tools/abl/<name>/, `git archive` of a commit
# built in place; 'tree' = the working tree). Optional env ROUNDS (default 2).
RX=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 ${ROUNDS:-2}); do
  for t in "$@"; do
    if [ "$t" = tree ]; then D=$GRAFT_REPO_ROOT; else D=$GRAFT_REPO_ROOT/tools/abl/$t; fi
    cd $D
    O=$GRAFT_REPO_ROOT/gpurun_out/kstats_$t
    rm -rf $O && mkdir -p $O
    echo "== $t [$ARGS] round $r"
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline $ARGS > $O/bench.json 2> $O/stats.log || { echo FAILED; tail -5 $O/stats.log; continue; }
    python3 - "$O" "$RX" <<'PY'
import csv, glob, re, sys
O, rx = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    if r['Name'].startswith('gh_') or 'gh_' in r['Name'][:12]:
        pass
    if re.search(rx, r['Name']):
        print(f"   {r['Name'][:58]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f}")
PY
    rm -rf $O/stats
    python3 bench.py --steps 50 --warmup 20 --repeats 3 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('   bench', round(d['value']), 'renders/s', round(d['ms_per_step'],4), 'ms; windows median', round(c['repeats']['ms_per_step_median'],4), {k: round(v['ms'],4) for k,v in d.get('stages',{}).items()})"
  done
done
