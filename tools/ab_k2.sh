# same-box A/B of library variants of ONE tree (tools/abl/<name>.so built by tools/abl_build.sh; 'tree' = the in-tree library):
# per-kernel rocprofv3 averages + the un-profiled bench line. usage: bash tools/ab_k2.sh '<kernel regex>' '<bench args>' name1 name2 ...
RX=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset GH_RASTER_LIB; else export GH_RASTER_LIB=$GRAFT_REPO_ROOT/tools/abl/$v.so; fi
    O=gpurun_out/kstats_$v
    rm -rf $O && mkdir -p $O
    echo "== $v [$ARGS] round $r"
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline $ARGS > $O/bench.json 2> $O/stats.log || { echo FAILED; tail -5 $O/stats.log; continue; }
    python3 - "$O" "$RX" <<'PY'
import csv, glob, re, sys
O, rx = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
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
