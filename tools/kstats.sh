# Quick per-kernel time table of the default bench step (usage: bash tools/kstats.sh [bench args]); prints the top kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline "$@" > $O/bench.json 2> $O/stats.log || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kstats/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f}")
PY
rm -rf $O/stats
