# Per-kernel time table of the full-size fit step (usage: bash tools/fit_kstats.sh <static 0|1>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/fit_kstats
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 tools/fit_loop.py $1 22 ${2:-8} > $O/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/fit_kstats/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows[:30]:
    print(f"{r['Name'][:56]:56s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f} max_us {float(r['MaxNs'])/1e3:8.2f}")
PY
rm -rf $O/stats
