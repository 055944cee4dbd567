cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-p1v}; mkdir -p $O
timeout 200 python3 bench.py --views-per-step 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['config']['repeats']['ms_per_step_median'], {k: round(v['ms'],4) for k,v in d['stages'].items()})"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --views-per-step 1 --steps 10 --warmup 3 --no-cpu-baseline --repeats 1 > $O/stats_bench.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/stats
python3 - <<P
import csv
rows=list(csv.reader(open("$O/kernel_stats.csv")))
for r in rows[1:32]:
    print(r[0][:58].ljust(58), r[1].rjust(4), r[3].rjust(9), r[4].rjust(7))
P
