# per-kernel time of the default bench workload: rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>/kernel_stats.csv
# usage: bash tools/prof_stats.sh [tag]
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/stats_bench.json 2> gpurun_out/$TAG/stats.log
f=$(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/stats
python3 - <<P
import csv
rows=list(csv.reader(open("gpurun_out/$TAG/kernel_stats.csv")))
for r in rows[1:26]:
    print(r[0][:60].ljust(60), r[1].rjust(4), r[3].rjust(9), r[4].rjust(7))
P
