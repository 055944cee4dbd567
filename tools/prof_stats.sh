cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1e
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r1e/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r1e/stats_bench.json 2> gpurun_out/r1e/stats.log
f=$(find gpurun_out/r1e/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r1e/kernel_stats.csv
rm -rf gpurun_out/r1e/stats
head -32 gpurun_out/r1e/kernel_stats.csv | cut -c1-150
