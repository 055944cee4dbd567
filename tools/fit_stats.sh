cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fit
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/fit/stats -o st --output-format csv -- python3 tools/fit_step_stats.py > gpurun_out/fit/out.txt 2>&1
cp $(find gpurun_out/fit/stats -name "*kernel_stats.csv" | head -1) gpurun_out/fit/kernel_stats.csv
rm -rf gpurun_out/fit/stats
grep "fit step" gpurun_out/fit/out.txt
