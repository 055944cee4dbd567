# HBM traffic per kernel: two separate counter passes (FETCH_SIZE, WRITE_SIZE) over the default bench workload.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
exec > gpurun_out/traffic/log.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/traffic/pmc_$C -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timing > gpurun_out/traffic/pmc_$C.log 2>&1
  cp $(find gpurun_out/traffic/pmc_$C -name "*counter_collection.csv" | head -1) gpurun_out/traffic/$C.csv
  rm -rf gpurun_out/traffic/pmc_$C
done
python3 tools/make_pmc_traffic.py gpurun_out/traffic/FETCH_SIZE.csv gpurun_out/traffic/WRITE_SIZE.csv gpurun_out/traffic/pmc_traffic.json gpurun_out/traffic/pmc_fetch_write_8views.csv
rm -f gpurun_out/traffic/FETCH_SIZE.csv gpurun_out/traffic/WRITE_SIZE.csv
