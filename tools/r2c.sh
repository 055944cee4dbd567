cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
echo "== accuracy new"; timeout 300 python tools/bwd_accuracy.py two_hands 2 2>&1 | tail -3
echo "== accuracy old"; GH_RASTER_LIB=tools/abl/old_bwd.so timeout 300 python tools/bwd_accuracy.py two_hands 2 2>&1 | tail -3
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$tag -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stage-timing > $O/pmc_$tag.log 2>&1
  python3 tools/summarize_pmc.py $(find $O/pmc_$tag -name "*counter_collection.csv" | head -1) | grep -E "gh_render" > $O/sum_$tag.csv
  rm -rf $O/pmc_$tag
done
cat $O/sum_*.csv
