# SQ counters of the render kernels on the default bench workload -> gpurun_out/$1/sum_*.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc}; mkdir -p $O
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$tag -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stage-timing > $O/pmc_$tag.log 2>&1
  python3 tools/summarize_pmc.py $(find $O/pmc_$tag -name "*counter_collection.csv" | head -1) | grep -E "${2:-gh_render}" > $O/sum_$tag.csv
  rm -rf $O/pmc_$tag
done
cat $O/sum_*.csv
