cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1e
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r1e/pmc_$tag -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stage-timing > gpurun_out/r1e/pmc_$tag.log 2>&1
  f=$(find gpurun_out/r1e/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 tools/summarize_pmc.py $f | grep -E "gh_render|gh_radix_scatter|gh_ranges|gh_record" > gpurun_out/r1e/sum_$tag.csv
  rm -rf gpurun_out/r1e/pmc_$tag
done
cat gpurun_out/r1e/sum_*.csv
