cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
for v in "" tools/abl/v2_l64.so tools/abl/v2_nobody.so tools/abl/old_bwd.so; do
  GH_RASTER_LIB=$v timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items()})"
done
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$tag -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stage-timing > $O/pmc_$tag.log 2>&1
  python3 tools/summarize_pmc.py $(find $O/pmc_$tag -name "*counter_collection.csv" | head -1) | grep -E "gh_render_bwd" > $O/sum_$tag.csv
  rm -rf $O/pmc_$tag
done
cat $O/sum_*.csv
