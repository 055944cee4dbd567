# One GPU call that regenerates everything under profiles/ for the current build (usage: bash tools/refresh_profiles.sh <tag>)
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
echo "== bench 1 view" && timeout 300 python3 bench.py --views-per-step 1 --no-cpu-baseline > $O/bench_1view.json 2> $O/bench_1view.err || exit 1
echo "== bench hd sh3" && timeout 300 python3 bench.py --config two_hands_hd --no-cpu-baseline > $O/bench_two_hands_hd_sh3.json 2> $O/bench_hd.err || exit 1
echo "== bench hd sh3 pose batch 32" && timeout 400 python3 bench.py --config two_hands_hd --pose-batch --views-per-step 32 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $O/bench_two_hands_hd_sh3_pose_batch32.json 2> $O/bench_hd_pb.err || exit 1
echo "== two-call protocol" && timeout 300 python3 tools/two_call_cost.py > $O/two_call_cost.txt 2> $O/two_call_cost.err || exit 1
echo "== valu rate microbenchmark"
(hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/micro/valu_rate.hip && timeout 120 /tmp/valu_rate > $O/valu_rate.txt) 2> $O/valu_rate.err
echo "== kernel stats"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.log || exit 1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/stats
echo "== kernel stats, 1024^2 SH3"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_hd -o st --output-format csv -- python3 bench.py --config two_hands_hd --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/stats_hd_bench.json 2> $O/stats_hd.log || exit 1
cp $(find $O/stats_hd -name "*kernel_stats.csv" | head -1) $O/kernel_stats_hd.csv && rm -rf $O/stats_hd
echo "== traffic"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-stage-timing > $O/pmc_$C.log 2>&1 || exit 1
  cp $(find $O/pmc_$C -name "*counter_collection.csv" | head -1) $O/$C.csv && rm -rf $O/pmc_$C
done
python3 tools/make_pmc_traffic.py $O/FETCH_SIZE.csv $O/WRITE_SIZE.csv $O/pmc_traffic.json $O/pmc_fetch_write_8views.csv > /dev/null
rm -f $O/FETCH_SIZE.csv $O/WRITE_SIZE.csv
echo "== sq counters"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$tag -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-stage-timing > $O/pmc_$tag.log 2>&1 || exit 1
  python3 tools/summarize_pmc.py $(find $O/pmc_$tag -name "*counter_collection.csv" | head -1) > $O/sum_$tag.csv
  rm -rf $O/pmc_$tag
done
python3 -c "import bench; print(bench.source_hash())" > $O/source_hash.txt
echo "== host breakdown of the drop-in, fit step"
timeout 300 python3 tools/dropin_time.py 2> /dev/null | grep -v amdgpu.ids > $O/dropin_host_breakdown.txt
timeout 300 python3 tools/fit_profile.py 2> /dev/null | grep -v amdgpu.ids > $O/fit_step_profile.txt
# the default bench line last: it quotes the counter summaries of THIS build (profiles/<tag>_pmc_*.json, written here on the box
# by the assemble step; run tools/assemble_profiles.py again at home to pick up the bench line itself)
python3 tools/assemble_profiles.py $TAG > /dev/null || exit 1
echo "== bench default" && timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo done
