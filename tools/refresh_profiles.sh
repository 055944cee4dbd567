# Regenerates everything under profiles/ for the current build (usage: bash tools/refresh_profiles.sh <tag> [a|b|all]; a = the three
# profiled workloads + micro benchmarks + timelines, b = the host-side figures and every bench line: two GPU calls of <= 20 minutes each)
# Per workload: the bench line, rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ passes.
TAG=${1:-r6}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python3 -c "import bench; print(bench.source_hash())" > $O/source_hash.txt
profile_workload() {   # <name> <bench args...>
  local W=$1; shift
  local D=$O/$W
  mkdir -p $D
  echo "== [$W] kernel stats"
  timeout 400 rocprofv3 --kernel-trace --stats -d $D/stats -o st --output-format csv -- python3 bench.py "$@" --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $D/stats_bench.json 2> $D/stats.log || return 1
  cp $(find $D/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv && rm -rf $D/stats
  echo "== [$W] traffic"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d $D/pmc_$C -o pmc --output-format csv -- python3 bench.py "$@" --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-stage-timing > $D/pmc_$C.log 2>&1 || return 1
    cp $(find $D/pmc_$C -name "*counter_collection.csv" | head -1) $D/$C.csv && rm -rf $D/pmc_$C
  done
  python3 tools/make_pmc_traffic.py $D/FETCH_SIZE.csv $D/WRITE_SIZE.csv $D/pmc_traffic.json $D/pmc_fetch_write.csv "$W: bench.py $*" > /dev/null || return 1
  rm -f $D/FETCH_SIZE.csv $D/WRITE_SIZE.csv
  echo "== [$W] sq counters"
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
    tag=$(echo $C | cut -d' ' -f1)
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d $D/pmc_$tag -o pmc --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-stage-timing > $D/pmc_$tag.log 2>&1 || return 1
    python3 tools/summarize_pmc.py $(find $D/pmc_$tag -name "*counter_collection.csv" | head -1) > $D/sum_$tag.csv
    rm -rf $D/pmc_$tag
  done
}
if [ $PART != b ]; then
profile_workload default || exit 1
profile_workload hd_sh3 --config two_hands_hd || exit 1
profile_workload hd_sh3_pose32 --config two_hands_hd --pose-batch --views-per-step 32 || exit 1
echo "== valu issue costs (s_memtime / wall clock, then GRBM cycles under the profiler)"
(hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cycles tools/micro/valu_cycles.hip && timeout 200 /tmp/valu_cycles > $O/valu_rate_wallclock.txt) 2> $O/valu_rate.err
bash tools/micro/valu_cycles_pmc.sh $O > $O/valu_cycles_pmc.log 2>&1
echo "== kernel boundary floor, FETCH_SIZE calibration, one-step timelines"
bash tools/micro/kernel_floor.sh $O > /dev/null 2>&1
bash tools/micro/fetch_calib.sh $O > /dev/null 2>&1
bash tools/kernel_timeline.sh > $O/timeline_8view.txt 2>&1
bash tools/kernel_timeline.sh --views-per-step 1 > $O/timeline_1view.txt 2>&1
fi
if [ $PART = a ]; then echo "part a done"; exit 0; fi
echo "== two-call protocol, host breakdown of the drop-in, fit step"
timeout 300 python3 tools/two_call_cost.py > $O/two_call_cost.txt 2> $O/two_call_cost.err
timeout 300 python3 tools/dropin_time.py 2> /dev/null | grep -v amdgpu.ids > $O/dropin_host_breakdown.txt
(timeout 300 python3 tools/fit_static_time.py 2> /dev/null | grep -v amdgpu.ids; bash tools/fit_kstats.sh 1 2> /dev/null | head -24) > $O/fit_step_profile.txt
# the bench lines last: they quote the counter summaries of THIS build (profiles/<tag>_pmc_*.json, written here on the box by the
# assemble step; run tools/assemble_profiles.py again at home to pick up the bench lines themselves)
# (run in two parts, the counter summaries of part a were assembled at home and travel with the tree: nothing to assemble on this box)
if [ -d $O/default ]; then python3 tools/assemble_profiles.py $TAG > /dev/null || exit 1; fi
for V in 2 4 16; do
  echo "== bench $V views" && timeout 300 python3 bench.py --views-per-step $V --no-cpu-baseline --no-stage-timing > $O/bench_${V}view.json 2> $O/bench_${V}view.err || exit 1
done
echo "== bench split streams" && timeout 300 python3 bench.py --split-streams on --no-cpu-baseline --no-stage-timing > $O/bench_split.json 2> $O/bench_split.err || exit 1
for V in 1 2 4; do timeout 200 python3 tools/fit_static_time.py $V 2> /dev/null | grep "fit step" >> $O/fit_step_views.txt; done
echo "== bench 1 view" && timeout 300 python3 bench.py --views-per-step 1 --no-cpu-baseline > $O/bench_1view.json 2> $O/bench_1view.err || exit 1
echo "== bench hd sh3" && timeout 300 python3 bench.py --config two_hands_hd --no-cpu-baseline > $O/bench_hd_sh3.json 2> $O/bench_hd.err || exit 1
echo "== bench hd sh3, split streams" && timeout 300 python3 bench.py --config two_hands_hd --split-streams on --no-cpu-baseline --no-stage-timing > $O/bench_hd_sh3_split.json 2> $O/bench_hd_split.err || exit 1
echo "== bench hd sh3 pose batch 32" && timeout 400 python3 bench.py --config two_hands_hd --pose-batch --views-per-step 32 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $O/bench_hd_sh3_pose32.json 2> $O/bench_hd_pb.err || exit 1
echo "== bench hd sh3 pose batch 32, split streams" && timeout 400 python3 bench.py --config two_hands_hd --pose-batch --views-per-step 32 --steps 5 --warmup 2 --repeats 3 --split-streams on --no-cpu-baseline --no-stage-timing > $O/bench_hd_sh3_pose32_split.json 2> $O/bench_hd_pb_split.err || exit 1
echo "== bench random1k (configs[0]: PyTorch CPU autograd baseline)" && timeout 300 python3 bench.py --config random1k --views-per-step 1 > $O/bench_random1k.json 2> $O/bench_random1k.err || exit 1
echo "== bench default" && timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo done
