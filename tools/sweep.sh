# The configurations quoted in DESIGN.md section 7, one line each (gpurun_out/sweep.log)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/sweep.log
: > $L
run() { echo "## $*" >> $L; timeout 300 python bench.py --no-cpu-baseline --no-stage-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'renders/s', round(d['ms_per_step'],3), 'ms/step', 'graph' if d['config']['hip_graph'] else 'eager', 'enqueue', round(d['config']['host_enqueue_ms_per_step'],3))" >> $L; }
run
run --no-graph
run --views-per-step 16
run --views-per-step 4
run --views-per-step 1
run --views-per-step 1 --no-graph
run --config one_hand
run --config two_hands_hd
run --config two_hands_hd --pose-batch
python tools/two_call_cost.py >> $L 2>&1
python tools/fit_profile.py 2>&1 | grep "fit step" >> $L
cat $L
