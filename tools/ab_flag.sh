# same-box A/B of bench FLAG variants on the in-tree library: per-kernel averages (rocprofv3) of the bench step.
# usage: bash tools/ab_flag.sh '<kernel regex>' '<common bench args>' '<variant args 1>' '<variant args 2>' ...
RX=$1; ARGS=$2; shift 2
for r in 1 2; do
  for v in "$@"; do
    echo "== [$ARGS $v] round $r"
    bash tools/kstats.sh $ARGS $v 2>&1 | grep -E "$RX"
    python3 -c "import json; d=json.load(open('gpurun_out/kstats/bench.json')); print('   bench', round(d['value']), d['ms_per_step'], {k: round(v['ms'],4) for k,v in d.get('stages',{}).items()})"
  done
done
