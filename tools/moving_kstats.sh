# Per-kernel GPU time per step of the moving-geometry loop without / with the occlusion bound (rocprofv3 kernel stats: immune to
# the host-bound eager loop). usage: bash tools/moving_kstats.sh [views] [config]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
V=${1:-8}; CFG=${2:-two_hands}; RF=${3:-4}
for MODE in none bound; do
  O=gpurun_out/mkstats_$MODE
  rm -rf $O && mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 tools/moving_geometry.py $V 1e-4 2e-3 $CFG 8 $MODE $RF > $O/out.txt 2> $O/log.txt || exit 1
  python3 - $MODE <<'PY'
import csv, glob, sys
mode = sys.argv[1]
f = glob.glob(f"gpurun_out/mkstats_{mode}/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 28.0                       # 4 sync + 24 timed steps (+ the ground-truth forward)
tot = 0.0
print(f"== {mode}")
for r in rows[:16]:
    us = float(r["TotalDurationNs"]) / 1e3 / steps
    print(f"  {r['Name'][:58]:58s} calls {r['Calls']:>5s} us/step {us:8.1f}")
for r in rows:
    if r["Name"].startswith(("gh_", "void gh_")):
        tot += float(r["TotalDurationNs"]) / 1e3 / steps
print(f"  all gh_* kernels: {tot:8.1f} us per step")
PY
  rm -rf $O/stats
done
