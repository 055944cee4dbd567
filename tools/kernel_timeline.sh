# Timeline of ONE steady-state step: every kernel in launch order with its start offset, duration and the gap to the previous
# kernel's end (usage: bash tools/kernel_timeline.sh [bench args]) — what launch boundaries cost inside a replayed graph.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ktl
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o tr --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-stage-timing "$@" > $O/bench.json 2> $O/log.txt || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktl/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps begin at the projection kernel; take the median-length one of the last 8 complete steps
starts = [i for i, r in enumerate(rows) if "gh_preprocess_fwd_kernel" in r["Kernel_Name"]]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-8:]
steps.sort(key=lambda ab: int(rows[ab[1] - 1]["End_Timestamp"]) - int(rows[ab[0]]["Start_Timestamp"]))
a, b = steps[len(steps) // 2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; busy = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}   {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"step {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us, {b - a} launches")
PY
rm -rf $O/tr
