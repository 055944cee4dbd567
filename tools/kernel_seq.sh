# Per-launch durations of one kernel, in launch order (usage: bash tools/kernel_seq.sh <kernel substring> [bench args])
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=$1; shift
O=gpurun_out/kseq
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o tr --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline "$@" > $O/bench.json 2> $O/log.txt || exit 1
python3 - "$K" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kseq/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    if sys.argv[1] in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:8.1f} us   after {prev[:50] if prev else '-'}")
    prev = r["Kernel_Name"]
PY
rm -rf $O/tr
