# Kernel-to-kernel gaps of the captured step: rocprofv3 kernel trace of the default bench, then per replayed step the sum of kernel
# durations against the span from the first kernel's start to the last kernel's end (usage: bash tools/gap_trace.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/gaps
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o tr --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-stage-timing > $O/bench.json 2> $O/log.txt || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gaps/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps: split at gh_preprocess_fwd_kernel; keep the last 10 complete ones
idx = [i for i, k in enumerate(ks) if k[0].startswith("gh_preprocess_fwd_kernel")]
out = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = ks[a:b]
    span = seg[-1][2] - seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    gaps = [(seg[i + 1][1] - seg[i][2], seg[i][0], seg[i + 1][0]) for i in range(len(seg) - 1)]
    out.append((span, busy, len(seg), gaps, seg[0][1], (ks[b][1] - seg[-1][2])))
for span, busy, n, gaps, t0, nxt in out[-6:]:
    print(f"step: {n} kernels, span {span/1e3:.1f} us, sum of durations {busy/1e3:.1f} us, gaps {sum(g for g,_,_ in gaps)/1e3:.1f} us, gap to next step {nxt/1e3:.1f} us")
span, busy, n, gaps, t0, nxt = out[-2]
for g, a, b in gaps:
    print(f"  {g/1e3:6.2f} us  {a} -> {b}")
PY
rm -rf $O/tr
