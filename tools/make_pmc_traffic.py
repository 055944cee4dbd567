"""Build profiles/pmc_traffic.json + the per-round csv from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: make_pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out json> <out csv>"""
import collections, csv, json, re, sys

fetch_csv, write_csv, out_json, out_csv = sys.argv[1:5]
workload = sys.argv[5] if len(sys.argv) > 5 else "default: bench.py"


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"<.*|\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    return {k: s / n for k, (s, n) in acc.items()}


fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py <workload args> --steps 3 "
        "--warmup 2 --no-cpu-baseline --no-stage-timing`; KB per dispatch; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
        "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts wide streaming reads at half)")
kern = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, 0) + wr.get(k, 0))):
    if not k.startswith("gh_"):
        continue
    kern[k] = {"fetch_size_kb": fe.get(k, 0.0), "write_size_kb": wr.get(k, 0.0),
               "traffic_bytes": (2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024}
json.dump({"note": note, "workload": workload, "kernels": kern},
          open(out_json, "w"), indent=1)
with open(out_csv, "w") as f:
    f.write("# " + note + "\nkernel,fetch_size_kb,write_size_kb,traffic_bytes\n")
    for k, v in kern.items():
        f.write(f"{k},{v['fetch_size_kb']:.1f},{v['write_size_kb']:.1f},{v['traffic_bytes']:.0f}\n")
print(open(out_csv).read())
