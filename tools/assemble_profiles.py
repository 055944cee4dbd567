"""Copy the summaries written by tools/refresh_profiles.sh <tag> (gpurun_out/<tag>/) into profiles/ (tracked).
usage: assemble_profiles.py <tag>   e.g. r1h"""
import csv, glob, json, re, shutil, sys
tag = sys.argv[1]
O = f"gpurun_out/{tag}"
shutil.copy(f"{O}/pmc_traffic.json", "profiles/pmc_traffic.json")
shutil.copy(f"{O}/pmc_fetch_write_8views.csv", f"profiles/{tag}_pmc_fetch_write_8views.csv")
shutil.copy(f"{O}/bench_default.json", f"profiles/{tag}_bench_default.json")
rows = list(csv.DictReader(open(f"{O}/kernel_stats.csv")))
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline  (MI355X, tools/prof_stats.sh)",
       "name,calls,total_ns,avg_ns,pct,min_ns,max_ns"]
for r in rows[:34]:
    n = re.sub(r"\(.*", "", r["Name"])[:70]
    out.append(f'{n},{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}')
open(f"profiles/{tag}_kernel_stats_bench_8views.csv", "w").write("\n".join(out) + "\n")
vals, lines = {}, []
for f in sorted(glob.glob(f"{O}/sum_*.csv")):
    for l in open(f):
        if "render" in l:
            lines.append(l.strip())
            k, c, n, v = l.strip().rsplit(",", 3)
            vals[(re.sub(r"void |<.*", "", k), c)] = float(v)
hdr = ("# rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "
       "--no-stage-timing ; per-dispatch averages (tools/pmc_sq.sh, tools/summarize_pmc.py)\nkernel,counter,dispatches,avg_value\n")
open(f"profiles/{tag}_pmc_sq_render_kernels.csv", "w").write(hdr + "\n".join(sorted(set(lines))) + "\n")
sq = {"note": "VALU busy fraction = SQ_ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): SQ_ACTIVE_INST_VALU counts quad-cycles "
              f"summed over all SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs (profiles/{tag}_pmc_sq_render_kernels.csv)",
      "workload": "two_hands P=98562 512x334 RGB blend, 8 views per launch", "kernels": {}}
for k in ("gh_render_fwd_kernel", "gh_render_bwd_kernel"):
    a, g = vals[(k, "SQ_ACTIVE_INST_VALU")], vals[(k, "GRBM_GUI_ACTIVE")]
    sq["kernels"][k] = {"valu_busy_frac": a * 4 / (g / 8 * 1024), "valu_insts_per_launch": vals[(k, "SQ_INSTS_VALU")],
                        "lds_insts_per_launch": vals[(k, "SQ_INSTS_LDS")],
                        "gpu_cycles": g / 8}
json.dump(sq, open("profiles/pmc_sq.json", "w"), indent=1)
print(json.dumps(sq["kernels"], indent=1))
print("\n".join(out[:16]))
