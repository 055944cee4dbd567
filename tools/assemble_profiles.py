"""Copy the summaries written by tools/refresh_profiles.sh <tag> (gpurun_out/<tag>/) into profiles/ (tracked), tagged with the
hash of the sources they were measured on. usage: assemble_profiles.py <tag>   e.g. r2"""
import csv, glob, json, os, re, shutil, sys
tag = sys.argv[1]
O = f"gpurun_out/{tag}"
src = open(f"{O}/source_hash.txt").read().strip()
tr = json.load(open(f"{O}/pmc_traffic.json"))
tr["source_hash"] = src
json.dump(tr, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
shutil.copy(f"{O}/pmc_fetch_write_8views.csv", f"profiles/{tag}_pmc_fetch_write_8views.csv")
for n in ("bench_default", "bench_1view", "bench_two_hands_hd_sh3", "bench_two_hands_hd_sh3_pose_batch32"):
    if os.path.exists(f"{O}/{n}.json"):
        shutil.copy(f"{O}/{n}.json", f"profiles/{tag}_{n}.json")
for n in ("two_call_cost.txt", "valu_rate.txt", "dropin_host_breakdown.txt", "fit_step_profile.txt"):
    if os.path.exists(f"{O}/{n}") and os.path.getsize(f"{O}/{n}") > 0:
        shutil.copy(f"{O}/{n}", f"profiles/{tag}_{n}")
rows = list(csv.DictReader(open(f"{O}/kernel_stats.csv")))
out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline  (MI355X, tools/refresh_profiles.sh; source {src})",
       "name,calls,total_ns,avg_ns,pct,min_ns,max_ns"]
for r in rows[:34]:
    n = re.sub(r"\(.*", "", r["Name"])[:70]
    out.append(f'{n},{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}')
open(f"profiles/{tag}_kernel_stats_bench_8views.csv", "w").write("\n".join(out) + "\n")
if os.path.exists(f"{O}/kernel_stats_hd.csv"):
    rows_hd = list(csv.DictReader(open(f"{O}/kernel_stats_hd.csv")))
    out_hd = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --config two_hands_hd --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline  (MI355X, 1024x1024, SH degree 3, 8 views; source {src})",
              "name,calls,total_ns,avg_ns,pct,min_ns,max_ns"]
    for r in rows_hd[:24]:
        n = re.sub(r"\(.*", "", r["Name"])[:70]
        out_hd.append(f'{n},{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}')
    open(f"profiles/{tag}_kernel_stats_hd_sh3_8views.csv", "w").write("\n".join(out_hd) + "\n")
vals, lines = {}, []
for f in sorted(glob.glob(f"{O}/sum_*.csv")):
    for l in open(f):
        if l.startswith("kernel,"):
            continue
        k, c, n, v = l.strip().rsplit(",", 3)
        k = re.sub(r"void |<.*", "", k)
        if k.startswith("gh_"):
            lines.append(l.strip())
            vals[(k, c)] = float(v)
hdr = (f"# rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline "
       f"--no-stage-timing ; per-dispatch averages (tools/refresh_profiles.sh, tools/summarize_pmc.py; source {src})\nkernel,counter,dispatches,avg_value\n")
open(f"profiles/{tag}_pmc_sq_kernels.csv", "w").write(hdr + "\n".join(sorted(set(lines))) + "\n")
NOTE = ("Counter arithmetic (MI355X_MICROARCH.md units: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over all SIMDs, GRBM_GUI_ACTIVE "
        "is summed over the 8 XCDs): cycles = GRBM_GUI_ACTIVE / 8; valu_instr_per_cycle_per_simd = SQ_INSTS_VALU / (cycles * 1024 SIMDs); the guide's "
        "peak is one wave64 VALU instruction per 2 cycles per SIMD (0.5), tools/micro/valu_rate.hip measures one plain fp32 op per ~4 cycles "
        f"(0.25: profiles/{tag}_valu_rate.txt) — both fractions are given; valu_busy_quad = SQ_ACTIVE_INST_VALU * 4 / (cycles * 1024); "
        "lds_busy = SQ_ACTIVE_INST_LDS * 4 / (cycles * 256 CUs) (the LDS pipe / crossbar is per CU); waves_per_simd = SQ_WAVE_CYCLES * 4 / "
        "(cycles * 1024); wait fractions are of SQ_WAVE_CYCLES.")
sq = {"note": NOTE, "source_hash": src, "workload": "two_hands P=98562 512x334 RGB blend, 8 views per launch", "kernels": {}}
kernels = sorted({k for k, c in vals if c == "SQ_INSTS_VALU"})
for k in kernels:
    try:
        g = vals[(k, "GRBM_GUI_ACTIVE")] / 8
        iv, av, al = vals[(k, "SQ_INSTS_VALU")], vals[(k, "SQ_ACTIVE_INST_VALU")], vals[(k, "SQ_ACTIVE_INST_LDS")]
        wc = vals[(k, "SQ_WAVE_CYCLES")]
    except KeyError:
        continue
    ipc = iv / (g * 1024)
    ent = {"gpu_cycles": g, "valu_insts_per_launch": iv, "salu_insts_per_launch": vals.get((k, "SQ_INSTS_SALU")),
           "lds_insts_per_launch": vals.get((k, "SQ_INSTS_LDS")), "valu_busy_quad": av * 4 / (g * 1024), "lds_busy": al * 4 / (g * 256),
           "waves_per_simd": wc * 4 / (g * 1024), "wait_any_frac": vals.get((k, "SQ_WAIT_ANY"), 0) / wc,
           "wait_inst_frac": vals.get((k, "SQ_WAIT_INST_ANY"), 0) / wc,
           "secondary": {"bound": "valu", "unit": "wave64 VALU instructions / cycle / SIMD", "achieved": ipc, "peak": 0.5, "frac": ipc / 0.5,
                         "peak_measured_plain_fp32": 0.25, "frac_of_measured": ipc / 0.25,
                         "lds": {"bound": "lds", "unit": "busy fraction of the per-CU LDS pipe", "achieved": al * 4 / (g * 256), "peak": 1.0,
                                 "frac": al * 4 / (g * 256)}}}
    sq["kernels"][k] = ent
json.dump(sq, open(f"profiles/{tag}_pmc_sq.json", "w"), indent=1)
for k in ("gh_render_fwd_kernel", "gh_render_bwd_kernel"):
    if k in sq["kernels"]:
        e = sq["kernels"][k]
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a != "secondary"}, e["secondary"]["frac"], e["secondary"]["frac_of_measured"])
print("\n".join(out[:18]))
