"""Copy the summaries written by tools/refresh_profiles.sh <tag> (gpurun_out/<tag>/) into profiles/ (tracked), tagged with the
hash of the sources they were measured on. usage: assemble_profiles.py <tag>   e.g. r3
Per workload W in (default, hd_sh3, hd_sh3_pose32): profiles/<tag>_kernel_stats_<W>.csv, <tag>_pmc_traffic_<W>.json,
<tag>_pmc_fetch_write_<W>.csv, <tag>_pmc_sq_<W>.json, <tag>_pmc_sq_kernels_<W>.csv, <tag>_bench_<W>.json (the default workload's
files carry no suffix beyond the historical names tools/design_table.py reads)."""
import csv, glob, json, os, re, shutil, sys
tag = sys.argv[1]
O = f"gpurun_out/{tag}"
src = open(f"{O}/source_hash.txt").read().strip()
WORKLOADS = {"default": ("two_hands P=98562 512x334 RGB blend, 8 views per launch (BASELINE configs[2])", ""),
             "hd_sh3": ("two_hands_hd P=98562 1024x1024 SH degree 3 blend, 8 views per launch (BASELINE configs[4]'s image / colour shape)",
                        "--config two_hands_hd"),
             "hd_sh3_pose32": ("two_hands_hd P=98562 1024x1024 SH degree 3 blend, 32 different poses per launch (BASELINE configs[4])",
                               "--config two_hands_hd --pose-batch --views-per-step 32")}
# Issue cost of a wave64 VALU instruction in GRBM cycles, measured with 8 waves per SIMD (profiles/<tag>_valu_cycles_pmc.txt):
# fma / min / max / cndmask / cvt / ldexp / DPP ~4.0-4.5, add / mul / sub / and ~2.5, rcp / exp ~8.3. The render kernels are
# made of the first class: their ceiling is one instruction per ~4.2 cycles per SIMD.
PEAK_GUIDE, PEAK_MEASURED = 0.5, 0.24
NOTE = ("Counter arithmetic (MI355X_MICROARCH.md units: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over all SIMDs, GRBM_GUI_ACTIVE "
        "is summed over the 8 XCDs): cycles = GRBM_GUI_ACTIVE / 8; valu_instr_per_cycle_per_simd = SQ_INSTS_VALU / (cycles * 1024 SIMDs). Ceilings: the "
        "guide prices a wave64 VALU instruction at 2 cycles per SIMD (0.5 / cycle); measured in the SAME clock (tools/micro/valu_cycles_pmc.sh: "
        f"GRBM cycles x 1024 / SQ_INSTS_VALU with 8 waves per SIMD, profiles/{tag}_valu_cycles_pmc.txt) v_fma / v_min / v_max / v_cndmask / v_cvt / "
        "v_ldexp / every DPP form cost 4.0-4.5 cycles, v_add / v_mul / v_sub / v_and 2.5, v_rcp / v_exp 8.3, and v_pk_fma_f32 twice a v_fma_f32 — "
        "peak_measured_plain_fp32 = 0.24 / cycle is the first class, which the render kernels are made of. valu_busy_quad = SQ_ACTIVE_INST_VALU * "
        "4 / (cycles * 1024); lds_busy = SQ_ACTIVE_INST_LDS * 4 / (cycles * 256 CUs) (the LDS pipe / crossbar is per CU); waves_per_simd = "
        "SQ_WAVE_CYCLES * 4 / (cycles * 1024); wait fractions are of SQ_WAVE_CYCLES.")


def suffix(w):
    return "" if w == "default" else "_" + w


def kernel_stats(w, desc, args):
    rows = list(csv.DictReader(open(f"{O}/{w}/kernel_stats.csv")))
    out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {args} --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline  (MI355X, "
           f"{desc}; tools/refresh_profiles.sh; source {src})", "name,calls,total_ns,avg_ns,pct,min_ns,max_ns"]
    for r in rows[:36]:
        n = re.sub(r"\(.*", "", r["Name"])[:70]
        out.append(f'{n},{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}')
    name = f"profiles/{tag}_kernel_stats_bench_8views.csv" if w == "default" else f"profiles/{tag}_kernel_stats_{w}.csv"
    open(name, "w").write("\n".join(out) + "\n")


def sq(w, desc, args):
    vals, lines = {}, []
    for f in sorted(glob.glob(f"{O}/{w}/sum_*.csv")):
        for l in open(f):
            if l.startswith("kernel,"):
                continue
            k, c, n, v = l.strip().rsplit(",", 3)
            k = re.sub(r"void |<.*", "", k)
            if k.startswith("gh_"):
                lines.append(l.strip())
                # several instantiations of a template share the name: keep the one with the larger value (the dominant one)
                vals[(k, c)] = max(vals.get((k, c), 0.0), float(v))
    hdr = (f"# rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 bench.py {args} --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline "
           f"--no-stage-timing ; per-dispatch averages ({desc}; tools/refresh_profiles.sh, tools/summarize_pmc.py; source {src})\n"
           "kernel,counter,dispatches,avg_value\n")
    open(f"profiles/{tag}_pmc_sq_kernels{suffix(w)}.csv", "w").write(hdr + "\n".join(sorted(set(lines))) + "\n")
    doc = {"note": NOTE, "source_hash": src, "workload": desc, "kernels": {}}
    for k in sorted({k for k, c in vals if c == "SQ_INSTS_VALU"}):
        try:
            g = vals[(k, "GRBM_GUI_ACTIVE")] / 8
            iv, av, al = vals[(k, "SQ_INSTS_VALU")], vals[(k, "SQ_ACTIVE_INST_VALU")], vals[(k, "SQ_ACTIVE_INST_LDS")]
            wc = vals[(k, "SQ_WAVE_CYCLES")]
        except KeyError:
            continue
        ipc = iv / (g * 1024)
        doc["kernels"][k] = {
            "gpu_cycles": g, "valu_insts_per_launch": iv, "salu_insts_per_launch": vals.get((k, "SQ_INSTS_SALU")),
            "lds_insts_per_launch": vals.get((k, "SQ_INSTS_LDS")), "valu_busy_quad": av * 4 / (g * 1024), "lds_busy": al * 4 / (g * 256),
            "waves_per_simd": wc * 4 / (g * 1024), "wait_any_frac": vals.get((k, "SQ_WAIT_ANY"), 0) / wc,
            "wait_inst_frac": vals.get((k, "SQ_WAIT_INST_ANY"), 0) / wc,
            "secondary": {"bound": "valu", "unit": "wave64 VALU instructions / cycle / SIMD", "achieved": ipc, "peak": PEAK_GUIDE,
                          "frac": ipc / PEAK_GUIDE, "peak_measured_plain_fp32": PEAK_MEASURED, "frac_of_measured": ipc / PEAK_MEASURED,
                          "lds": {"bound": "lds", "unit": "busy fraction of the per-CU LDS pipe", "achieved": al * 4 / (g * 256), "peak": 1.0,
                                  "frac": al * 4 / (g * 256)}}}
    json.dump(doc, open(f"profiles/{tag}_pmc_sq{suffix(w)}.json", "w"), indent=1)
    return doc


for w, (desc, args) in WORKLOADS.items():
    if not os.path.exists(f"{O}/{w}/kernel_stats.csv"):
        continue
    kernel_stats(w, desc, args)
    tr = json.load(open(f"{O}/{w}/pmc_traffic.json"))
    tr["source_hash"], tr["workload"] = src, desc
    json.dump(tr, open(f"profiles/{tag}_pmc_traffic{suffix(w)}.json", "w"), indent=1)
    shutil.copy(f"{O}/{w}/pmc_fetch_write.csv", f"profiles/{tag}_pmc_fetch_write_8views.csv" if w == "default" else f"profiles/{tag}_pmc_fetch_write_{w}.csv")
    doc = sq(w, desc, args)
    for k in ("gh_render_fwd_kernel", "gh_render_bwd_kernel"):
        if k in doc["kernels"]:
            e = doc["kernels"][k]
            print(w, k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a != "secondary"},
                  round(e["secondary"]["frac"], 3), round(e["secondary"]["frac_of_measured"], 3))
for n in ("bench_default", "bench_1view", "bench_2view", "bench_4view", "bench_16view", "bench_split", "bench_hd_sh3", "bench_hd_sh3_split", "bench_hd_sh3_pose32", "bench_hd_sh3_pose32_split", "bench_random1k"):
    if os.path.exists(f"{O}/{n}.json") and os.path.getsize(f"{O}/{n}.json") > 0:
        shutil.copy(f"{O}/{n}.json", f"profiles/{tag}_{n}.json")
for n in ("two_call_cost.txt", "valu_rate_wallclock.txt", "valu_cycles_pmc.txt", "dropin_host_breakdown.txt", "fit_step_profile.txt", "fit_step_views.txt",
          "kernel_floor.txt", "fetch_calib.txt", "timeline_8view.txt", "timeline_1view.txt"):
    if os.path.exists(f"{O}/{n}") and os.path.getsize(f"{O}/{n}") > 0:
        shutil.copy(f"{O}/{n}", f"profiles/{tag}_{n}")
