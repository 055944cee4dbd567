"""The measured-numbers block of DESIGN.md §5, generated from profiles/ so that it cannot drift:
    python tools/design_table.py            prints the block
    python tools/design_table.py --write    replaces the text between the GENERATED markers in DESIGN.md
tests/test_docs.py asserts DESIGN.md holds exactly this block for the committed profiles."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r6"
BEGIN, END = "<!-- BEGIN GENERATED: tools/design_table.py -->", "<!-- END GENERATED -->"


def block() -> str:
    P = lambda n: os.path.join(ROOT, "profiles", f"{TAG}_{n}")
    b = json.load(open(P("bench_default.json")))
    tr = json.load(open(P("pmc_traffic.json")))
    sq = json.load(open(P("pmc_sq.json")))
    stats = {}
    for l in open(P("kernel_stats_bench_8views.csv")):
        if l.startswith("#") or l.startswith("name,"):
            continue
        name, calls, total_ns = l.strip().rsplit(",", 6)[:3]          # template names hold commas
        k = re.sub(r"void |<.*", "", name)
        c, t = stats.get(k, (0, 0.0))
        stats[k] = (c + int(calls), t + float(total_ns))
    # Steps a kernel ran in: the bench renders a few forward-only images (the target, the D_rect count) besides its steps, so the
    # forward kernels have more calls than the backward ones — every kernel is divided by ITS OWN step count.
    fwd_steps, bwd_steps = stats["gh_preprocess_fwd_kernel"][0], stats["gh_render_bwd_kernel"][0]

    def steps_of(calls):
        return bwd_steps if (calls % bwd_steps == 0 and calls % fwd_steps != 0) else fwd_steps
    L = [BEGIN,
         f"Generated from `profiles/{TAG}_bench_default.json`, `{TAG}_kernel_stats_bench_8views.csv`, `{TAG}_pmc_traffic.json`, "
         f"`{TAG}_pmc_sq.json` (sources `{tr['source_hash']}`): default bench line **{b['value']:.0f} renders/s, {b['ms_per_step']:.3f} ms per step** "
         f"(windows min / median {b['config']['repeats']['ms_per_step_min']:.3f} / {b['config']['repeats']['ms_per_step_median']:.3f}), "
         f"roofline of `{b['roofline']['kernel']}`: {b['roofline']['achieved']:.0f} GB/s algorithmic = {b['roofline']['frac']:.3f} of 8 TB/s.", "",
         "| stage (HIP events, `bench.py`) | ms | algorithmic GB/s |", "|---|---|---|"]
    for k, v in b["stages"].items():
        L.append(f"| {k} | {v['ms']:.3f} | {v['alg_GBs']:.0f} |")
    L += ["", "| kernel (rocprofv3, per step) | launches/step | µs/step | PMC traffic MB/launch | VALU instr/cycle/SIMD (of the guide's 0.5 / of the measured 0.24) | LDS pipe busy |",
          "|---|---|---|---|---|---|"]
    for k, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if not k.startswith("gh_"):
            continue
        trk = tr["kernels"].get(k)
        s = sq["kernels"].get(k)
        traffic = f"{trk['traffic_bytes'] / 1e6:.0f}" if trk else "—"
        if s:
            sec = s["secondary"]
            valu = f"{sec['achieved']:.3f} ({sec['frac']:.2f} / {sec['frac_of_measured']:.2f})"
            lds = f"{s['lds_busy']:.2f}"
        else:
            valu, lds = "—", "—"
        L.append(f"| `{k}` | {c / steps_of(c):.0f} | {t / steps_of(c) / 1e3:.1f} | {traffic} | {valu} | {lds} |")
    L.append(END)
    return "\n".join(L)


BEGIN7, END7 = "<!-- BEGIN GENERATED NUMBERS: tools/design_table.py -->", "<!-- END GENERATED NUMBERS -->"


def numbers() -> str:
    """The configuration table of DESIGN.md §7, from the committed bench lines and fit-step timings."""
    B = lambda n: json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench_{n}.json")))
    d, sp, v1, v2, v4, v16, hd, pb, rk = (B(n) for n in ("default", "split", "1view", "2view", "4view", "16view", "hd_sh3", "hd_sh3_pose32", "random1k"))
    med = lambda x: x["config"]["step_ms"]["median"]
    fit = {}
    for f in ("fit_step_profile.txt", "fit_step_views.txt"):
        for l in open(os.path.join(ROOT, "profiles", f"{TAG}_{f}")):
            m = re.match(r"fit step, (\d+) views, .*static_geometry=(\w+): eager ([\d.]+) ms, captured graph ([\d.]+) ms", l)
            if m:
                fit[(int(m.group(1)), m.group(2) == "True")] = float(m.group(4))
    two8, two1 = (re.search(r": ([\d.]+) ms per view", l).group(1) for l in open(os.path.join(ROOT, "profiles", f"{TAG}_two_call_cost.txt")).readlines()[:2])
    st = lambda x: " / ".join(f"{x['stages'][k]['ms']:.3f}" for k in ("preprocess_fwd", "binning", "render_fwd", "render_bwd", "preprocess_bwd"))
    L = [BEGIN7,
         f"Round-6 numbers (MI355X, 1 GPU; `profiles/{TAG}_*`; window 1 of the bench line, per-step median in brackets):", "",
         "| configuration | renders/s | ms/step | round 5 |", "|---|---|---|---|",
         f"| two hands, 8 views/step (default, BASELINE configs[2]/[3] shape), graph replay | **{d['value']:,.0f}** | {d['ms_per_step']:.3f} ({med(d):.3f}; p10 / p90 {d['config']['step_ms']['p10']:.3f} / {d['config']['step_ms']['p90']:.3f}) | 12,523 / 0.639 |",
         f"| same with `--split-streams on` (two halves of the views on two streams, bit-identical) | {sp['value']:,.0f} | {sp['ms_per_step']:.3f} ({med(sp):.3f}) | 12,206 / 0.655 |",
         f"| two hands, 1 view/step (the reference's own batch shape) | {v1['value']:,.0f} | {v1['ms_per_step']:.3f} | 4,694 / 0.213 |",
         f"| two hands, 2 / 4 / 16 views/step | {v2['value']:,.0f} / {v4['value']:,.0f} / {v16['value']:,.0f} | {v2['ms_per_step']:.3f} / {v4['ms_per_step']:.3f} / {v16['ms_per_step']:.3f} | 6,725 / 9,387 / 13,876 |",
         f"| two hands 1024², SH degree 3 (configs[4] shape), 8 views/step; dominant kernel `{hd['roofline']['kernel']}` {hd['roofline']['frac']:.3f} of 8 TB/s, counter traffic {hd['roofline']['traffic'] / 1e6:.0f} MB | {hd['value']:,.0f} | {hd['ms_per_step']:.2f} | 5,226 / 1.53 |",
         f"| same with `--split-streams on` (bit-identical) | {B('hd_sh3_split')['value']:,.0f} | {B('hd_sh3_split')['ms_per_step']:.2f} | 5,714 / 1.40 |",
         f"| same, **32 different poses per step** (configs[4]'s batch: `--pose-batch --views-per-step 32`); dominant kernel `{pb['roofline']['kernel']}` {pb['roofline']['frac']:.3f} of 8 TB/s, counter traffic {pb['roofline']['traffic'] / 1e9:.2f} GB | {pb['value']:,.0f} | {pb['ms_per_step']:.2f} | 4,553 / 7.03 |",
         f"| same with `--split-streams on` (bit-identical) | {B('hd_sh3_pose32_split')['value']:,.0f} | {B('hd_sh3_pose32_split')['ms_per_step']:.2f} | 4,764 / 6.72 |",
         f"| 1k random Gaussians, 128², 1 view (configs[0]) | {rk['value']:,.0f} | {rk['ms_per_step']:.3f} | 10,960 / 0.091 |",
         f"| reference protocol through the drop-in (2 rasteriser calls per view, one loss and one backward per step, every leaf), per view fwd+bwd: 8 views per step / 1 view per step (`tools/two_call_cost.py` = `bench.two_call_cost`, the bench line's own figures: {d['config']['two_call_ms_per_view']:.2f} / {d['config']['two_call_ms_per_view_one_view_steps']:.2f}; `profiles/{TAG}_two_call_reconcile.txt`) | — | {two8} / {two1} (host-bound) | two harnesses, two protocols: 0.714 / 1.11 |",
         f"| full-size one-shot fit step, 8 views (fused α, 1024×2048 maps, active texels, static geometry, captured graph; `profiles/{TAG}_fit_step_profile.txt`) | — | **{fit[(8, True)]:.2f}** (full path {fit[(8, False)]:.2f}) | 0.51 (full path 0.67) |",
         f"| same, 4 / 2 / 1 view(s) (configs[3] on 2 / 4 / 8 GPUs; `profiles/{TAG}_fit_step_views.txt`) | — | {fit[(4, True)]:.2f} / {fit[(2, True)]:.2f} / **{fit[(1, True)]:.2f}** | 0.34 / 0.25 / 0.18 (full path 0.46 / 0.34 / 0.25); this round's full path: {fit[(4, False)]:.2f} / {fit[(2, False)]:.2f} / {fit[(1, False)]:.2f} |",
         f"| CPU oracle in its all-core baseline mode, {d['cpu_baseline']['threads']} threads ({d['cpu_baseline']['cpu_model']}), serial fraction {d['cpu_baseline']['serial_fraction']:.4f}, configs[2] | {d['cpu_baseline']['value']:.1f} | — | 11.2 |",
         f"| PyTorch CPU autograd (dense `oracle_torch`, {rk['cpu_baseline']['cores']} threads), configs[0] | {rk['cpu_baseline']['value']:.3f} | — | — |", "",
         f"Stage times (ms; projection / binning / render forward / render backward / per-Gaussian backward) at 8 views: {st(d)}; at 1 view: {st(v1)} — the render",
         "kernels are as long as their heaviest waves (§5: launch order by measured work, the heaviest tiles in the fine-grained form), binning is the",
         f"latency of 12 small kernels (profiles/r6_timeline_1view.txt); at 1024² SH3 × 8 views: {st(hd)}; 32 poses: {st(pb)}.",
         END7]
    return "\n".join(L)


if __name__ == "__main__":
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, z = s.index(BEGIN), s.index(END) + len(END)
        s = s[:a] + block() + s[z:]
        a, z = s.index(BEGIN7), s.index(END7) + len(END7)
        s = s[:a] + numbers() + s[z:]
        open(p, "w").write(s)
    else:
        print(block())
        print(numbers())
