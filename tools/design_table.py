"""The measured-numbers block of DESIGN.md §5, generated from profiles/ so that it cannot drift:
    python tools/design_table.py            prints the block
    python tools/design_table.py --write    replaces the text between the GENERATED markers in DESIGN.md
tests/test_docs.py asserts DESIGN.md holds exactly this block for the committed profiles."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r3"
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


if __name__ == "__main__":
    text = block()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, z = s.index(BEGIN), s.index(END) + len(END)
        open(p, "w").write(s[:a] + text + s[z:])
    else:
        print(text)
