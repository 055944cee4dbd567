"""DESIGN.md's measured-numbers block is generated from profiles/ (tools/design_table.py): the committed text must be
exactly what the committed profiles produce, and the profiles must all come from one build of the sources."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_numbers_are_the_profiles():
    import design_table as T
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, z = s.index(T.BEGIN), s.index(T.END) + len(T.END)
    assert s[a:z] == T.block(), "run `python tools/design_table.py --write` after refreshing profiles/"
    a, z = s.index(T.BEGIN7), s.index(T.END7) + len(T.END7)
    assert s[a:z] == T.numbers(), "run `python tools/design_table.py --write` after refreshing profiles/"


def test_profiles_share_one_source_hash():
    tr = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_traffic.json")))
    sq = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_sq.json")))
    head = open(os.path.join(ROOT, "profiles", "r6_kernel_stats_bench_8views.csv")).readline()
    assert tr["source_hash"] == sq["source_hash"] and tr["source_hash"] in head
    b = json.load(open(os.path.join(ROOT, "profiles", "r6_bench_default.json")))
    # the committed bench line quotes counters of its own build (null + the reason otherwise)
    assert b["roofline"]["traffic"] is not None and tr["source_hash"] in b["roofline"]["traffic_source"]


def test_profiles_were_measured_on_the_committed_sources():
    """A source change after the last `tools/refresh_profiles.sh` leaves bench.py without counters to quote (it reports
    `traffic: null` and why): the committed profiles must carry the hash of the committed HIP sources + C header."""
    sys.path.insert(0, ROOT)
    import bench
    tr = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_traffic.json")))
    assert tr["source_hash"] == bench.source_hash(), "sources changed since profiles/ were refreshed: bash tools/refresh_profiles.sh r6"
