"""The measured tables of README.md / DESIGN.md are GENERATED from the committed profiles (tools/gen_kernel_table.py): this test fails when
someone edits the numbers by hand or commits new profiles without regenerating ("docs drift")."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_kernel_table", os.path.join(ROOT, "tools", "gen_kernel_table.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _block(path, name):
    s = open(os.path.join(ROOT, path)).read()
    a, b = f"<!-- BEGIN GENERATED {name} -->", f"<!-- END GENERATED {name} -->"
    assert a in s and b in s, (path, name)
    return s[s.index(a) + len(a):s.index(b)].strip()


def test_generated_tables_match_the_committed_profiles():
    g = _gen()
    tag = "r05_z"
    assert os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_bench_c3.json"))
    assert _block("README.md", "headline") == g.headline(tag).strip()
    assert _block("DESIGN.md", "headline") == g.headline(tag).strip()
    assert _block("DESIGN.md", "kernel-table") == g.kernel_table(tag).strip()


def test_the_bench_lines_in_profiles_carry_counters_of_their_own_kernel_sources():
    """`roofline.traffic` of the committed headline line must come from counters taken at the kernel sources the line was measured on."""
    import json
    d = json.loads(open(os.path.join(ROOT, "profiles", "r05_z_bench_c3.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    assert r["traffic"] and r["traffic_source"]["stale"] is False
    pm = json.load(open(os.path.join(ROOT, "profiles", "r05_z_hbm_traffic_pmc.json")))
    assert r["traffic_source"].get("csrc_sha", pm["_meta"]["csrc_sha"]) == pm["_meta"]["csrc_sha"]
