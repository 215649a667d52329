"""DESIGN.md's table of measured numbers is generated from the committed bench line (tools/design_table.py): this fails when
someone edits one without the other, and checks the line the table comes from for the keys the round-5 protocol promises."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06_bench_driver.json")


def test_design_table_matches_the_committed_bench_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"], cwd=ROOT)
    assert p.returncode == 0, "DESIGN.md's generated table is stale: run `python tools/design_table.py`"


def test_committed_bench_line_is_self_verifying_and_ends_in_its_summary():
    d = json.load(open(LINE))
    assert list(d)[-1] == "summary" and d["summary"]["parity_all_ok"] is True
    assert d["metric"] == "agent-steps/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "u8"
    rows = [d] + d["configs"]
    assert [r["config"] for r in d["configs"]] == ["C2", "C3", "C5", "C1"]
    for r in rows:
        par = r["parity_in_run"]
        assert par["ok"] is True and par["per_step_steps"] > 0 and par["fused_steps"] > 0 and "oracle" in par["checker"]
        roof = r["roofline"]
        assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
        committed = roof.get("traffic_committed_source") or roof["traffic_source"]  # (the headline's traffic is measured in the run since round 6)
        assert roof["traffic_ratio"] > 0 and "profiles/traffic.json @" in committed
        assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] >= 1 and "_oracle_state" not in r["cpu_baseline"]
    # the PMC constant behind `traffic` names the commit and kernel-source hash it was measured on (profiles/traffic.json)
    prov = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["_provenance"]
    assert len(prov["git_head"]) == 40 and len(prov["kernels_sha16"]) == 16 and prov["git_head"][:12] in (d["roofline"].get("traffic_committed_source") or d["roofline"]["traffic_source"])
