"""DESIGN.md quotes the committed profiles: its measured table is written by tools/assemble_design.py from profiles/r05_*, and
this test regenerates it - a number edited by hand in the text, or a profile replaced without regenerating the text, fails here."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_design_md_is_what_the_profiles_say():
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "assemble_design.py"), str(ROOT / "tools" / "DESIGN.in.md")], capture_output=True, text=True, check=True).stdout
    assert out == (ROOT / "DESIGN.md").read_text()


def test_bench_line_of_the_round_carries_the_contract_fields():
    import json

    d = json.loads((ROOT / "profiles" / "r05_bench_default.json").read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] and r["traffic"] > r["alg_bytes_per_launch"]  # PMC traffic measured on these kernel sources
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
