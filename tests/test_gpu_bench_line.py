"""The driver's contract for bench.py: ONE JSON line on stdout with the metric, the roofline object (path fraction, the
dominant kernel with a bound that can be approached, frac <= 1) and the CPU baseline.  Run as the driver runs it (a child
process), on the small workload C2 so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench("--workload", "C2", "--steps", "3", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "cells/s" and d["value"] > 1e10 and "workload" in d["config"] and "model" not in d["config"]
    # value = cells of one step / ms_per_step
    assert abs(d["value"] - 2500 * 100000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    dk = r["dominant_kernel"]
    assert dk["avg_launch_ms"] > 0 and "algorithmic_frac" in dk
    # the bound that can be approached comes from the PMC summary committed for THESE kernel sources; absent one, it says so
    if dk.get("bound") is None:
        assert "no PMC issue summary" in dk.get("note", "")
    else:
        assert dk["bound"] in ("valu_issue", "hbm") and 0.0 < dk["frac"] <= 1.0 and 1.5 < dk["clock_ghz"] < 2.6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "cells/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


def test_headline_line_carries_the_other_workloads():
    """Without --workload the N = 1 line is the headline C3 and measures C2, C5 and C4 beside it."""
    d = _bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert "C3" in d["config"]["workload"]
    ow = d["config"]["other_workloads"]
    assert set(ow) == {"C2", "C5", "C4"}
    for name, v in ow.items():
        assert v["value"] > 1e10 and v["ms_per_step"] > 0 and set(v["phases_ms"]) >= {"ms_phase_a", "ms_phase_c", "ms_dp", "ms_pass2"}, name
    assert d["roofline"]["dominant_kernel"].get("frac") is None or d["roofline"]["dominant_kernel"]["frac"] <= 1.0
