"""The driver's contract for bench.py: ONE JSON line on stdout with the metric, the roofline object (the path's accounting
figure, the HBM traffic the PMC passes counted, the dominant phase with a bound that can be approached, frac <= 1), the
cold first run and the CPU baseline.  Run as the driver runs it (a child process), on the small workload C2 so that it takes
seconds; the headline test runs the driver's plain command (C4 and, in child processes, C2, C3, C5)."""
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
    # (the path figure is SURVEY 8(d)'s accounting against a column-at-a-time HBM design: it may exceed 1 since a column updates
    # the blocks' representative rows only; what is bounded by 1 is the dominant phase's fraction and the measured HBM share)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["frac"] > 0.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "hbm_frac_measured" in r and (r["hbm_frac_measured"] is None or 0.0 < r["hbm_frac_measured"] <= 1.0)
    assert d["config"]["cold_first_run_ms"] >= d["ms_per_step"] * 0.5 and "reduced_rows_mean" in d["config"]
    dk = r["dominant_kernel"]
    assert dk["avg_launch_ms"] > 0 and "algorithmic_frac" in dk
    # the bound that can be approached comes from the PMC summary committed for THESE kernel sources; absent one, it says so
    if dk.get("bound") is None:
        assert "no PMC phase summary" in dk.get("note", "")
    else:
        assert dk["bound"] in ("valu_issue", "hbm") and 0.0 < dk["frac"] <= 1.0 and 1.5 < dk["clock_ghz"] < 2.6
        assert dk["lane_instructions_per_cell"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "cells/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


def test_headline_line_carries_the_other_workloads():
    """Without --workload the N = 1 line is the headline C4 -- the configuration the metric is quoted on -- and measures C2,
    C3 and C5 beside it (child processes)."""
    d = _bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert "C4" in d["config"]["workload"] and d["steps"] == 3
    assert abs(d["value"] - 100000 * 5000000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["config"]["cold_first_run_ms"] > 0 and d["config"]["reduced_blocks"] > 0
    ow = d["config"]["other_workloads"]
    assert set(ow) == {"C2", "C3", "C5"}
    for name, v in ow.items():
        assert "error" not in v, (name, v)
        assert v["value"] > 1e10 and v["ms_per_step"] > 0 and set(v["phases_ms"]) >= {"ms_phase_a", "ms_phase_c", "ms_dp", "ms_pass2"}, name
        assert v["cold_first_run_ms"] > 0
    assert d["roofline"]["dominant_kernel"].get("frac") is None or d["roofline"]["dominant_kernel"]["frac"] <= 1.0
