#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of profiles/collect_profiles.sh into the committed summaries:

    python profiles/summarize_pmc.py gpurun_out/prof_<tag> r02_<tag> <workload> [steps+warmup]

writes profiles/<prefix>_kernel_stats_<workload>.csv, _pmc_*.csv (per-kernel averages of every counter pass
found), _bench_<workload>.json and _pmc_traffic_<workload>.json: HBM bytes per launch and kernel, and per step
and phase (FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read
stream -- MI355X_MICROARCH.md, HBM section -- so reads are doubled; calibrated in round 1 on k_synth /
k_colblock<MODE_RANK>, whose traffic is known).  The JSON records the hash of the kernel sources it was taken
on; bench.py quotes it only while that hash matches."""
import csv
import glob
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def find(pattern):
    return sorted(glob.glob(pattern, recursive=True))


def per_kernel(path):
    """{kernel: {counter: (sum, launches, summed dispatch duration in ns)}} from a rocprofv3 counter_collection.csv.
    [r5] k_columns_red runs twice per step under one name: with the lists as its output (phase C) and with the class tables
    (pass 2's sweeps, behind the DP); the dispatches behind a k_dp / k_tb_* launch and in front of the next phase A get the
    suffix " #tables" so that the two are accounted apart."""
    acc = {}
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    disp = {}
    for row in rows:
        disp.setdefault(int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0])
    behind_dp = False
    suffix = {}
    for did in sorted(disp):
        nm = disp[did]
        if "k_blockkeys" in nm or "k_blocktrie" in nm or "k_colblock<" in nm and ", 0," in nm or "k_synth" in nm:
            behind_dp = False
        if "k_dp<" in nm or "k_tb_" in nm:
            behind_dp = True
        suffix[did] = " #tables" if (behind_dp and "k_columns_red" in nm) else ""
    for row in rows:
            name = row["Kernel_Name"].split("(")[0] + suffix[int(row["Dispatch_Id"])]
            a = acc.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0, 0.0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            try:
                a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    return acc


def kernel_avg_ns(path):
    """{kernel: average duration in ns} from a rocprofv3 kernel_stats.csv (the --kernel-trace --stats pass)"""
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            out[row["Name"].split("(")[0]] = float(row["AverageNs"])
    return out


def phase_of(name):
    mo = re.search(r"k_colblock<[^>]*?,\s*(\d),\s*(true|false)>", name) or re.search(r"k_colblock_stream<(\d)>", name)
    if mo:
        return "phase_a" if mo.group(1) == "0" else "pass_2"
    if "k_blockkeys" in name or "k_blocktrie" in name:
        return "phase_a"
    if "#tables" in name or "k_chain_snap" in name:
        return "pass_2"
    if "k_chain" in name or "k_boundary_recent" in name or "k_cm_" in name:
        return "phase_b"
    if "k_columns" in name or "k_reduce_" in name:
        return "phase_c"
    if "k_dp<" in name or "k_spec_" in name:
        return "phase_d"
    if "k_gather" in name or "k_seg_" in name or "k_tb_" in name or "copyBuffer" in name or "fillBuffer" in name:
        return "host"
    return None


def main():
    src, prefix, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    import bench
    stats = find(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(HERE, "%s_kernel_stats_%s.csv" % (prefix, workload)))
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(HERE, "%s_bench_%s.json" % (prefix, workload)))
    counters = {}
    for d in sorted(os.listdir(src)):
        for path in find(os.path.join(src, d, "**", "*counter_collection.csv")):
            for k, cs in per_kernel(path).items():
                for cname, (tot, cnt, dur) in cs.items():
                    counters.setdefault(k, {})[cname] = {"sum": tot, "launches": cnt, "avg_per_launch": tot / cnt, "duration_ns": dur}
    # flat csv of every counter
    with open(os.path.join(HERE, "%s_pmc_counters_%s.csv" % (prefix, workload)), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "launches", "avg_per_launch", "sum"])
        for k in sorted(counters):
            for cname in sorted(counters[k]):
                c = counters[k][cname]
                wr.writerow([k, cname, c["launches"], "%.6g" % c["avg_per_launch"], "%.6g" % c["sum"]])
    kernels, phases = {}, {}
    for k, cs in counters.items():
        if "FETCH_SIZE" not in cs and "WRITE_SIZE" not in cs:
            continue
        f_ = cs.get("FETCH_SIZE", {"sum": 0.0, "launches": 1, "avg_per_launch": 0.0})
        w_ = cs.get("WRITE_SIZE", {"sum": 0.0, "launches": 1, "avg_per_launch": 0.0})
        per_step = (2.0 * f_["sum"] + w_["sum"]) * 1024.0 / nsteps
        kernels[k] = {
            "launches_per_step": max(f_["launches"], w_["launches"]) / nsteps,
            "FETCH_SIZE_KiB_avg_per_launch": f_["avg_per_launch"],
            "WRITE_SIZE_KiB_avg_per_launch": w_["avg_per_launch"],
            "hbm_bytes_per_launch_corrected": (2.0 * f_["avg_per_launch"] + w_["avg_per_launch"]) * 1024.0,
            "hbm_bytes_per_step_corrected": per_step,
            "phase": phase_of(k),
        }
        if phase_of(k):
            phases[phase_of(k)] = phases.get(phase_of(k), 0.0) + per_step
    # issue side, per kernel (round 4): every SQ counter's average per launch, the launch's average duration in the
    # kernel-trace pass, and the EFFECTIVE clock of the profiled launches -- GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    # clock = GRBM_GUI_ACTIVE / 8 / dispatch duration (MI355X_MICROARCH.md, "DVFS give-back"; reads high on dispatches
    # shorter than ~0.3 ms).  bench.py prices the dominant kernel's VALU issue share with it.
    avg_ns = kernel_avg_ns(stats[0]) if stats else {}
    issue = {}
    for k, cs in counters.items():
        d = {c: v["avg_per_launch"] for c, v in cs.items() if c.startswith("SQ_") or c.startswith("GRBM_")}
        if not d:
            continue
        if k in avg_ns:
            d["avg_ns"] = avg_ns[k]
        g = cs.get("GRBM_GUI_ACTIVE")
        if g and g.get("duration_ns"):
            d["clock_ghz_effective"] = g["sum"] / 8.0 / g["duration_ns"]
            d["avg_ns_in_grbm_pass"] = g["duration_ns"] / g["launches"]
        if k in kernels:
            d["hbm_bytes_per_launch_corrected"] = kernels[k]["hbm_bytes_per_launch_corrected"]
        issue[k] = d
    # [r5] per phase and step: every SQ counter summed over the phase's kernels (phase C is several configurations of
    # k_columns_red side by side: bench.py prices the phase, not one of its launches)
    groups = {}
    for k, cs in counters.items():
        ph = phase_of(k)
        if not ph:
            continue
        g = groups.setdefault(ph, {})
        for cname, v in cs.items():
            if cname.startswith("SQ_"):
                g[cname] = g.get(cname, 0.0) + v["sum"] / nsteps
        if k in kernels:
            g["hbm_bytes_corrected"] = g.get("hbm_bytes_corrected", 0.0) + kernels[k]["hbm_bytes_per_step_corrected"]
        gr = cs.get("GRBM_GUI_ACTIVE")
        if gr and gr.get("duration_ns"):
            g["_grbm"] = g.get("_grbm", 0.0) + gr["sum"]
            g["_grbm_ns"] = g.get("_grbm_ns", 0.0) + gr["duration_ns"]
    for g in groups.values():
        if g.get("_grbm_ns"):
            g["clock_ghz_effective"] = g["_grbm"] / 8.0 / g["_grbm_ns"]
        g.pop("_grbm", None); g.pop("_grbm_ns", None)
    out = {
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --workload %s --steps 5 --warmup 2 "
                   "--no-cpu-baseline --no-batched (one pass per counter set, profiles/collect_profiles.sh)" % workload,
        "workload": workload,
        "steps_profiled": nsteps,
        "csrc_sha": bench.csrc_sha(),
        "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (wide coalesced reads are under-counted by half)",
        "phases_hbm_bytes_per_step": phases,
        "kernels": kernels,
        "issue_units": "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles summed over the SIMDs, SQ_INSTS_* in wave-instructions, "
                       "GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs; averages per launch",
        "issue": issue,
        "phase_totals_per_step": groups,
    }
    with open(os.path.join(HERE, "%s_pmc_traffic_%s.json" % (prefix, workload)), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: round(v / 1e6, 2) for k, v in phases.items()}, indent=1))


if __name__ == "__main__":
    main()
