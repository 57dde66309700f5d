#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of profiles/collect_profiles.sh into the committed summaries:

    python profiles/summarize_pmc.py gpurun_out/prof_<tag> r01_<tag>

writes profiles/r01_<tag>_kernel_stats.csv, _pmc_fetch_size.csv, _pmc_write_size.csv, _bench.json and
_pmc_traffic.json (average HBM bytes per launch and kernel: FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE counts half of a wide coalesced read stream -- MI355X_MICROARCH.md, HBM section --
so reads are doubled; calibrated on k_synth / k_colblock<MODE_RANK>, whose traffic is known)."""
import csv
import glob
import json
import os
import shutil
import sys


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("no file matches " + pattern)
    return hits[0]


def per_kernel(path):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            a = acc.setdefault(name, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    stats = one(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    fetch = one(os.path.join(src, "fetch", "**", "*counter_collection.csv"))
    write = one(os.path.join(src, "write", "**", "*counter_collection.csv"))
    shutil.copy(stats, os.path.join(here, tag + "_kernel_stats.csv"))
    shutil.copy(fetch, os.path.join(here, tag + "_pmc_fetch_size.csv"))
    shutil.copy(write, os.path.join(here, tag + "_pmc_write_size.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(here, tag + "_bench.json"))
    fk, wk = per_kernel(fetch), per_kernel(write)
    kernels = {}
    for name in fk:
        kernels[name] = {
            "FETCH_SIZE_KiB_avg_per_launch": fk[name],
            "WRITE_SIZE_KiB_avg_per_launch": wk.get(name, 0.0),
            "hbm_bytes_per_launch_corrected": (2.0 * fk[name] + wk.get(name, 0.0)) * 1024.0,
        }
    out = {
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                   "--no-batched (one pass per counter, profiles/collect_profiles.sh); workload C2 m=2500 n=100000",
        "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (wide coalesced reads are under-counted by half)",
        "kernels": kernels,
    }
    with open(os.path.join(here, tag + "_pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 2) for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
