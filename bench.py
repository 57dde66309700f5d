#!/usr/bin/env python3
"""bench.py -- alignment cells/s through pBWT + segmentation DP on MI355X.

One "step" = one complete pass of the hot path (everything segmentation_lp_context does:
pBWT pass 1 + DP, traceback, segment merge, pass-2 boundary states) over one synthetic
founder-mosaic alignment that is already resident in HBM (column-major, 2 bits per cell for
sigma <= 4, 4 bits for sigma <= 16, else 1 B).
Workload at N=1 [r5]: BASELINE.json configs[3]'s alignment (C4: m=100,000 x n=5,000,000, sigma=4, L=200) -- the
configuration the metric's 1/2/4/8-GPU curve is quoted on; its 125 GB of packed cells fit one MI355X.  The other
single-GPU configurations (C2, C3, C5) are measured in child processes of the same invocation and reported under
config.other_workloads (a failure or a hang there cannot take the headline line with it).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the SAME alignment sharded over the ranks by contiguous column blocks (fseq_set_shard; exchanges through
torch.distributed / RCCL, founder-sequences_amd/dist.py) -> "scaling": "strong"; value = the alignment's cells /
max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

# The HIP runtime maps streams onto 4 hardware queues by default; kernels of different contexts that land on the
# same queue run one after the other.  Only the secondary several-alignments-in-flight figure depends on it
# (measured: 8 in flight 47 -> 94 G cells/s); it must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_CELL = 17             # 1 (symbol) + 4+4 (a_k,d_k read) + 4+4 (a_k+1,d_k+1 written)

WORKLOADS = {
    # name: m, n, L, founders K, recombination block, mutation rate, seed, kind
    "C1": dict(m=8, n=1000, L=10, K=3, B=100, mu=5e-3, seed=0x5EED0001, kind=0),
    "C2": dict(m=2500, n=100000, L=50, K=16, B=2000, mu=1e-4, seed=0x5EED0002, kind=0),
    "C3": dict(m=2504, n=1000000, L=100, K=24, B=5000, mu=1e-4, seed=0x5EED0003, kind=0),
    "C5": dict(m=10000, n=1000000, L=100, K=32, B=5000, mu=1e-4, seed=0x5EED0005, kind=1),
    # BASELINE C4's rows (m = 100,000: HBM-streamed block state) on a column prefix, and C4 itself
    # (5e11 cells = 125 GB at 2 bits per cell: one MI355X holds it; one step takes tens of seconds)
    "C4cols50k": dict(m=100000, n=50000, L=200, K=64, B=10000, mu=5e-5, seed=0x5EED0004, kind=0),
    # [r5] a slice of C4 with the FULL run's geometry (512 blocks of 814 columns: two rounds of the 256 CUs' workgroups of the
    # reduced column kernel, stride states every 32 columns) -- what the PMC passes are taken on (the full C4 under
    # rocprofv3 --pmc is a quarter of an hour per pass); its per-cell figures are carried over to C4
    "C4slice": dict(m=100000, n=416768, L=200, K=64, B=10000, mu=5e-5, seed=0x5EED0004, kind=0, block_len=814),
    "C4": dict(m=100000, n=5000000, L=200, K=64, B=10000, mu=5e-5, seed=0x5EED0004, kind=0),
}


def csrc_sha():
    """Hash of the kernel sources: a PMC summary is only quoted for the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "founder-sequences_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hpp", ".hip")) and "join" not in f:
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc_summary(workload):
    """The newest profiles/r*_pmc_traffic_<workload>.json taken on the current kernel sources (rocprofv3 cannot run
    inside this process), or None: HBM bytes per phase and kernel, and -- since round 4 -- the issue-side counters
    and the effective clock of every kernel (profiles/summarize_pmc.py)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % workload)))
    sha = csrc_sha()
    for path in reversed(hits):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("csrc_sha") != sha:
                continue
            d["_path"] = os.path.relpath(path, ROOT)
            return d
        except Exception:
            continue
    return None


def load_pmc(workload):
    """(HBM bytes per step and phase, source) from load_pmc_summary; (None, reason) without one.  C4: the passes are taken on
    C4slice -- 512 blocks of the full run's geometry -- and its bytes carried over per cell."""
    d = load_pmc_summary(workload)
    scale, note = 1.0, ""
    if d is None and workload == "C4":
        d = load_pmc_summary("C4slice")
        if d is not None:
            scale = (WORKLOADS["C4"]["m"] * WORKLOADS["C4"]["n"]) / float(WORKLOADS["C4slice"]["m"] * WORKLOADS["C4slice"]["n"])
            note = "; taken on C4slice (512 of C4's blocks, the full run's geometry), scaled by cells"
    if d is None:
        return None, "no PMC summary under profiles/ for workload %s on kernel sources %s" % (workload, csrc_sha())
    return ({k: v * scale for k, v in d["phases_hbm_bytes_per_step"].items()},
            d["_path"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 on gfx950%s)" % note)


N_SIMD = 1024                   # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
NOMINAL_CLOCK_GHZ = 2.4


def dominant_kernel_bound(workload, m, n, phase_c_ms, reduced):
    """What bounds phase C (the per-column update + lists: the phase the step spends most of its time in), from the PMC
    passes committed for THIS workload on THESE kernel sources and the phase's time measured live.

    [r5] Phase C runs on the blocks' representative rows (k_columns_red, several configurations side by side), whose order
    lives in LDS: HBM bytes bound nothing, the SIMDs' vector issue does.  frac = SQ_ACTIVE_INST_VALU [quad-cycles, summed over
    the SIMDs and the phase's kernels] x 4 / (1024 SIMDs x clock x phase time): the share of all vector issue slots of the
    phase that issued.  The clock is the effective one of the profiled launches (GRBM_GUI_ACTIVE / 8 XCDs / duration), not an
    assumed 2.4 GHz.  The HBM share -- the bytes the PMC passes counted (FETCH_SIZE x 2 + WRITE_SIZE) over the phase time
    against 8 TB/s -- is put beside it; `bound` names the one nearer its ceiling.  C4's counters are taken on a slice with the
    full run's geometry (C4slice: 512 blocks of 814 columns) and carried over per cell."""
    d = load_pmc_summary(workload)
    prefix = None
    if d is None and workload == "C4":
        d = load_pmc_summary("C4slice")
        prefix = "C4slice"
    if d is None or "phase_totals_per_step" not in d or "phase_c" not in d["phase_totals_per_step"]:
        return {"bound": None, "frac": None, "note": "no PMC phase summary under profiles/ for workload %s on kernel sources %s "
                "(profiles/collect_profiles.sh + summarize_pmc.py write it)" % (workload, csrc_sha())}
    c = d["phase_totals_per_step"]["phase_c"]
    scale = 1.0
    if prefix:
        scale = (m * n) / float(WORKLOADS[prefix]["m"] * WORKLOADS[prefix]["n"])
    clock_ghz = c.get("clock_ghz_effective") or NOMINAL_CLOCK_GHZ
    cycles = clock_ghz * 1e9 * phase_c_ms * 1e-3
    valu_q = c.get("SQ_ACTIVE_INST_VALU", 0.0) * scale
    lds_q = c.get("SQ_ACTIVE_INST_LDS", 0.0) * scale
    conflict = c.get("SQ_LDS_BANK_CONFLICT", 0.0) * scale
    allph = d["phase_totals_per_step"]
    out = {
        "kernel": "k_reduce_prep + k_reduce_msa + k_columns_red<*> (every configuration of the phase)" if reduced else "k_columns / k_columns_stream2",
        "source": d["_path"] + (" (per-cell figures of the %s slice -- the full run's block geometry -- scaled to the full workload)" % prefix if prefix else ""),
        "clock_ghz": round(clock_ghz, 3),
        "clock_source": "GRBM_GUI_ACTIVE / 8 / duration of the profiled launches" if c.get("clock_ghz_effective") else "nominal (no GRBM pass in the summary)",
        "lane_instructions_per_cell": round(c.get("SQ_INSTS_VALU", 0.0) * scale * 64.0 / (m * n), 2),
        "lane_instructions_per_cell_whole_step": round(sum(g.get("SQ_INSTS_VALU", 0.0) for g in allph.values()) * scale * 64.0 / (m * n), 2),
        "valu_issue_frac": round(valu_q * 4.0 / (N_SIMD * cycles), 4) if cycles else None,
        "lds_issue_frac": round(lds_q * 4.0 / (N_SIMD * cycles), 4) if cycles else None,
        "lds_bank_conflict_over_lds_active": round(conflict / lds_q, 3) if lds_q else None,
        "salu_over_valu_insts": round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3) if c.get("SQ_INSTS_VALU") else None,
        "pmc_hbm_bytes_per_step": c.get("hbm_bytes_corrected", 0.0) * scale,
    }
    hbm_frac = out["pmc_hbm_bytes_per_step"] / (phase_c_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if phase_c_ms else None
    out["hbm_frac_pmc_bytes"] = round(hbm_frac, 4) if hbm_frac is not None else None
    out["pmc_bytes_per_cell"] = round(out["pmc_hbm_bytes_per_step"] / (m * n), 3)
    if (out["valu_issue_frac"] or 0.0) >= (out["hbm_frac_pmc_bytes"] or 0.0):
        out["bound"] = "valu_issue"
        out["frac"] = out["valu_issue_frac"]
    else:
        out["bound"] = "hbm"
        out["frac"] = out["hbm_frac_pmc_bytes"]
    return out


def single_gpu_reference(workload):
    """value / ms_per_step of the newest committed single-GPU bench line of `workload` (profiles/rNN_bench_<workload>.json)."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_bench_%s.json" % workload)), reverse=True):
        try:
            with open(path) as f:
                d = json.loads(f.read().strip().splitlines()[-1])
            if d.get("n_gpus") == 1:
                return {"value": d["value"], "ms_per_step": d["ms_per_step"], "source": os.path.relpath(path, here)}
        except Exception:
            continue
    return None


def cpu_baseline(ctx, w, threads):
    """The oracle (CPU restatement, -O2) timed on this host: pass 1 on one core as the reference
    runs it, pass 2 on `threads` threads.  Sample = the whole workload when it is
    small enough, else a column prefix."""
    import numpy as np
    import fso
    m, n, L = w["m"], w["n"], w["L"]
    ncols = n
    budget_cells = 2.6e9                     # ~10-30 s of CPU work (the whole of C3 is 17 s at 0.15 G cells/s)
    if m * n > budget_cells:
        ncols = max(4 * L, int(budget_cells // m))
    msa = np.ascontiguousarray(ctx.get_sequences(0, ncols))        # row-major, as the reference holds it
    t0 = time.perf_counter()
    r = fso.segment_long(msa, L, sample_rate=fso.sample_rate_for(ncols), threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": m * ncols / dt,
        "unit": "cells/s",
        "cores": int(r["pass2_threads"]),       # threads used at the widest point (pass 2); pass 1 + DP are one thread, as in the reference
        "kind": "port",
        "sample": "oracle/fseq_oracle.c on the same synthetic input, first %d of %d columns, all %d rows, row-major; "
                  "pass 1 + DP on 1 thread (%.2f s), pass 2 on %d threads (%.2f s), host has %d cores"
                  % (ncols, n, m, r["t_pass1"], r["pass2_threads"], r["t_pass2"], os.cpu_count() or 0),
        "seconds": dt,
    }


def rowshard_leg(args, w, pkg, fdist, torch, dist, rank, world, local_rank, rehearsal):
    """The north-star partition (positions of the pBWT order sharded over the ranks, two all-reduces per column) on a
    column prefix of the workload; prints one JSON line on rank 0."""
    import threading
    import numpy as np
    m, cols = w["m"], args.rowshard_cols
    sigma, bits = (16, 4) if w["kind"] else (4, 2)
    gen = pkg.SegmentationContext(m, cols, max(1, cols // 4), device=local_rank)
    gen.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
    msa = gen.get_sequences(0, cols)
    gen.close()
    alphabet = np.sort(np.frombuffer(b"ACGTRYSWKMBDHVN-" if w["kind"] else b"ACGT", dtype=np.uint8))
    lut = np.zeros(256, dtype=np.uint8)
    lut[alphabet] = np.arange(len(alphabet), dtype=np.uint8)      # consecutive codes in byte order, as the library assigns them
    packed, ld = pkg.pack_columns(lut[msa], bits)
    dev = torch.device("cuda", local_rank)
    nthreads = args.rowshard_threads if world == 1 else 0
    W = nthreads or world
    words = pkg.rowshard_xbuf_words(m, bits, W)
    per = 8 // bits

    def my_columns(r):
        lo, hi = pkg.rowshard_rows(m, bits, r, W)
        mine = np.zeros_like(packed)
        mine[:, lo // per:(hi + per - 1) // per] = packed[:, lo // per:(hi + per - 1) // per]      # this rank's rows only
        return torch.from_numpy(mine).to(dev)

    results = {}
    if nthreads:
        tw = fdist.ThreadWorld(W)

        def work(r):
            colsd = my_columns(r)
            tr, ar = tw.transport(words, r, dev)
            for _ in range(2):                                     # once to warm up, once timed
                tw.barrier.wait()
                t0 = time.perf_counter()
                out = pkg.rowshard_pbwt(colsd.data_ptr(), ld, m, sigma, bits, cols, r, W, tr.ptr, tr.words, ar if W > 1 else None, device=local_rank)
                tw.barrier.wait()
                results[r] = (time.perf_counter() - t0, out[5])
        ths = [threading.Thread(target=work, args=(r,)) for r in range(W)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = max(v[0] for v in results.values())
        nex = results[0][1]
        transport_name = "ranks as threads of one process on one GPU: barrier + device reduction"
    else:
        colsd = my_columns(rank)
        tr = fdist.ShardTransport(words, dev, dist if world > 1 else None, via_host=rehearsal)
        nex = 0

        def step():
            nonlocal nex
            nex = pkg.rowshard_pbwt(colsd.data_ptr(), ld, m, sigma, bits, cols, rank, world, tr.ptr, tr.words,
                                    tr.allreduce if world > 1 else None, device=local_rank)[5]
        dt = fdist.timed_steps(step, 1, 1, dist=dist if world > 1 else None, device_sync=torch.cuda.synchronize,
                               tensor_factory=lambda v: torch.tensor(v, dtype=torch.float64, device="cpu" if rehearsal else "cuda"))
        transport_name = "1 GPU, no exchange" if world == 1 else ("gloo via host (rehearsal: ranks share a GPU)" if rehearsal else "RCCL")
    # devices the ranks really ran on (ranks as threads, or a rehearsal with ranks sharing the cards: fewer than ranks)
    n_devices = 1 if (nthreads or world == 1) else (min(world, max(1, torch.cuda.device_count())) if rehearsal else world)
    if rank == 0:
        print(json.dumps({
            "metric": "row-sharded pBWT sweep, columns/s (north-star partition; conformance path)",
            "value": cols / dt, "unit": "columns/s", "n_gpus": n_devices, "ranks": W, "higher_is_better": True,
            "us_per_column": dt / cols * 1e6, "cells_per_s": m * cols / dt, "exchanges": int(nex),
            "exchanges_per_column": nex / cols, "scaling": "strong", "data": "synthetic", "dtype": "u32",
            "config": {"workload": "%s prefix: m=%d x %d columns, sigma=%d; positions of the order and rows of the alignment sharded over %d ranks"
                                   % (args.workload, m, cols, sigma, W),
                       "transport": transport_name}}), flush=True)


# BASELINE.json's other single-GPU configurations measured beside the headline: (steps, warm-up)
OTHER_WORKLOADS = {"C2": (20, 3), "C3": (20, 5), "C5": (10, 3)}


def run_workload(pkg, fdist, torch, dist, name, steps, warmup, args, world, rank, local_rank, rehearsal):
    """`warmup` untimed and exactly `steps` timed steps of workload `name` (barrier + torch.cuda.synchronize() on both
    sides, max over ranks); returns {"line": the JSON line without cpu_baseline, "ctx": the context, "extra": []}."""
    w = dict(WORKLOADS[name])
    m, n, L = w["m"], w["n"], w["L"]
    ctx = pkg.SegmentationContext(m, n, L, block_len=args.block_len or w.get("block_len", 0), list_cap=args.list_cap, device=local_rank)
    transport = None
    if world > 1:
        # ONE alignment over the ranks: this rank generates and keeps its own column blocks only
        transport = fdist.shard_context(ctx, rank, world, dist, torch.device("cuda", local_rank), via_host=rehearsal)
        if rehearsal:
            # ranks that share a card must not each plan with the whole of it (the pass-2 stride states take what is free)
            sharing = (world + max(1, torch.cuda.device_count()) - 1) // max(1, torch.cuda.device_count())
            ctx.set_memory_budget(int(torch.cuda.get_device_properties(local_rank).total_memory * 0.9 / sharing))
    ctx.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
    # optional: more alignments in flight on the same GPU (the DP of one alignment occupies one CU)
    extra = []
    for j in range(1, max(1, args.concurrent)):
        e = pkg.SegmentationContext(m, n, L, block_len=args.block_len, list_cap=args.list_cap, device=local_rank)
        e.generate_synthetic(fdist.seed_for_alignment(w["seed"], j), w["K"], w["B"], w["mu"], w["kind"])
        extra.append(e)

    phase = {}
    counted = {"on": False}

    def step():
        if extra:
            import threading
            ths = [threading.Thread(target=e.run) for e in extra]
            for th in ths:
                th.start()
            ctx.run()
            for th in ths:
                th.join()
        else:
            ctx.run()
        if counted["on"]:
            t = ctx.timings()
            for k in ("ms_phase_a", "ms_phase_b", "ms_phase_c", "ms_dp", "ms_pass2", "ms_host", "ms_total"):
                phase[k] = phase.get(k, 0.0) + t[k]

    # the first run on a fresh context, timed by itself: allocation of the work buffers, the list-capacity estimate, the plan
    # of the reduced phase C -- what a one-shot caller (the CLI) pays.  It counts as the first warm-up step.
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t_cold) * 1e3
    for _ in range(max(0, warmup - 1)):
        step()
    counted["on"] = True
    # barrier + torch.cuda.synchronize() on both sides of exactly `steps` steps, max over ranks
    dt = fdist.timed_steps(step, steps, 0, dist=dist if world > 1 else None, device_sync=torch.cuda.synchronize,
                           tensor_factory=lambda v: torch.tensor(v, dtype=torch.float64, device="cpu" if rehearsal else "cuda"))

    t = ctx.timings()
    res = ctx.result
    nsteps = max(1, steps)
    ph = {k: v / nsteps for k, v in phase.items()}
    step_ms = dt / nsteps * 1e3
    R = int(t["pass2_cells"])
    # ONE figure for the path -- 17 B per cell per pass-1 column update, pass 2's R cells at the
    # same price, over the whole step (pass 1 + DP + traceback + merge + pass 2)
    path_bytes = BYTES_PER_CELL * (m * n + R)
    path_gbps = path_bytes / (step_ms * 1e-3) / 1e9
    # per kernel: HIP-event time on the library's own stream (fseq_timings), algorithmic bytes of what the launch
    # processes, and -- when profiles/ holds PMC passes of THIS workload taken on THESE kernel sources -- the HBM
    # bytes rocprofv3 counted (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section)
    pmc, pmc_src = load_pmc(name)
    launches_c = 1 + t["retries"]
    reduced = t["reduced_blocks"] > 0
    kernels = {
        "phase_a k_blockkeys / k_blocktrie (block keys ranked in key space)": {"ms": ph["ms_phase_a"], "algorithmic_bytes": BYTES_PER_CELL * m * n},
        "phase_b k_chain / k_cm_* (boundary states)": {"ms": ph["ms_phase_b"], "algorithmic_bytes": 0},
        ("phase_c k_reduce_prep + k_columns_red (per-column update + lists on the blocks' representative rows)" if reduced else
         "phase_c k_columns (per-column update + lists)"): {"ms": ph["ms_phase_c"], "algorithmic_bytes": BYTES_PER_CELL * m * n * launches_c},
        "phase_d k_dp<SPEC> sweeps + rebuild kernels": {"ms": ph["ms_dp"], "algorithmic_bytes": 0},
        ("pass_2 k_columns_red (class tables) + k_chain_snap (one chain step per boundary)" if reduced else "pass_2 k_colblock<SNAP>"): {"ms": ph["ms_pass2"], "algorithmic_bytes": BYTES_PER_CELL * R},
        "host (traceback walk, merge, copies)": {"ms": ph["ms_host"], "algorithmic_bytes": 0},
    }
    traffic = None
    if pmc:
        traffic = 0.0
        for label, d in kernels.items():
            if label.split()[0] in pmc:
                d["pmc_hbm_bytes_per_step"] = pmc[label.split()[0]]
                traffic += pmc[label.split()[0]]
    for d in kernels.values():
        d["ms"] = round(d["ms"], 4)
        d["algorithmic_GBps"] = round(d["algorithmic_bytes"] / (d["ms"] * 1e-3) / 1e9, 1) if d["ms"] > 0 and d["algorithmic_bytes"] else None
    ms_c = ph["ms_phase_c"]
    # the dominant kernel: what really bounds it (VALU issue for the LDS-resident kernels, PMC-counted HBM bytes for the
    # streamed one); the 17 B/cell figure stays as `algorithmic_frac` -- an accounting device that can exceed 1 because
    # the order of a block never leaves LDS
    dom = dominant_kernel_bound(name, m, n, ms_c / launches_c, reduced)
    dom_out = {
        "name": ("phase C on the blocks' representative rows: k_reduce_prep, k_reduce_msa (streamed rows), k_columns_red in the configurations that hold "
                 "the blocks, side by side -- the per-column pBWT update + divergence-histogram top list" if reduced else
                 "k_columns (phase C: per-column pBWT update + divergence-histogram top list, one launch over all column blocks)"),
        "avg_launch_ms": ms_c / launches_c,
        "bound": dom.get("bound"),
        "frac": dom.get("frac"),
        "algorithmic_bytes_per_launch": BYTES_PER_CELL * m * n,
        "algorithmic_achieved_GBps": BYTES_PER_CELL * m * n / (ms_c / launches_c * 1e-3) / 1e9,
        "algorithmic_frac": BYTES_PER_CELL * m * n / (ms_c / launches_c * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "algorithmic_frac_note": "17 B/cell over the phase time against 8 TB/s: NOT a fraction of anything (it exceeds 1): the order (a, d) of a block "
                                 "never leaves LDS, and since round 5 a column updates the block's representative rows only -- the rows that "
                                 "can differ in what the DP reads --, not all m; `frac` is the bound that can be approached",
        "rows_updated_per_column_mean": t["reduced_rows_mean"] if reduced else m,
    }
    dom_out.update({k: v for k, v in dom.items() if k not in ("bound", "frac")})
    out = {
        "metric": "alignment cells/s (m*n/T) through pBWT+DP",
        "value": max(1, args.concurrent) * m * n * nsteps / dt,
        "unit": "cells/s",
        "n_gpus": (min(world, max(1, torch.cuda.device_count())) if rehearsal else world),      # devices, not ranks: a rehearsal shares cards
        "ranks": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "%s: m=%d x n=%d synthetic founder-mosaic DNA (sigma=%d), segment-length-bound L=%d, "
                        "input resident in HBM column-major, %d bits per cell; %s"
                        % (name, m, n, 16 if w["kind"] else 4, L, 4 if w["kind"] else 2,
                           "one alignment on one GPU" if world == 1 else
                           "ONE alignment sharded over %d ranks by contiguous column blocks (each rank holds its own columns)" % world),
            "parallelism": "1 GPU" if world == 1 else "column-block shards x%d: phase A/C/pass 2 and the DP chunks local, %d all-reduces (%.1f MB) per step over %s"
                           % (world, transport.calls // max(1, steps + warmup), transport.words_moved * 4 / max(1, steps + warmup) / 1e6,
                              "gloo via host (rehearsal: ranks share a GPU)" if rehearsal else "RCCL"),
            # strong scaling is judged against ONE GPU on the SAME workload: the N = 1 default of this script is C3, so the
            # committed single-GPU line of this workload is quoted here (python bench.py --workload C4 reproduces it; since
            # round 4 the N = 1 line also carries it under config.other_workloads)
            "single_gpu_same_workload": single_gpu_reference(name) if world > 1 else None,
            "alignments_in_flight_per_gpu": max(1, args.concurrent),
            "cold_first_run_ms": round(cold_ms, 3),
            "reduced_blocks": t["reduced_blocks"], "reduced_rows_mean": t["reduced_rows_mean"], "reduced_redone": t["reduced_redone"],
            "block_len": t["block_len"], "n_blocks": t["n_blocks"], "list_cap": t["list_cap_used"],
            "dp_chunks": t["dp_chunks"], "dp_sweeps": t["dp_sweeps"],
            "segments": int(res.segment_count), "dp_segments": int(res.dp_segment_count),
            "max_segment_size": int(res.max_segment_size),
            "phases_ms": {k: round(v, 4) for k, v in ph.items()},
            "pass2_cells": R,
            # pass 1 alone (pBWT phases A-C + DP)
            "pass1_only_cells_per_s": m * n / (max(1e-9, ph["ms_phase_a"] + ph["ms_phase_b"] + ph["ms_phase_c"] + ph["ms_dp"]) * 1e-3),
        },
        "roofline": {
            "bound": "hbm",
            "scope": "whole path, SURVEY.md 8(d)'s ACCOUNTING figure: 17 B x (m*n + R) / step time -- what a column-at-a-time implementation that "
                     "streams (a, d) of all m rows through HBM would move.  This implementation keeps a block's order in LDS and updates its "
                     "representative rows only, so frac exceeds 1 and is not a fraction of a physical limit; the bytes HBM really moves are under "
                     "`traffic` (hbm_frac_measured = traffic / step time / peak), the bound that can be approached under dominant_kernel",
            "achieved": path_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": path_gbps / HBM_PEAK_GBPS,
            "frac_vs_measured_copy_6290": path_gbps / 6290.0,
            "traffic": traffic,
            "traffic_source": pmc_src,
            "hbm_frac_measured": (traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            "algorithmic_bytes_per_step": path_bytes,
            "dominant_kernel": dom_out,
            "kernels": kernels,
        },
    }
    return {"line": out, "ctx": ctx, "extra": extra}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--block-len", type=int, default=0)
    ap.add_argument("--list-cap", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-batched", action="store_true", help="(default) skip the secondary 8-alignments-in-flight measurement")
    ap.add_argument("--batched", action="store_true", help="also measure 8 alignments in flight on the one GPU (secondary figure)")
    ap.add_argument("--concurrent", type=int, default=1,
                    help="alignments in flight per GPU (one context + host thread each); 1 = the headline single-alignment workload")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="N = 1 without --workload also measures C2, C3 and C5 in child processes (config.other_workloads); this skips them")
    ap.add_argument("--rowshard-cols", type=int, default=0,
                    help="instead of the headline step: the north-star row split (fseq_rowshard_pbwt) over this many columns of the "
                         "workload, one JSON line of its own (columns/s; latency-bound by design, DESIGN.md section 6)")
    ap.add_argument("--rowshard-threads", type=int, default=0,
                    help="with --rowshard-cols on ONE GPU: this many ranks as threads of the process (all-reduce = barrier + "
                         "device reduction) instead of torch.distributed ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the segmentation path has no CPU fallback")
    # rehearsal on a box with fewer GPUs than ranks (FSEQ_BENCH_REHEARSAL=1): ranks share the cards and the
    # timing collectives run over gloo on CPU tensors; everything else is the path the driver launches
    rehearsal = bool(os.environ.get("FSEQ_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("founder-sequences_amd")
    fdist = importlib.import_module("founder-sequences_amd.dist")
    headline_default = args.workload is None
    if args.workload is None:
        args.workload = "C4"
    w = dict(WORKLOADS[args.workload])
    m, n, L = w["m"], w["n"], w["L"]
    if args.rowshard_cols:
        rowshard_leg(args, w, pkg, fdist, torch, dist, rank, world, local_rank, rehearsal)
        if world > 1:
            dist.destroy_process_group()
        return
    run = run_workload(pkg, fdist, torch, dist, args.workload, args.steps, args.warmup, args, world, rank, local_rank, rehearsal)
    ctx, out = run["ctx"], run["line"]
    # secondary figure (not `value`): the DP of one alignment occupies one CU for ~2/3 of a step, so
    # several alignments (chromosomes) in flight share the chip; measured with 8 contexts / host threads
    if world == 1 and args.concurrent == 1 and args.batched:
        others = []
        for j in range(1, 8):
            e = pkg.SegmentationContext(m, n, L, block_len=args.block_len, list_cap=args.list_cap, device=local_rank)
            e.generate_synthetic(fdist.seed_for_alignment(w["seed"], j), w["K"], w["B"], w["mu"], w["kind"])
            others.append(e)

        def bstep():
            rcs = pkg.run_batch([ctx] + others)          # fseq_run_segmentation_batch: one native thread and stream per context
            assert all(rc == pkg.FSEQ_OK for rc in rcs), rcs

        bdt = fdist.timed_steps(bstep, 3, 1, dist=None, device_sync=torch.cuda.synchronize, tensor_factory=None)
        out["batched_throughput"] = {"alignments_in_flight": 8, "value": 8 * m * n * 3 / bdt, "unit": "cells/s",
                                     "ms_per_step": bdt / 3 * 1e3,
                                     "note": "8 alignments of the same shape in flight on the one GPU through fseq_run_segmentation_batch (one context, two streams and one host thread each; GPU_MAX_HW_QUEUES=%s); not the headline metric" % os.environ.get("GPU_MAX_HW_QUEUES")}
        for e in others:
            e.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctx, w, args.cpu_threads or min(os.cpu_count() or 1, 16))
    else:
        out["cpu_baseline"] = None
    ctx.close()
    for e in run["extra"]:
        e.close()
    # The other single-GPU configurations of BASELINE.json, each with its own step count, in the same invocation (only when
    # the driver's plain command is run: no --workload given): C2 (configs[1]), C3 (configs[2]) and C5 (configs[4]'s
    # alignment on one GPU).  Each in a CHILD PROCESS of its own with a time limit: a failure, a hang or a GPU fault there
    # cannot take the headline line with it.  Not part of `value`.
    if world == 1 and headline_default and not args.no_other_workloads and args.concurrent == 1:
        import subprocess
        others_out = {}
        for name, (k_steps, k_warm) in OTHER_WORKLOADS.items():
            t0 = time.perf_counter()
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(k_steps), "--warmup", str(k_warm),
                                     "--no-cpu-baseline", "--no-other-workloads"], capture_output=True, text=True, timeout=300)
                ln = json.loads(cp.stdout.strip().splitlines()[-1])
                others_out[name] = {
                    "value": ln["value"], "unit": "cells/s", "ms_per_step": ln["ms_per_step"], "steps": k_steps, "warmup": k_warm,
                    "workload": ln["config"]["workload"], "phases_ms": ln["config"]["phases_ms"], "pass2_cells": ln["config"]["pass2_cells"],
                    "cold_first_run_ms": ln["config"]["cold_first_run_ms"],
                    "reduced_blocks": ln["config"]["reduced_blocks"], "reduced_rows_mean": ln["config"]["reduced_rows_mean"],
                    "block_len": ln["config"]["block_len"], "n_blocks": ln["config"]["n_blocks"], "list_cap": ln["config"]["list_cap"],
                    "dp_sweeps": ln["config"]["dp_sweeps"], "segments": ln["config"]["segments"], "max_segment_size": ln["config"]["max_segment_size"],
                    "path_frac": ln["roofline"]["frac"], "path_achieved_GBps": ln["roofline"]["achieved"],
                    "traffic": ln["roofline"]["traffic"], "hbm_frac_measured": ln["roofline"]["hbm_frac_measured"],
                    "dominant_kernel": ln["roofline"]["dominant_kernel"],
                    "wall_s_incl_setup": round(time.perf_counter() - t0, 2),
                }
            except Exception as ex:                       # a failure here must not take the headline line with it
                others_out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        out["config"]["other_workloads"] = others_out
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
