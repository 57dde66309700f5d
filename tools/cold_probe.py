"""Where a first run on a fresh context spends its wall time: FSEQ_DEBUG=1 python tools/cold_probe.py <workload>"""
import importlib, sys, time, os
import torch
torch.cuda.init()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import bench
pkg = importlib.import_module("founder-sequences_amd")
name = sys.argv[1]
w = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2:
    w["K"] = int(sys.argv[2])
if len(sys.argv) > 3:
    w["mu"] = float(sys.argv[3])
ctx = pkg.SegmentationContext(w["m"], w["n"], w["L"])
ctx.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
torch.cuda.synchronize()
for i in range(2):
    t0 = time.perf_counter()
    try:
        ctx.run()
    except pkg.NoReduction:
        pass
    torch.cuda.synchronize(); print("run", i, (time.perf_counter() - t0) * 1e3, "ms", file=sys.stderr)
