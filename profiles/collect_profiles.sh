#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash profiles/collect_profiles.sh <tag> <workload>
# Separate rocprofv3 runs of the same bench command: kernel trace + stats; then one run per counter set
# (counters never combined with a trace domain).  Outputs under gpurun_out/prof_<tag>/; the summaries are then
# copied into profiles/ by profiles/summarize_pmc.py (run in the development container).
set -e
tag=${1:-x}
wl=${2:-C3}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $wl --steps ${FSEQ_PROFILE_STEPS:-5} --warmup 2 --no-cpu-baseline --no-batched --no-other-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $BENCH > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $BENCH > $out/bench_write.json 2> $out/write.err
# issue side: instructions by class, LDS conflicts, wave and busy cycles
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/insts -- $BENCH > $out/bench_insts.json 2> $out/insts.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/active -- $BENCH > $out/bench_active.json 2> $out/active.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/waits -- $BENCH > $out/bench_waits.json 2> $out/waits.err
# the effective clock of every launch (round 4): GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (GRBM slots are independent of SQ / TCC)
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- $BENCH > $out/bench_grbm.json 2> $out/grbm.err
python3 bench.py --workload $wl --steps ${FSEQ_PROFILE_STEPS:-5} --warmup 2 --no-other-workloads > $out/bench.json 2> $out/bench.err
find $out -name '*.csv' | sort
