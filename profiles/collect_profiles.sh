#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash profiles/collect_profiles.sh <tag>
# Three separate rocprofv3 runs of the same bench command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE --
# counters never combined with a trace domain), outputs under gpurun_out/prof_<tag>/.  The summaries are
# then copied into profiles/ by profiles/summarize_pmc.py (run in the development container).
set -e
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batched"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $BENCH > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $BENCH > $out/bench_write.json 2> $out/write.err
python3 bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err
find $out -name '*.csv' | sort
