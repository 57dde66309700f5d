#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash tools/collect_all.sh <tag>
# The round's measurements in one go: rocprofv3 kernel stats + PMC passes for the workloads bench.py quotes them for
# (profiles/collect_profiles.sh), the driver's own bench command, and the single-workload bench lines.  Outputs under
# gpurun_out/<tag>/ and gpurun_out/prof_<tag>_<workload>/; profiles/summarize_pmc.py then makes the committed summaries.
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for wl in C4slice C3 C5 C2; do
	bash profiles/collect_profiles.sh ${tag}_$wl $wl > gpurun_out/$tag/collect_$wl.log 2>&1 || echo "collect_profiles failed for $wl"
done
python3 bench.py --steps 5 --warmup 2 > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
for wl in C2 C3 C5; do
	python3 bench.py --workload $wl --steps 5 --warmup 2 > gpurun_out/$tag/bench_$wl.json 2> gpurun_out/$tag/bench_$wl.err
done
echo "collect_all done"
