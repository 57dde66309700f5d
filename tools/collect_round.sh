#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash tools/collect_round.sh <tag>
# The round's measurements beside tools/collect_all.sh: the input's diversity, end to end (upload, segmentation, join, founders
# file), first runs on fresh contexts, and the driver's 2-rank launch shape with both ranks on the one card (a rehearsal: the
# pool has one GPU per box).  Outputs under gpurun_out/<tag>/; the ones to be judged are copied into profiles/ afterwards.
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python3 tools/diversity_sweep.py > gpurun_out/$tag/diversity_sweep.txt 2> gpurun_out/$tag/diversity_sweep.err
for w in "C2 greedy" "C3 greedy" "C3 bipartite"; do
	set -- $w
	python3 tools/e2e_profile.py $1 $2 > gpurun_out/$tag/e2e_$1_$2.json 2> gpurun_out/$tag/e2e_$1_$2.err
done
for wl in C3 C4; do
	echo "== $wl" >> gpurun_out/$tag/cold_probe.txt
	FSEQ_DEBUG=1 python3 tools/cold_probe.py $wl >> gpurun_out/$tag/cold_probe.txt 2>&1
done
FSEQ_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
	bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/$tag/rehearsal_C4_2ranks_one_gpu.json 2> gpurun_out/$tag/rehearsal.err
echo "collect_round done"
