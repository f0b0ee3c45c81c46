#!/bin/bash
# Collect PMC counters for the march kernel: separate rocprofv3 passes (SQ / FETCH_SIZE / WRITE_SIZE),
# as MI355X_MICROARCH.md's HBM section prescribes.  Usage (on the GPU box, from the repo root):
#   bash profiles/run_pmc.sh <tag>
set -e
# the library must exist BEFORE the profiler starts: nothing may build (exec hipcc) under rocprofv3
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing: run __graft_entry__.build() first" >&2; exit 1; }
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/misc -- python3 $ARGS > $OUT/misc.json 2> $OUT/misc.err || true
python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json
cat $OUT/summary.json
