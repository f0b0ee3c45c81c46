#!/bin/bash
# Profile of the shipped march kernel (rounds 2 to 5; round 5: KERN=k_march_cull, the default launch, or
# LF_BENCH_CULL=0 KERN=k_march for the path tree): rocprofv3 kernel stats + PMC counters in SEPARATE
# passes (SQ / FETCH_SIZE / WRITE_SIZE / misc), as MI355X_MICROARCH.md's HBM section prescribes,
# reduced to profiles-ready JSON.  Usage (GPU box, repo root):  bash profiles/run_pmc_march.sh <tag> [config]
# Nothing is built here: the library must exist BEFORE the profiler starts (no exec of hipcc under it).
set -e
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing: run __graft_entry__.build() first" >&2; exit 1; }
TAG=${1:-r05}
CFG=${2:-c3}
KERN=${KERN:-k_march_cull}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu --config $CFG"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu --config $CFG > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/misc -- python3 $ARGS > $OUT/misc.json 2> $OUT/misc.err || true
python3 profiles/summarize_pmc.py $OUT $KERN $CFG > $OUT/summary.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
cat $OUT/summary.json
