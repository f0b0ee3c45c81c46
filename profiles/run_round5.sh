#!/bin/bash
# Round-5 records (GPU box, repo root): bench lines of every configuration, kernel stats + PMC summaries of the
# march on each, into gpurun_out/ (copied to profiles/r05_* by hand).  bash profiles/run_round5.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
echo "default done"
for c in c2 c3 c4_1gpu c4_maxplanck_1gpu c5_1gpu; do
  python3 bench.py --no-cpu --steps 3 --config $c > gpurun_out/r05_bench_$c.json 2> gpurun_out/r05_bench_$c.err
  python3 profiles/bench_line.py $c < gpurun_out/r05_bench_$c.json
done
for c in c3 c2 c4_1gpu c5_1gpu; do
  bash profiles/run_pmc_march.sh r05_$c $c > gpurun_out/pmc_r05_$c.log 2>&1 || echo "pmc $c failed"
  echo "pmc $c done"
done
