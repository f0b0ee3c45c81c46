#!/bin/bash
# A/B timing of alternative builds / settings of k_march on the GPU box (experiments):
#   bash profiles/ab_march.sh <tag> "<name>=<env assignments>" ...
# each variant runs bench.py --no-cpu --steps 5 and its line goes to gpurun_out/ab_<tag>_<name>.json
TAG=$1; shift
for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  env $envs python3 bench.py --no-cpu --steps 5 --warmup 1 > gpurun_out/ab_${TAG}_${name}.json 2> gpurun_out/ab_${TAG}_${name}.err || echo "variant $name failed"
  python3 - "$name" gpurun_out/ab_${TAG}_${name}.json <<'P'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(b["ms_per_step"],2), "march_ms", round(b["roofline"]["avg_launch_ms"],2), "events/frame", b["config"].get("events_per_frame"))
except Exception as e:
    print(sys.argv[1], "no result", e)
P
done
