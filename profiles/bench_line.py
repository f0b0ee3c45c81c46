#!/usr/bin/env python3
"""One line per bench run read from stdin (the JSON line of bench.py): the numbers an A/B needs."""
import json
import sys
b = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
c = b.get("culling") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", round(b["ms_per_step"], 2), "value", round(b["value"] / 1e6, 4), "T/s  culled", c.get("culled"),
      "started", c.get("started_fraction"), "prepass ms", round(c.get("prepass_ms_per_frame") or 0, 2), "march ms",
      round(b["roofline"]["avg_launch_ms"], 2), "events/frame", b["config"]["events_executed_per_frame"])
