#!/bin/bash
# diagnostic: time the rule-update stage with parts of the update kernel switched off
for d in 0 1 2 4 8 15; do
  SSMAP_DEBUG=$d timeout -k 10 120 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$d', d['stage_ms_per_step'])" || exit 1
done
