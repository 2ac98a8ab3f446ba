#!/bin/bash
# K1 time with parts of the partition search disabled (MI_DEBUG_LEVEL: 9 = no 4x4 at all, 10 = 4x4 trials but never chosen, 7 = no split trials)
for d in "$@"; do
  MI_DEBUG_LEVEL=$d python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-identity-check 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d', d['value'], d['stage_ms_per_step']['tile_search'], d['stage_ms_per_step']['entropy'])"
done
