#!/bin/bash
# Short A/B line on the GPU box: per-stage milliseconds of the default bench (pipeline 1), N runs.   tools/gpu_ab.sh [label] [runs]
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-threads-line --pipeline 1 --no-identity-check"
for i in $(seq 1 ${2:-2}); do timeout 200 $B 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); s = d['stage_ms_per_step']; print('${1:-run}', 'ms/step', d['ms_per_step'], {k: round(v, 1) for k, v in s.items()})"; done
