#!/bin/bash
# steady-state throughput against the number of resident batch slots, the driver's form of the run (20 steps, 5 warm-up): tools/pipeline_sweep.sh 1 2 3 4
for p in "$@"; do
  python bench.py --steps 20 --warmup 5 --pipeline $p --no-cpu-baseline --no-identity-check --no-pcie-loop --end-to-end 0 --no-threads-line 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipeline=$p', d['value'], d['ms_per_step'])"
done
