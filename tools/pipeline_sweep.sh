#!/bin/bash
# steady-state throughput against the number of resident batch slots
for p in "$@"; do
  python bench.py --steps 12 --warmup 4 --pipeline $p --no-cpu-baseline --no-identity-check 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipeline=$p', d['value'], d['ms_per_step'])"
done
