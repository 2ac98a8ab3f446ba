#!/bin/bash
# quick parity (8 cases vs the oracle) + single-slot bench line (value, stage ms, identity)
python tools/gpu_quickcheck.py 2>&1 | tail -8 | cut -c1-110
python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['output_identity'])"
