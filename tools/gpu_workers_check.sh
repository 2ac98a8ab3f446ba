#!/bin/bash
# one GPU call: parity quick check (small images run with 2+ row workers per tile), single-image latency with and without row workers,
# and the 1024-tile batch with 1 and 2 workers per tile
python tools/gpu_quickcheck.py 2>&1 | tail -8 | cut -c1-120
python tools/secondary_latency.py 2 3 5 2>&1 | tail -3
MI_K1_WORKERS=1 python tools/secondary_latency.py 2 3 2>&1 | tail -2
for W in 1 2; do MI_K1_WORKERS=$W python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-pcie-loop 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch workers=$W', d['value'], d['stage_ms_per_step']['tile_search'], d['output_identity'])"; done
