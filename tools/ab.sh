#!/bin/bash
# A/B timing of library variants inside one gpurun call (boxes differ by several percent): tools/ab.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do
  MI_AVIF_LIB=$lib python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['stage_ms_per_step']['tile_search'], d['stage_ms_per_step']['entropy'], d['output_identity'])"
done; done
