#!/bin/bash
# A/B timing of library variants inside one gpurun call (boxes differ by several percent): tools/ab.sh lib1.so lib2.so ...
# prints: lib, MPix/s, tile search ms, entropy ms, cdef ms, output identity
for rep in 1 2; do for lib in "$@"; do
  MI_AVIF_LIB=$lib python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('$lib', d['value'], st['tile_search'], st['entropy'], st['cdef'], d['output_identity'])"
done; done
