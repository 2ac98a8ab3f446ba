#!/bin/bash
# A/B timing of library variants inside one gpurun call (boxes differ by several percent): tools/ab.sh lib1.so lib2.so ...
# prints: lib, MPix/s (one batch slot), isolated tile search ms, entropy ms, cdef ms, output identity
for rep in 1 2; do for lib in "$@"; do
  MI_AVIF_LIB=$lib python bench.py --steps 4 --warmup 1 --pipeline 1 --no-cpu-baseline --no-pcie-loop --end-to-end 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('$lib', d['value'], 'K1', d['roofline']['avg_launch_ms'], 'K4', round(st['entropy'],2), 'cdef+lr', round(st['cdef'],2), 'dbk', round(st['deblock'],2), d['output_identity'].get('equal'), '/', d['output_identity'].get('checked'))"
done; done
