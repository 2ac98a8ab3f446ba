#!/bin/bash
# A/B timing of library variants inside one gpurun call (boxes differ by several percent): tools/ab.sh lib1.so lib2.so ...
# prints: lib, MPix/s (one batch slot), tile search ms (mean over the steps; one slot = nothing overlaps), isolated launch, entropy ms, cdef ms, output identity
STEPS=${AB_STEPS:-6}
for rep in 1 2 ${AB_REPS}; do for lib in "$@"; do
  MI_AVIF_LIB=$lib python bench.py --steps $STEPS --warmup 1 --pipeline ${AB_PIPELINE:-1} --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-threads-line 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('%-36s %7.2f MPix/s  K1 %7.3f (iso %7.3f)  K4 %5.2f  cdef+lr %5.2f  dbk %5.2f  id %s/%s' % ('$lib'.split('/')[-1], d['value'], st['tile_search'], d['roofline']['avg_launch_ms'], st['entropy'], st['cdef'], st['deblock'], d['output_identity'].get('equal'), d['output_identity'].get('checked')))"
done; done
