#!/bin/bash
# the stages after the tile search on a high-priority stream, with persistent / leaving tile-search workgroups: tools/prio_probe.sh LIB
LIB=$1
for P in 4; do for PR in 0 1; do for N in 0 1; do
  MI_AVIF_LIB=$LIB MI_POSTK1_PRIORITY=$PR MI_K1_ITEMS_PER_WG=$N python bench.py --steps 12 --warmup 3 --pipeline $P --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-threads-line 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('slots $P priority $PR items_per_wg $N:', d['value'], 'MPix/s', d['ms_per_step'], 'ms/step', {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, d['output_identity'].get('equal'), '/', d['output_identity'].get('checked'))"
done; done; done
