#!/bin/bash
# occupancy experiment: K1 variants (waves/SIMD cap) x batch size
for w in 2 3 4; do for b in 32 64 128; do
  MI_AVIF_LIB=$PWD/cavif_rs_amd/libmi_avif_w$w.so timeout 300 python bench.py --steps 1 --warmup 1 --batch $b --no-cpu-baseline --no-identity-check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w=$w batch=$b', d['value'], 'MPix/s', d['stage_ms_per_step'])"
done; done
