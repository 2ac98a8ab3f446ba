B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie-loop --end-to-end 0 --pipeline 1 --no-identity-check"
p() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'K1', d['stage_ms_per_step']['tile_search'], 'K4', d['stage_ms_per_step']['entropy'])"; }
for t in 1 2 3; do MI_AVIF_LIB=$PWD/cavif_rs_amd/libmi_t$t.so timeout 100 $B 2>&1 | tail -1 | p trialdbg$t; done
for d in 0 9 7 5 3; do MI_DEBUG_LEVEL=$d MI_AVIF_LIB=$PWD/cavif_rs_amd/libmi_dbg.so timeout 100 $B 2>&1 | tail -1 | p dbg$d; done
