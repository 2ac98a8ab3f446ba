# probe (a -DMI_TUNING_KNOBS build in cavif_rs_amd/libmi_v_knobs.so): K1's synchronisation grain (MI_K1_FINE), list order key (MI_K1_KEY=a,b) and direction
# (MI_K1_ORDER=1: longest remaining chain first) for 32 and 16 tiles per image.   usage: tools/fine_probe.sh "FINE KEY ORDER" ...
for T in 0 16; do for V in "$@"; do A=($V)
MI_K1_FINE=${A[0]} MI_K1_KEY=${A[1]} MI_K1_ORDER=${A[2]} MI_AVIF_LIB=cavif_rs_amd/libmi_v_knobs.so python bench.py --steps 6 --warmup 1 --pipeline 1 --threads $T --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-threads-line 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('threads $T fine ${A[0]} key ${A[1]} order ${A[2]}  %7.2f MPix/s  K1 %7.3f  id %s/%s' % (d['value'], st['tile_search'], d['output_identity'].get('equal'), d['output_identity'].get('checked')))"
done; done
