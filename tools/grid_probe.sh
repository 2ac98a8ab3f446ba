# probe (tools/variants/knobs.so = a -DMI_TUNING_KNOBS build): batches in flight against step time
for p in 3 4 5 6 8; do
MI_AVIF_LIB=${1:-cavif_rs_amd/libmi_avif.so} python bench.py --steps 16 --warmup 3 --pipeline $p --no-cpu-baseline --no-pcie-loop --end-to-end 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']; print('pipeline', $p, d['value'], d['ms_per_step'], 'identity', d['output_identity'].get('equal'), '/', d['output_identity'].get('checked'), 'overlapped', {k: round(v,1) for k,v in st.items()})"
done
