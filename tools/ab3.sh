for rep in 1 2 3; do for lib in tools/variants/o3_cur.so tools/variants/o2_cur.so; do
  MI_AVIF_LIB=$lib python bench.py --steps 4 --warmup 1 --pipeline 1 --no-cpu-baseline --no-pcie-loop --end-to-end 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('$lib', d['value'], 'K1', d['roofline']['avg_launch_ms'], 'K4', round(st['entropy'],2), 'cdef+lr', round(st['cdef'],2), 'dbk', round(st['deblock'],2))"
done; done
