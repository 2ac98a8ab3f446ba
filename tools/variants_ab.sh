#!/bin/bash
# One GPU call: parity quick check and single-slot / three-slot bench of the product library and the prepared variants (tools/build_variants.sh first).
# Every run sits under its own `timeout` (a variant whose waves wait for each other must not hang the box).
# Prints per variant: MPix/s, tile search ms, entropy ms, output identity.
L=$PWD/cavif_rs_amd
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('$1', d['value'], 'K1', st['tile_search'], 'K4', st['entropy'], 'identity', d.get('output_identity'))"; }
for P in 1 3; do
  ID=""; [ $P = 3 ] && ID="--no-identity-check"      # the oracle encodes one 1080p image per identity check (10 s of CPU): once per variant is enough
  timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "default/slots$P"
  MI_K1_QUEUE=1 MI_AVIF_LIB=$L/libmi_queue.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "queue/slots$P"
  MI_AVIF_LIB=$L/libmi_pipe.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "pipe/slots$P"
  MI_AVIF_LIB=$L/libmi_pipe3.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "pipe3/slots$P"
  MI_AVIF_LIB=$L/libmi_pipek.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "pipek (three kernels)/slots$P"
  MI_AVIF_LIB=$L/libmi_diet.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "diet(5 wg/cu, 96 vgpr)/slots$P"
  MI_K1_LDS_PAD=544 MI_AVIF_LIB=$L/libmi_diet.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "diet(96 vgpr, LDS padded to 33 024 B: 4 searches + 2 entropy coders per CU)/slots$P"
  MI_AVIF_LIB=$L/libmi_diet4.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "diet4(4 wg/cu, 128 vgpr)/slots$P"
  MI_K1_QUEUE=1 MI_AVIF_LIB=$L/libmi_combo.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "combo (queue + diet5 + pipek)/slots$P"
  MI_K1_QUEUE=1 MI_K1_QUEUE_WG_PER_CU=4 MI_K1_LDS_PAD=544 MI_AVIF_LIB=$L/libmi_combo.so timeout 180 python bench.py --steps 4 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop $ID 2>&1 | tail -1 | line "combo, 4 searches per CU + room for the entropy kernels/slots$P"
done
for V in queue pipe pipe3 pipek diet diet4 combo; do MI_K1_QUEUE=1 MI_AVIF_LIB=$L/libmi_$V.so timeout 180 python tools/gpu_quickcheck.py 2>&1 | tail -8 | grep -c "bytes_equal=True recon_equal=True" | sed "s/^/$V quickcheck ok cases: /"; done
MI_ORACLE_LIB=$PWD/oracle/_build/liboracle_rect.so MI_AVIF_LIB=$L/libmi_rect.so timeout 180 python tools/gpu_quickcheck.py 2>&1 | tail -8 | grep -c "bytes_equal=True recon_equal=True" | sed "s/^/rect quickcheck ok cases: /"
MI_AVIF_LIB=$L/libmi_rect.so timeout 180 python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-pcie-loop --no-identity-check 2>&1 | tail -1 | line "rect/slots1 (identity not checked: the bench compares with the default oracle)"
