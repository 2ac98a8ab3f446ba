#!/bin/bash
# rocprofv3 PC sampling (stochastic, hardware) of one single-slot bench step: where the tile search's wave cycles go, per instruction.
# Needs a library with line tables: hipcc ... -gline-tables-only -o cavif_rs_amd/libmi_lines.so.  Output: gpurun_out/TAG_pc_samples.csv.gz
TAG=${1:-pcs}; INTERVAL=${2:-4194304}
OUT=$(readlink -f gpurun_out); mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
cd /tmp; rm -rf /tmp/pcs
MI_AVIF_LIB=$ROOT/cavif_rs_amd/libmi_lines.so timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval $INTERVAL \
  --kernel-trace -d /tmp/pcs --output-format csv -- python $ROOT/bench.py --steps 1 --warmup 0 --pipeline 1 --no-cpu-baseline --no-identity-check --no-pcie-loop > $OUT/${TAG}_pc_run.log 2>&1
echo "rocprofv3 rc=$?"; tail -2 $OUT/${TAG}_pc_run.log | cut -c1-300
find /tmp/pcs -type f | head -20
F=$(find /tmp/pcs -name '*pc_sampling*stochastic*.csv' | head -1); [ -z "$F" ] && F=$(find /tmp/pcs -name '*pc_sampling*.csv' | head -1)
if [ -n "$F" ]; then ls -la $F; head -3 $F | cut -c1-600; gzip -c $F > $OUT/${TAG}_pc_samples.csv.gz; ls -la $OUT/${TAG}_pc_samples.csv.gz; fi
K=$(find /tmp/pcs -name '*kernel_trace.csv' | head -1); [ -n "$K" ] && gzip -c $K > $OUT/${TAG}_kernel_trace.csv.gz
