#!/bin/bash
# PC sampling of the default bench (one batch slot) on the GPU box: tools/pc_sample.sh TAG LIB [method] [interval]
# -> gpurun_out/TAG_pcs_by_line.txt (tools/pcs_summary.py).  LIB must carry line tables (-gline-tables-only).
TAG=${1:-pcs}; LIB=${2:-cavif_rs_amd/libmi_pcs.so}; METHOD=${3:-host_trap}; IVAL=${4:-1000}
UNIT=time; [ "$METHOD" = stochastic ] && UNIT=cycles
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pcs_$TAG
rocprofv3 -L > $OUT/${TAG}_list_avail.txt 2>&1
MI_AVIF_LIB=$ROOT/$LIB timeout 280 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IVAL --kernel-trace -f csv -d /tmp/pcs_$TAG -- \
  python $ROOT/bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-identity-check > $OUT/${TAG}_pcs_run.log 2>&1
echo "rc=$?"; tail -2 $OUT/${TAG}_pcs_run.log | cut -c1-300
find /tmp/pcs_$TAG -type f | head; du -sh /tmp/pcs_$TAG
python $ROOT/tools/pcs_summary.py /tmp/pcs_$TAG > $OUT/${TAG}_pcs_by_line.txt 2> $OUT/${TAG}_pcs_summary.err
head -50 $OUT/${TAG}_pcs_by_line.txt
