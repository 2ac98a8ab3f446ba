#!/bin/bash
export TMPDIR=/tmp; R=$(pwd)
for lib in ${PROBE_LIBS:-cur cdeflb3}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$lib_$C; (cd /tmp && MI_AVIF_LIB=$R/cavif_rs_amd/libmi_v_$lib.so rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm_${lib}_$C -- python $R/bench.py --steps 1 --warmup 1 --pipeline 1 --no-cpu-baseline --no-identity-check --no-pcie-loop --end-to-end 0 --no-threads-line > /dev/null 2>&1)
    python $R/tools/pmc_summary.py /tmp/pm_${lib}_$C | python -c "
import sys,json; d=json.load(sys.stdin)
for k,v in d.items():
    if '${PROBE_KERNEL:-cdef}' in k: print('$lib', '$C', v)"
  done
done
