#!/bin/bash
# tools/pmc_collect.sh OUTDIR "CMD" "COUNTERS PASS 1" ["COUNTERS PASS 2" ...]
# One rocprofv3 --pmc pass per counter group (kernel-trace only: no sys/hip trace domains alongside --pmc),
# csv output under OUTDIR/passN; summarise with tools/pmc_summary.py OUTDIR.
set -u
OUT=$(readlink -f "$1"); CMD=$2; shift 2
mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
