#!/bin/bash
# Round-end evidence run (one gpurun call): kernel-trace stats of the default bench, and HBM counters (separate --pmc passes).
# Usage: tools/final_profile.sh TAG     -> gpurun_out/TAG_*  (copy what should be judged into profiles/)
TAG=${1:-r01}
OUT=$(readlink -f gpurun_out); mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for P in 4 1; do
  rm -rf /tmp/prof_$P
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$P -- python $ROOT/bench.py --steps 3 --warmup 1 --pipeline $P --no-cpu-baseline --no-pcie-loop --end-to-end 0 --no-threads-line > $OUT/${TAG}_bench_under_rocprof_pipeline$P.log 2>&1
  DB=$(find /tmp/prof_$P -name '*.db' | head -1)
  if [ -n "$DB" ]; then python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats_pipeline$P.txt; fi
  find /tmp/prof_$P -name '*kernel_stats*.csv' -exec cp {} $OUT/${TAG}_kernel_stats_pipeline$P.csv \;
  tail -1 $OUT/${TAG}_bench_under_rocprof_pipeline$P.log | cut -c1-400
done
cd $ROOT
tools/pmc_collect.sh $OUT/${TAG}_pmc "python bench.py --steps 1 --warmup 1 --pipeline 1 --no-cpu-baseline --no-identity-check --no-pcie-loop --end-to-end 0 --no-threads-line" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
python tools/pmc_summary.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc_summary.json
rm -rf $OUT/${TAG}_pmc/pass*/
python tools/make_hbm_counters.py $OUT/${TAG}_pmc_summary.json $TAG > $OUT/${TAG}_hbm_counters.json
python tools/make_valu_counters.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_kernel_stats_pipeline1.txt $TAG > $OUT/${TAG}_valu_counters.json
head -c 600 $OUT/${TAG}_pmc_summary.json
