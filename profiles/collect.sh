#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + separate PMC passes of the bench command, raw output under
# gpurun_out/prof_$TAG/, condensed by profiles/summarize.py into gpurun_out/prof_$TAG/summary/ (copy what you want
# judged into profiles/).  Counters are collected in their own runs (one --pmc group per run), never together with
# tracing domains other than --kernel-trace.
#   usage: bash profiles/collect.sh TAG [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:---workload ${WORKLOAD:-cfg2} --steps 8 --warmup 8 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel trace ($ARGS)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $ARGS > $OUT/kt.log 2>&1
tail -2 $OUT/kt.log
PMC_ARGS="--workload ${WORKLOAD:-cfg2} --steps ${PMC_STEPS:-8} --warmup 0 --no-cpu-baseline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  echo "== pmc pass $i: $grp"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py $PMC_ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
  tail -1 $OUT/pmc$i.log | cut -c1-200
done
python3 profiles/summarize.py $OUT
