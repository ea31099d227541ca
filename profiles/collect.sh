#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + separate PMC passes of one command, raw output under
# gpurun_out/prof_$TAG/, condensed by profiles/summarize.py into gpurun_out/prof_$TAG/summary/ (copy what you want
# judged into profiles/rNN/, and merge summary/roofs.json into profiles/roofs.json, which bench.py reads).
# Counters are collected in their own runs (one --pmc group per run), never together with tracing domains other than
# --kernel-trace.  The program itself follows `--` (python3 ...): no env / shell hop under the profiler.
#   usage: bash profiles/collect.sh TAG [--script tools/x.py args... | bench args...]
#   env  : ROOF_KEY  launch-shape key of the loss kernel in this command, e.g. cfg2/poses256/f16 (-> roofs.json)
#          POINT_POSES  point-poses per launch of that kernel (N x poses), for the VALU instructions per point-pose
set -u
TAG=${1:-r02}; shift || true
if [ "${1:-}" = "--script" ]; then shift; CMD="$*"; PMC_CMD="$*"; else
  ARGS=${*:---workload ${WORKLOAD:-cfg2} --steps 8 --warmup 8 --no-cpu-baseline}
  # one launch shape per run: without the single-image pass the kernel-stats average IS the headline launch's duration
  CMD="bench.py $ARGS --no-single-image --no-also"
  # counter passes: one cold pass is enough (every dispatch is serialised by the counter reads), no warm-up steps
  PMC_CMD="bench.py $ARGS --min-seconds 0 --no-single-image --no-also --warmup 0 --prewarm-ms 0"
fi
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel trace (python3 $CMD)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $CMD > $OUT/kt.log 2>&1
tail -2 $OUT/kt.log | cut -c1-300
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_ATOMIC_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  echo "== pmc pass $i: $grp"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $PMC_CMD > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
  tail -1 $OUT/pmc$i.log | cut -c1-160
done
python3 profiles/summarize.py $OUT
# the raw rocprofv3 trees are tens of MB per pass: only the summaries and the logs travel back
rm -rf $OUT/kt $OUT/pmc[0-9]
