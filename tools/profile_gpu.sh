#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the bench
# command, then separate PMC passes (FETCH_SIZE / WRITE_SIZE never together with other traces).
# Summaries land in gpurun_out/prof/ and are copied to profiles/ by tools/summarise_profile.py.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT   # (gpurun merges files back into the local gpurun_out/: stale traces of earlier runs would be summarised along)
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-40}
CMD="python3 $ROOT/bench.py --steps $STEPS --warmup 10 --no-extras"
cd /tmp
echo "== kernel trace + stats" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_under_trace.json 2>> $OUT/log.txt || exit 1
echo "== pmc FETCH_SIZE" | tee -a $OUT/log.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2>> $OUT/log.txt || exit 1
echo "== pmc WRITE_SIZE" | tee -a $OUT/log.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2>> $OUT/log.txt || exit 1
# SQ / GRBM counters, each set in its own pass (8 SQ counters per pass; MI355X_MICROARCH.md, counter table).  A counter name
# this rocprofv3 does not know fails that pass only.
SQ_A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
SQ_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA"
if [ -z "$NO_SQ" ]; then
echo "== pmc SQ set A" | tee -a $OUT/log.txt
rocprofv3 --pmc $SQ_A --kernel-trace --output-format csv -d $OUT/pmc_sq_a -- $CMD > /dev/null 2>> $OUT/log.txt || echo "SQ set A failed" | tee -a $OUT/log.txt
echo "== pmc SQ set B" | tee -a $OUT/log.txt
rocprofv3 --pmc $SQ_B --kernel-trace --output-format csv -d $OUT/pmc_sq_b -- $CMD > /dev/null 2>> $OUT/log.txt || echo "SQ set B failed" | tee -a $OUT/log.txt
echo "== pmc GRBM" | tee -a $OUT/log.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $CMD > /dev/null 2>> $OUT/log.txt || echo "GRBM failed" | tee -a $OUT/log.txt
fi
cd $ROOT
find $OUT -name "*.csv" | head -20 | tee -a $OUT/log.txt
python3 tools/summarise_profile.py $OUT > $OUT/summary.md 2>> $OUT/log.txt
tail -40 $OUT/summary.md
