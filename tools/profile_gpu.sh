#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the bench
# command, then separate PMC passes (FETCH_SIZE / WRITE_SIZE never together with other traces).
# Summaries land in gpurun_out/prof/ and are copied to profiles/ by tools/summarise_profile.py.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
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
cd $ROOT
find $OUT -name "*.csv" | head -20 | tee -a $OUT/log.txt
python3 tools/summarise_profile.py $OUT > $OUT/summary.md 2>> $OUT/log.txt
tail -40 $OUT/summary.md
