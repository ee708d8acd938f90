#!/bin/bash
# Round-5 measurement batch (run ON THE GPU BOX via gpurun from the repo root); everything lands in gpurun_out/r05p/.
#   tools/r05_measure.sh [quick]     quick: no count-games window
set -o pipefail
MODE=${1:-full}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
echo "== driver command" | tee -a $OUT/log.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>> $OUT/log.txt || echo "bench failed" | tee -a $OUT/log.txt
echo "== bench trace + pmc (tools/profile_gpu.sh)" | tee -a $OUT/log.txt
bash tools/profile_gpu.sh > $OUT/profile_gpu.txt 2>&1 || echo "profile_gpu failed" | tee -a $OUT/log.txt
cd /tmp
echo "== stem at the north_star shape: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (separate passes)" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stem_trace -- python3 $ROOT/tools/stem_profile.py > $OUT/stem_hip_events.json 2>> $OUT/log.txt || echo "stem trace failed" | tee -a $OUT/log.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/stem_pmc_fetch -- python3 $ROOT/tools/stem_profile.py > /dev/null 2>> $OUT/log.txt || echo "stem fetch failed" | tee -a $OUT/log.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/stem_pmc_write -- python3 $ROOT/tools/stem_profile.py > /dev/null 2>> $OUT/log.txt || echo "stem write failed" | tee -a $OUT/log.txt
echo "== config 2: evaluator lanes x pipeline groups" | tee -a $OUT/log.txt
cd $ROOT
for lp in "1 2" "2 2" "2 4" "4 4" "1 2" "2 2"; do
    set -- $lp
    python3 tools/config2_run.py 3000 $1 $2 >> $OUT/config2_lanes.jsonl 2>> $OUT/log.txt || echo "config2 $lp failed" | tee -a $OUT/log.txt
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_config2 -- python3 $ROOT/tools/config2_run.py 1500 2 2 > $OUT/config2_under_trace.json 2>> $OUT/log.txt || echo "config2 trace failed" | tee -a $OUT/log.txt
echo "== bf16x3 line: kernel trace + stats" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bf16x3 -- python3 $ROOT/bench.py --only-bf16x3 --steps 40 --warmup 10 > $OUT/bf16x3_under_trace.json 2>> $OUT/log.txt || echo "bf16x3 trace failed" | tee -a $OUT/log.txt
cd $ROOT
echo "== config table" | tee -a $OUT/log.txt
python3 tests/config_table.py > $OUT/config_table.json 2>> $OUT/log.txt || echo "config table failed" | tee -a $OUT/log.txt
echo "== train bench" | tee -a $OUT/log.txt
python3 tools/train_bench.py --batches 128,512 --steps 10 --no-torch > $OUT/train_bench.txt 2>> $OUT/log.txt || echo "train bench failed" | tee -a $OUT/log.txt
if [ "$MODE" != quick ]; then
    echo "== count games" | tee -a $OUT/log.txt
    python3 bench.py --count-games 240 > $OUT/count_games.json 2>> $OUT/count_games.log || echo "count games failed" | tee -a $OUT/log.txt
fi
echo "== done" | tee -a $OUT/log.txt
find $OUT -name "*kernel_stats.csv" | tee -a $OUT/log.txt
