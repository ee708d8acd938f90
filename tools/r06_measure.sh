#!/bin/bash
# Round-6 measurement batch (run ON THE GPU BOX via gpurun from the repo root); everything lands in gpurun_out/r06p/.
#   tools/r06_measure.sh a    the driver's command, rocprofv3 trace + PMC passes of the bench command, kernel A/B + PMC of the trunk kernels
#   tools/r06_measure.sh b    counted games (240 s window), config table, train bench, config 2 lanes
set -o pipefail
PART=${1:-a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06p
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = a ]; then
    echo "== driver command" | tee -a $OUT/log.txt
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>> $OUT/log.txt || echo "bench failed" | tee -a $OUT/log.txt
    echo "== bench trace + pmc (tools/profile_gpu.sh)" | tee -a $OUT/log.txt
    bash tools/profile_gpu.sh > $OUT/profile_gpu.txt 2>&1 || echo "profile_gpu failed" | tee -a $OUT/log.txt
    echo "== trunk kernels side by side: tools/_build/wino3h_bench" | tee -a $OUT/log.txt
    timeout -k 10 200 tools/_build/wino3h_bench > $OUT/wino3h_bench.log 2>&1 || echo "wino3h_bench failed" | tee -a $OUT/log.txt
    echo "== PMC passes of the trunk kernels (tools/wino3h_pmc.sh)" | tee -a $OUT/log.txt
    bash tools/wino3h_pmc.sh > $OUT/wino3h_pmc.log 2>&1 || echo "wino3h_pmc failed" | tee -a $OUT/log.txt
    cp gpurun_out/w3h_pmc/summary.txt $OUT/wino3h_pmc_summary.txt 2>/dev/null
    cd /tmp
    echo "== exact-fp32 line under trace" | tee -a $OUT/log.txt
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- python3 $ROOT/bench.py --only-arith f32 --steps 40 --warmup 10 > $OUT/f32_under_trace.json 2>> $OUT/log.txt || echo "f32 trace failed" | tee -a $OUT/log.txt
    cd $ROOT
else
    echo "== count games" | tee -a $OUT/log.txt
    python3 bench.py --count-games 240 > $OUT/count_games.json 2>> $OUT/count_games.log || echo "count games failed" | tee -a $OUT/log.txt
    echo "== train bench" | tee -a $OUT/log.txt
    python3 tools/train_bench.py --batches 128,512 --steps 10 --no-torch > $OUT/train_bench.txt 2>> $OUT/log.txt || echo "train bench failed" | tee -a $OUT/log.txt
    echo "== config 2: evaluator lanes x pipeline groups" | tee -a $OUT/log.txt
    for lp in "1 2" "2 2" "1 2" "2 2"; do
        set -- $lp
        python3 tools/config2_run.py 3000 $1 $2 >> $OUT/config2_lanes.jsonl 2>> $OUT/log.txt || echo "config2 $lp failed" | tee -a $OUT/log.txt
    done
    echo "== config table" | tee -a $OUT/log.txt
    python3 tests/config_table.py > $OUT/config_table.json 2>> $OUT/log.txt || echo "config table failed" | tee -a $OUT/log.txt
fi
echo "== done $PART" | tee -a $OUT/log.txt
find $OUT -name "*kernel_stats.csv" | tee -a $OUT/log.txt
