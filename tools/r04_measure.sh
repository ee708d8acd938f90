#!/bin/bash
# Round-4 measurement batch (run ON THE GPU BOX via gpurun from the repo root); everything lands in gpurun_out/r04p/.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04p
mkdir -p $OUT
export TMPDIR=/tmp
echo "== driver command" | tee -a $OUT/log.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>> $OUT/log.txt || echo "bench failed" | tee -a $OUT/log.txt
echo "== bench trace + pmc (tools/profile_gpu.sh)" | tee -a $OUT/log.txt
bash tools/profile_gpu.sh > $OUT/profile_gpu.txt 2>&1 || echo "profile_gpu failed" | tee -a $OUT/log.txt
cd /tmp
echo "== bf16x3 line: kernel trace + stats" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bf16x3 -- python3 $ROOT/bench.py --only-bf16x3 --steps 40 --warmup 10 > $OUT/bf16x3_under_trace.json 2>> $OUT/log.txt || echo "bf16x3 trace failed" | tee -a $OUT/log.txt
echo "== latency path: kernel trace + stats" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_latency -- python3 $ROOT/tools/latency_probe.py > $OUT/latency_probe.txt 2>> $OUT/log.txt || echo "latency trace failed" | tee -a $OUT/log.txt
cd $ROOT
echo "== config table" | tee -a $OUT/log.txt
python3 tests/config_table.py > $OUT/config_table.json 2>> $OUT/log.txt || echo "config table failed" | tee -a $OUT/log.txt
echo "== train bench" | tee -a $OUT/log.txt
python3 tools/train_bench.py --batches 128,512 --steps 10 --no-torch > $OUT/train_bench.txt 2>> $OUT/log.txt || echo "train bench failed" | tee -a $OUT/log.txt
echo "== count games" | tee -a $OUT/log.txt
python3 bench.py --count-games 240 > $OUT/count_games.json 2>> $OUT/count_games.log || echo "count games failed" | tee -a $OUT/log.txt
echo "== done" | tee -a $OUT/log.txt
find $OUT -name "*kernel_stats.csv" | tee -a $OUT/log.txt
