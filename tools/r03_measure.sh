#!/bin/bash
# Round-3 measurement batch (run ON THE GPU BOX via gpurun from the repo root); everything lands in gpurun_out/r03p/.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
echo "== config table" | tee -a $OUT/log.txt
python3 tests/config_table.py > $OUT/config_table.json 2>> $OUT/log.txt || echo "config table failed" | tee -a $OUT/log.txt
echo "== train bench" | tee -a $OUT/log.txt
python3 tools/train_bench.py --batches 128,512 --steps 10 --no-torch > $OUT/train_bench.txt 2>> $OUT/log.txt || echo "train bench failed" | tee -a $OUT/log.txt
echo "== wgrad bench" | tee -a $OUT/log.txt
python3 tools/wgrad_bench.py > $OUT/wgrad_bench.txt 2>> $OUT/log.txt || echo "wgrad bench failed" | tee -a $OUT/log.txt
cd /tmp
echo "== wgrad pmc" | tee -a $OUT/log.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_wgrad_a -- python3 $ROOT/tools/wgrad_bench.py > /dev/null 2>> $OUT/log.txt || echo "wgrad pmc a failed" | tee -a $OUT/log.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_wgrad_b -- python3 $ROOT/tools/wgrad_bench.py > /dev/null 2>> $OUT/log.txt || echo "wgrad pmc b failed" | tee -a $OUT/log.txt
cd $ROOT
echo "== count games" | tee -a $OUT/log.txt
python3 bench.py --count-games 240 > $OUT/count_games.json 2>> $OUT/count_games.log || echo "count games failed" | tee -a $OUT/log.txt
echo "== done" | tee -a $OUT/log.txt
