#!/bin/bash
# 8x8 split-operand convolutions (csrc/conv8_split.h) beside conv8_kernel on ONE box: per-layer launch times at 32 boards and
# BASELINE config 2 (two lanes, fresh process each) in both arithmetics.  Output: gpurun_out/conv8h_ab.log
set -e
mkdir -p gpurun_out
L=gpurun_out/conv8h_ab.log
: > $L
for a in f32 f16x2; do
  echo "== layers, trunk_arith=$a" >> $L
  python3 -c "
import sys, json; sys.path.insert(0, 'tests')
import config_table as ct
r = ct.config2_roofline(32, trunk_arith='$a')
for l in r['layers']: print(l['layer'], round(l['us_per_launch'], 2))
" >> $L 2>&1
done
for a in f32 f16x2 f32 f16x2; do
  echo "== config 2, two lanes, trunk_arith=$a" >> $L
  python3 tools/config2_run.py 6000 2 2 $a >> $L 2>&1
done
for a in f32 f16x2; do
  echo "== config 2, one lane, trunk_arith=$a" >> $L
  python3 tools/config2_run.py 6000 1 2 $a >> $L 2>&1
done
