#!/bin/bash
# dist.all_gather_tuples of a full round (80 MB) at world size 1 on the GPU box: large block copies by one thread / by four
for t in 1 4 1 4; do
  echo "== APZ_COPY_THREADS=$t"
  APZ_COPY_THREADS=$t python3 -c "
import json, bench
o = bench.exchange_probe_world1(70175)
print(json.dumps({k: o[k] for k in ('ms', 'phases_ms_rank0')}), json.dumps({k: o['pipeline_mode'][k] for k in ('ms', 'phases_ms_rank0')}))
" 2>&1 | grep -v amdgpu.ids
done
