#!/bin/bash
# The headline configuration (15x15, 10 blocks, 1024 games) on one / two evaluator lanes (HIP streams) x 2 / 4 pipeline groups:
# do two independent forwards interleaved on the GPU (de-phased epilogues, no launch gaps) beat back-to-back ones?
for lp in "1 2" "2 2" "2 4" "1 4" "1 2" "2 2"; do
  set -- $lp
  echo "== lanes $1, pipeline groups $2"
  python3 -c "
import sys, json; sys.path.insert(0, 'tests')
import config_table as ct
r = ct.gpu_config('resnet', 15, 5, 400, 1024, 600, lanes=$1, pipeline=$2)
print(json.dumps({k: r[k] for k in ('leaf_evals_per_s', 'ms_per_step', 'host_tree_s', 'evaluator_s', 'evaluator_lanes', 'pipeline_groups')}))
" 2>&1 | grep -v amdgpu.ids
done
