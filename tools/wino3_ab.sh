#!/bin/bash
# A/B timing of trunk15_wino3_kernel build variants on ONE GPU box: tools/wino3_ab.sh build "<flags A>" "<flags B>" ...
# builds tools/_build/wino3_ab_<i>; `tools/wino3_ab.sh run <n>` (on the GPU box) runs them round-robin.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    shift; i=0; rm -f tools/_build/wino3_ab_*
    for flags in "$@"; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialphapig_amd/csrc $flags tools/wino3_bench.hip -o tools/_build/wino3_ab_$i
        echo "$flags" > tools/_build/wino3_ab_$i.flags; i=$((i+1))
    done
else
    n=${2:-3}
    for rep in 1 2; do
        for b in tools/_build/wino3_ab_?; do
            echo "== $(basename $b) [$(cat $b.flags)] rep $rep"
            APZ_AB=1 timeout -k 10 120 $b | grep "n=  512\|MISMATCH\|RESULT" | tail -5
        done
    done
fi
