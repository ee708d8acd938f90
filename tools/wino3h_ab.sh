#!/bin/bash
# A/B timing of trunk15_wino3h_kernel build variants on ONE GPU box: tools/wino3h_ab.sh build "<flags A>" "<flags B>" ...
# builds tools/_build/wino3h_ab_<i>; `tools/wino3h_ab.sh run` (on the GPU box) runs them round-robin, twice.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    shift; i=0; rm -f tools/_build/wino3h_ab_*
    for flags in "$@"; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialphapig_amd/csrc $flags tools/wino3h_bench.hip -o tools/_build/wino3h_ab_$i &
        echo "$flags" > tools/_build/wino3h_ab_$i.flags; i=$((i+1))
        if [ $((i % 6)) = 0 ]; then wait; fi
    done
    wait
else
    for rep in 1 2; do
        for b in tools/_build/wino3h_ab_?; do
            echo "== $(basename $b) [$(cat $b.flags)] rep $rep"
            timeout -k 10 120 $b | grep "n=  512\|n=  128\|n= 1024\|MISMATCH\|MISSED\|RESULT\|stamps" | grep -v "^check.*OK" | tail -14
        done
    done
fi
