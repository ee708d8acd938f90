#!/bin/bash
# Build the ablation variants of trunk15_wino_kernel (run on the CPU container; binaries travel with gpurun).
set -e
cd "$(dirname "$0")/.."
for v in base "$@"; do
  flags=""
  [ "$v" != base ] && for m in ${v//+/ }; do case $m in PF*) flags="$flags -DAPZ_WINO_PF=${m#PF}";; V1) flags="$flags -DAPZ_HARNESS_V1";; STAMPS) flags="$flags -DAPZ_WINO_STAMPS";; TWOSTREAMS) flags="$flags -DAPZ_HARNESS_TWO_STREAMS";; WIN*) flags="$flags -DAPZ_WINO2_WIN=${m#WIN}";; GSH*) flags="$flags -DAPZ_WINO2_GSH=${m#GSH}";; *) flags="$flags -DAPZ_WINO_ABL_$m";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Ialphapig_amd/csrc $flags tools/wino_ablate.hip -o tools/_build/wino_$v 2>&1 | grep -E "error|spill|Scratch" || true
done
