#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the host library (board rules, PUCT tree pool, pure-MCTS
# rollouts) under the CPU tests that drive it -- CPU only (GPU sanitizers are not available on the MI355X pool).
# Builds a sanitized libalphapig_host.so in place, runs the tests with libasan preloaded, restores the normal build.
set -e
cd "$(dirname "$0")/.."
cp alphapig_amd/libalphapig_host.so /tmp/libalphapig_host.so.normal
trap 'cp /tmp/libalphapig_host.so.normal alphapig_amd/libalphapig_host.so' EXIT
g++ -O1 -g -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -Iinclude alphapig_amd/csrc/host_tree.cpp -o alphapig_amd/libalphapig_host.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_golden.py tests/test_selfplay_engine.py tests/test_arena.py tests/test_sgf.py tests/test_host_sampler.py -x -q
