#!/bin/bash
# round-3 experiment 1: dist tests, in-kernel phase timers of the hash bins and the heavy kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/exp1
timeout 600 python -m pytest tests/test_dist_native_gpu.py -x -q -s 2>&1 | grep -vE "^Read mtx" | tail -15
for c in rmat22 rmat18 webbase1m stencil; do
  echo "=== $c TB_PROF"
  NSPARSE_TB_PROF=1 NSPARSE_TILED_PROF=1 timeout 300 python tools/one_call_cfg.py $c 2 2>&1 | grep -E "^\[tb\]|^\[tiled\]|^\[ranked\]" | tail -8
  echo "=== $c unsorted"
  NSPARSE_UNSORTED=1 timeout 300 python tools/one_call_cfg.py $c 2 2>&1 | tail -1 | cut -c1-600
done
