#!/bin/bash
export TMPDIR=/tmp
for c in rmat22 rmat18 webbase1m; do
  echo "=== $c FLAT=2 serial"; NSPARSE_FLAT=2 timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700
  echo "=== $c FLAT=2 overlapped"; NSPARSE_FLAT=2 timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | cut -c1-330
done
timeout 300 python tools/alloc_modes.py cant cant_irr 2>&1 | tail -3
timeout 900 python -m pytest tests/test_samples_gpu.py tests/test_dist_native_gpu.py -x -q -s 2>&1 | grep -vE "^Read mtx" | tail -12
