#!/bin/bash
export TMPDIR=/tmp
for c in rmat16 rmat18 rmat22 webbase1m; do
  echo "=== $c LIST=1 overlapped"; timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | cut -c1-900
  echo "=== $c LIST=1 serial"; timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700
done
timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | tail -8
