#!/bin/bash
# round-3 experiment 2: flat walk on / off
export TMPDIR=/tmp
for c in rmat22 rmat18 webbase1m; do
  for f in 1 0; do
    echo "=== $c FLAT=$f serial"; NSPARSE_FLAT=$f timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700
    echo "=== $c FLAT=$f overlapped"; NSPARSE_FLAT=$f timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | cut -c1-330
  done
done
timeout 900 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | tail -5
