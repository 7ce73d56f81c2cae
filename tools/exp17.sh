#!/bin/bash
export TMPDIR=/tmp
NSPARSE_LIST=1 NSPARSE_TILED_PROF=1 timeout 300 python tools/one_call_cfg.py rmat22 3 2>&1 | grep -E "^\[ranked\]|serial" | tail -2 | cut -c1-700
for c in rmat22 rmat18 rmat16 webbase1m; do
  for l in 0 1 2; do
    echo "=== $c LIST=$l"; NSPARSE_LIST=$l timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
  done
done
timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -5
NSPARSE_LIST=2 timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -5
