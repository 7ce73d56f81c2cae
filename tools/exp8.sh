#!/bin/bash
export TMPDIR=/tmp
for c in rmat22 rmat18 webbase1m; do
  for p in 0 1 2 4; do
    echo "=== $c TB_PERSIST=$p"; NSPARSE_TB_PERSIST=$p NSPARSE_RUN_CHECK=$([ $p = 2 ] && echo 1 || echo 0) timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
  done
  NSPARSE_TB_PERSIST=2 timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700
done
timeout 900 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | tail -5
NSPARSE_LIST=1 timeout 900 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | tail -5
