#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 1; do
  echo "=== LIST=$v"; NSPARSE_LIST=$v timeout 300 python tools/one_call_cfg.py rmat22 3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_total'], d['phase'], 'sym', d['sym_ms'], 'heavy', d['num_ms'][5])"
  NSPARSE_LIST=$v timeout 300 python tools/run_configs.py rmat22 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlapped', {k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
done; done
for c in rmat18 webbase1m rmat16; do for v in 0 1; do echo "=== $c LIST=$v"; NSPARSE_LIST=$v timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlapped', {k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"; done; done
timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -5
NSPARSE_LIST=2 timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -5
