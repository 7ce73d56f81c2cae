#!/bin/bash
export TMPDIR=/tmp
for l in 0 1; do
  echo "=== rmat22 LIST=$l"; NSPARSE_LIST=$l timeout 300 python tools/run_configs.py rmat22 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
done
NSPARSE_LIST=1 timeout 300 python tools/one_call_cfg.py rmat22 3 2>&1 | tail -1 | cut -c1-700
NSPARSE_LIST=2 timeout 300 python tools/one_call_cfg.py rmat18 3 2>&1 | tail -1 | cut -c1-700
