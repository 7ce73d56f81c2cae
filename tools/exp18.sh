#!/bin/bash
export TMPDIR=/tmp
NSPARSE_LIST=1 timeout 300 python tools/one_call_cfg.py rmat22 3 2>&1 | tail -1 | cut -c1-700
NSPARSE_LIST=1 NSPARSE_LIST_TILES=1 timeout 300 python tools/run_configs.py rmat22 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LIST_TILES=1', {k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
timeout 300 python tools/run_configs.py rmat22 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default', {k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error|FAILED" | tail -6
