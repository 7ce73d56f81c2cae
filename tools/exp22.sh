#!/bin/bash
export TMPDIR=/tmp
for c in rmat22 rmat18 webbase1m; do
for v in 0 1; do
  echo "=== $c TB_BUCKET=$v"; NSPARSE_TB_BUCKET=$v timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_total'], d['phase'], 'num', d['num_ms'][:6])"
done; done
timeout 600 python -m pytest tests/test_spgemm_gpu.py -x -q -k "bucket or rmat or power" 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -5
