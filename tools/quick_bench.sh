#!/bin/bash
# usage: bash tools/quick_bench.sh case [case ...]   (env passes through; per-bin events on)
for c in "$@"; do NSPARSE_BIN_TIMING=${NSPARSE_BIN_TIMING:-1} timeout 120 python tools/run_configs.py $c 2>/dev/null | grep "^{" | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print(d['case'], d['ms'], d['gflops'], d['phase'], 'sym', [x for x in d['sym_ms'] if x>0], 'num', [x for x in d['num_ms'] if x>0], d.get('rpt_ok'), d.get('col_ok'), d.get('val_fails'))"; done
