#!/bin/bash
# same-box A/B of one environment switch:  bash tools/ab.sh VAR "v1 v2 ..." case [case ...]   (3 repetitions each)
export TMPDIR=/tmp
VAR=$1; VALS=$2; shift 2
for rep in 1 2 3; do for v in $VALS; do for c in "$@"; do
  echo -n "$VAR=$v $c: "; env $VAR=$v NSPARSE_RUN_CHECK=${NSPARSE_RUN_CHECK:-0} timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['gflops'], d.get('rpt_ok'), d.get('col_ok'), d.get('val_fails'))"
done; done; done
