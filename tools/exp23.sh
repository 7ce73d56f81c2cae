#!/bin/bash
export TMPDIR=/tmp
for a in 64 128 256; do for b in 64 128 256; do
  echo -n "NUMD6_BS=$a NUMD7_BS=$b: "; NSPARSE_NUMD6_BS=$a NSPARSE_NUMD7_BS=$b NSPARSE_RUN_CHECK=0 timeout 120 python tools/run_configs.py cant_irr 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['gflops'])"
done; done
for k in 0 1; do echo -n "KEYED=$k: "; NSPARSE_KEYED=$k NSPARSE_RUN_CHECK=0 timeout 120 python tools/run_configs.py cant_irr 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['gflops'])"; done
for s in 128 256; do echo -n "SYMD6_BS=$s: "; NSPARSE_SYMD6_BS=$s NSPARSE_RUN_CHECK=0 timeout 120 python tools/run_configs.py cant_irr 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['gflops'])"; done
