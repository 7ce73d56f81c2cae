#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^Read mtx" | tail -8
for c in cant cant_irr brick20 brick40 stencil webbase; do
  timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | cut -c1-200
done
