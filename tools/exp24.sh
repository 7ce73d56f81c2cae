#!/bin/bash
export TMPDIR=/tmp
for kb in 0 1; do for w in 2 8; do
  echo -n "KEYED_B=$kb world=$w kind 5: "; NSPARSE_KEYED_B=$kb timeout 300 python tools/emulate_rank.py $w $((w/2)) 5 2>&1 | tail -1
done; done
echo -n "world=1 (A = B): "; timeout 300 python tools/emulate_rank.py 1 0 5 2>&1 | tail -1
echo -n "world=8 brick: "; timeout 300 python tools/emulate_rank.py 8 4 0 2>&1 | tail -1
timeout 900 python -m pytest tests/test_partition_gpu.py tests/test_spgemm_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error" | tail -4
