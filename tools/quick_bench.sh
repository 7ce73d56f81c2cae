#!/bin/bash
# quick A/B of tuning knobs on the GPU box: prints GFLOPS / ms / phase times per setting
run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu --no-large 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms']
print(d['value'], 'GFLOPS', d['ms_per_step'], 'ms | setup', p['setup'], 'sym', p['symbolic'], p['symbolic_bins'], 'num', p['numeric'], p['numeric_bins'], '| spmv', d['spmv']['ms_per_spmv'], d['spmv']['value'])"; }
for cfg in "$@"; do run $cfg; done
