#!/bin/bash
# serialised per-bin times (profiling mode) for: round-3 kernels, lean (per-key retry blocks), lean (branch-free retry)
out=gpurun_out/ab_lean2.log; : > $out
for c in ${@:-stencil webbase1m rmat18 rmat22}; do
  echo "== $c old" >> $out;  NSPARSE_TB_LEAN=0 python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
  echo "== $c lean" >> $out; NSPARSE_TB_LEAN=3 python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
  echo "== $c lean_bf" >> $out; NSPARSE_LIB_DIR=$PWD/nsparse_amd/lib_bf NSPARSE_TB_LEAN=3 python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
done
python - <<PY
import json
lines=open("$out").read().split("\n")
for i in range(0,len(lines)-1,2):
    try: d=json.loads(lines[i+1])
    except Exception: print(lines[i], "FAILED"); continue
    print(lines[i], d["ms_total"], d["phase"], "sym", [x for x in d["sym_ms"]], "num", [x for x in d["num_ms"]])
PY
