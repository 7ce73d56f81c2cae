#!/bin/bash
# serialised per-bin times (profiling mode): round-3 kernels vs the four lean builds (retry blocks / branch-free) x (grouped / pipelined)
# The three variant libraries are built HERE (no GPU needed) before the gpurun call -- they travel with the snapshot:
#   cd nsparse_amd/csrc && make -j8 OUT=$PWD/../lib_bf EXTRA=-DNSP_LEAN_RETRY_BF libs \
#     && make -j8 OUT=$PWD/../lib_pipe EXTRA=-DNSP_LEAN_PIPE libs \
#     && make -j8 OUT=$PWD/../lib_bfpipe EXTRA="-DNSP_LEAN_RETRY_BF -DNSP_LEAN_PIPE" libs
# (a variant directory that is missing is skipped)
out=gpurun_out/ab_lean3.log; : > $out
for c in ${@:-stencil webbase1m rmat18 rmat22}; do
  echo "== $c old" >> $out;  NSPARSE_TB_LEAN=0 python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
  for v in lib lib_bf lib_pipe lib_bfpipe; do
    [ -f nsparse_amd/$v/libnsparse_d.so ] || continue
    echo "== $c $v" >> $out; NSPARSE_LIB_DIR=$PWD/nsparse_amd/$v NSPARSE_TB_LEAN=3 python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
  done
done
python - <<PY
import json
lines=open("$out").read().split("\n")
i=0
while i < len(lines)-1:
    if not lines[i].startswith("=="): i+=1; continue
    try: d=json.loads(lines[i+1]); print("%-22s"%lines[i][3:], d["ms_total"], d["phase"], "sym", [x for x in d["sym_ms"][:6]], "num", [x for x in d["num_ms"][:6]]); i+=2
    except Exception: print(lines[i], "FAILED"); i+=1
PY
