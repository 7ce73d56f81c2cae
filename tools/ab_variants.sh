#!/bin/bash
# A/B of the opt-in kernel families against the default path, serialised per-bin times (profiling mode), ONE library:
# nsparse_amd/lib_exp (the -DNSPARSE_EXPERIMENTS build that __graft_entry__.build() makes; it reads the switches and
# carries every form as a template instantiation -- round 5 needed four library builds for the same table).
#   default        the product's kernels (round-3 hash kernels k_sym_tb / k_num_tb, cursor heavy-row kernels)
#   lean           NSPARSE_TB_LEAN=3      lean.h in the hash bins 1-4 of both phases (retry blocks, grouped walk)
#   lean_bf        NSPARSE_TB_LEAN=7      ... with branch-free retry rounds
#   lean_pipe      NSPARSE_TB_LEAN=11     ... with the pipelined walk
#   lean_bfpipe    NSPARSE_TB_LEAN=15     ... with both
#   flat           NSPARSE_HEAVY_FLAT=7   stateless heavy-row tiles over the panel table (heavy_flat.h): dense + ranked + symbolic
#   flat_dense     NSPARSE_HEAVY_FLAT=1   only the dense tiles
#   flat_ranked    NSPARSE_HEAVY_FLAT=2   only the list-driven ranked tiles
#   flat_sym       NSPARSE_HEAVY_FLAT=4   only the symbolic twin (windows wider than 2^20 columns)
#   bash tools/ab_variants.sh [case ...]      cases of tools/one_call_cfg.py (default: stencil webbase1m rmat18 rmat22)
out=gpurun_out/ab_variants.log; : > $out
EXP=$PWD/nsparse_amd/lib_exp
[ -f $EXP/libnsparse_d.so ] || { echo "no $EXP (python -c 'import __graft_entry__ as g; g.build()')"; exit 1; }
run() {  # <case> <label> <env...>
  local c=$1 l=$2; shift 2
  echo "== $c $l" >> $out
  env NSPARSE_LIB_DIR=$EXP "$@" python tools/one_call_cfg.py $c 2>/dev/null | grep "^{" >> $out
}
for c in ${@:-stencil webbase1m rmat18 rmat22}; do
  run $c default NSPARSE_TB_LEAN=0
  case $c in
    rmat*) run $c flat NSPARSE_HEAVY_FLAT=7; run $c flat_dense NSPARSE_HEAVY_FLAT=1; run $c flat_ranked NSPARSE_HEAVY_FLAT=2; run $c flat_sym NSPARSE_HEAVY_FLAT=4 ;;
  esac
  run $c lean NSPARSE_TB_LEAN=3
  run $c lean_bf NSPARSE_TB_LEAN=7
  run $c lean_pipe NSPARSE_TB_LEAN=11
  run $c lean_bfpipe NSPARSE_TB_LEAN=15
done
python - <<PY
import json
lines=open("$out").read().split("\n")
i=0
while i < len(lines)-1:
    if not lines[i].startswith("=="): i+=1; continue
    try: d=json.loads(lines[i+1]); print("%-24s"%lines[i][3:], d["ms_total"], d["phase"], "sym", [x for x in d["sym_ms"][:6]], "num", [x for x in d["num_ms"][:6]]); i+=2
    except Exception: print(lines[i], "FAILED"); i+=1
PY
