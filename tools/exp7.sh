#!/bin/bash
export TMPDIR=/tmp
for c in rmat22 rmat18; do
  for lw in 0 60000 150000 400000; do
    echo "=== $c LIST_WORK=$lw"; NSPARSE_LIST=$([ $lw = 0 ] && echo 0 || echo 1) NSPARSE_LIST_WORK=$lw NSPARSE_RUN_CHECK=$([ $lw = 150000 ] && echo 1 || echo 0) timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
  done
  NSPARSE_LIST_WORK=150000 timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700
done
for c in rmat16 webbase1m; do
  for l in 0 1; do echo "=== $c LIST=$l"; NSPARSE_LIST=$l timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"; done
done
W=/tmp/t_x; rm -rf $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W -o t -- python tools/one_call_cfg.py rmat22 3 > /dev/null 2> /tmp/err.txt
f=$(find $W -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    k = re.sub(r"\(.*$", "", r["Name"]).replace("void ", "").replace("nsp::spgemm::", "")
    print(f"  {k[:55]:55s} n={r['Calls']:>3s} avg={float(r['AverageNs'])/1e3:10.1f} us")
PY
