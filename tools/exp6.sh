#!/bin/bash
export TMPDIR=/tmp
for c in rmat18 rmat22; do
  for l in 1 0; do
  W=/tmp/t_$c_$l; rm -rf $W
  NSPARSE_LIST=$l timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W -o t -- python tools/one_call_cfg.py $c 3 > /dev/null 2> /tmp/err.txt
  echo "=== $c LIST=$l"
  f=$(find $W -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    k = re.sub(r"\(.*$", "", r["Name"]).replace("void ", "").replace("nsp::spgemm::", "")
    print(f"  {k[:55]:55s} n={r['Calls']:>3s} avg={float(r['AverageNs'])/1e3:10.1f} us")
PY
  done
done
