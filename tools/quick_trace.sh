#!/bin/bash
# usage (through gpurun): bash tools/quick_trace.sh case [case ...]  -> gpurun_out/qt/<case>.csv (rocprofv3 kernel stats)
export TMPDIR=/tmp NSPARSE_BIN_TIMING=${NSPARSE_BIN_TIMING:-0}
mkdir -p gpurun_out/qt
for c in "$@"; do
  rm -rf /tmp/qt_$c
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qt_$c -o t -- python tools/run_configs.py $c > /dev/null 2> gpurun_out/qt/$c.err < /dev/null
  find /tmp/qt_$c -name "*kernel_stats.csv" -exec cp {} gpurun_out/qt/$c.csv \;
  python - "$c" <<'PY'
import csv, sys
c = sys.argv[1]
try:
    rows = list(csv.DictReader(open(f"gpurun_out/qt/{c}.csv")))
except OSError:
    print(c, "no stats"); sys.exit(0)
for r in rows[:18]:
    print(c, r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1000, 1), "us")
PY
done
