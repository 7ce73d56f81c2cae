#!/bin/bash
# on the GPU box: build + two PMC passes; prints bytes-per-counter-unit for every stream kernel
export TMPDIR=/tmp
cd $(dirname $0)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 calib.hip -o /tmp/calib || exit 1
OUT=$PWD/../../gpurun_out/calib; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- /tmp/calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- /tmp/calib > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- /tmp/calib > /dev/null 2>&1
python3 - <<PY
import csv, collections
B = 1 << 30
for tag, fn in (("FETCH_SIZE", "$OUT/f/f_counter_collection.csv"), ("WRITE_SIZE", "$OUT/w/w_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        m = sum(v) / len(v)
        print("%-10s %-40s raw %12.0f  bytes/unit %8.1f" % (tag, k, m, B / m if m else float("nan")))
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/t/t_kernel_trace.csv")):
    agg[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print("%-40s %8.1f us  %7.1f GB/s" % (k, min(v) / 1e3, B / min(v)))
PY
