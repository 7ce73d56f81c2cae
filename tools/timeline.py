#!/usr/bin/env python3
"""Kernel timeline of the LAST spgemm_kernel_hash call in a rocprofv3 kernel trace CSV:
start offset, duration and gap to the previous kernel (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last k_b_info marks the start of the last call
idx = max(i for i, n in enumerate(names) if "k_b_info" in n)
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:idx + 60]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nsp::spgemm::", "")[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = max(prev_end, e)
    if "k_publish" in n:
        npub = globals().get("npub", 0) + 1
        globals()["npub"] = npub
        if npub == 3:
            break
