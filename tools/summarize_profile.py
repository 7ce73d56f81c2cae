#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<tag>/) into the small files kept under profiles/:
  <tag>_kernel_stats.csv    per-kernel calls / total / average / min / max duration (ns)
  <tag>_pmc.json            per-kernel average FETCH_SIZE / WRITE_SIZE per launch, raw and
                            corrected as MI355X_MICROARCH.md (HBM section) prescribes:
                            counters are in KiB-like units of 1024 B?  -> we keep the RAW
                            counter value, the value * 1024 (bytes if the unit is KiB) and, for
                            FETCH_SIZE, the x2 gfx950 correction for wide coalesced reads.
Usage: python tools/summarize_profile.py gpurun_out/r01a r01
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("nsp::spgemm::", "").replace("nsp::amb::", "").replace("nsp::spmv::", "")
    return name[:90]


def kernel_stats(src, dst):
    files = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        return None
    agg = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    rows = sorted(((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in agg.items()),
                  key=lambda r: -r[2])
    total = sum(r[2] for r in rows) or 1
    with open(dst, "w") as out:
        out.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for r in rows:
            out.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0 * r[2] / total:.2f}\n")
    return rows


def pmc(src, which):
    files = glob.glob(os.path.join(src, "pmc_" + which, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    rows = kernel_stats(src, os.path.join("profiles", f"{tag}_kernel_stats.csv"))
    fetch, write = pmc(src, "fetch"), pmc(src, "write")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (None, 0)), write.get(k, (None, 0))
        e = {"launches_sampled": max(f[1], w[1])}
        # calibrated on this box (profiles/r01_pmc_calibration.txt, tools/pmc_calib): 1 GiB streams
        # read at 2/4/8/16 B per lane all give 2048 B per FETCH_SIZE unit (KiB unit x the gfx950
        # half-count), writes 1024 B per WRITE_SIZE unit.
        if f[0] is not None:
            e["FETCH_SIZE_raw_per_launch"] = f[0]
            e["fetch_bytes"] = f[0] * 2048
        if w[0] is not None:
            e["WRITE_SIZE_raw_per_launch"] = w[0]
            e["write_bytes"] = w[0] * 1024
        if f[0] is not None and w[0] is not None:
            e["hbm_bytes_per_launch"] = f[0] * 2048 + w[0] * 1024
        out[k] = e
    json.dump(out, open(os.path.join("profiles", f"{tag}_pmc.json"), "w"), indent=1)
    # what bench.py reads back as roofline.traffic: the numeric kernel of every bin + SpMV
    latest = {"source": f"profiles/{tag}_pmc.json"}
    for k, e in out.items():
        if "hbm_bytes_per_launch" not in e:
            continue
        m = re.match(r"k_num_(tb|dense)<(\d+), (\d+)", k)
        if m:
            latest["spgemm_" + k.replace(" ", "")] = e["hbm_bytes_per_launch"]
        if k.startswith("k_spmv_amb"):
            latest["spmv_" + k.replace(" ", "")] = e["hbm_bytes_per_launch"]
    json.dump(latest, open(os.path.join("profiles", "pmc_latest.json"), "w"), indent=1)
    if rows:
        for r in rows[:14]:
            print(f"{r[3] / 1e3:10.1f} us avg  x{r[1]:5d}  {100.0 * r[2] / sum(x[2] for x in rows):5.1f}%  {r[0]}")
    for k, e in out.items():
        if any(s in k for s in ("k_num_", "k_sym_", "k_spmv")):
            print(k, {a: round(b) for a, b in e.items()})


if __name__ == "__main__":
    main()
