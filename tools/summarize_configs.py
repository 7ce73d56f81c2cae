#!/usr/bin/env python3
"""Condense the output of tools/profile_configs.sh (gpurun_out/<tag>/) into what is kept under profiles/:

  profiles/<tag>_configs.jsonl        one line per case: the whole-call time with the bins overlapped (what a
                                      caller gets), the serialised per-bin times, and a `roofline` block for the
                                      case's dominant kernel and for the whole call
  profiles/<tag>_<case>_kernels.csv   per kernel of one serialised call: calls, average duration (rocprofv3
                                      --kernel-trace --stats), HBM bytes per launch from the FETCH_SIZE /
                                      WRITE_SIZE passes (corrected as bench.py does: 2048 B / 1024 B per unit,
                                      profiles/r01_pmc_calibration.txt), SQ wait / issue shares and instruction counts

Roofline model per case: compulsory HBM bytes of C = A * A = (4 + w) (nnz A + nnz B + nnz C) + 12 M,
over the whole-call time, against 8 TB/s; per kernel the MEASURED traffic over its duration is given beside it
(traffic far above the compulsory bytes = re-reads that missed the L2s).
Usage: python tools/summarize_configs.py gpurun_out/r03 r03 [case ...]
"""
import csv
import glob
import json
import os
import re
import sys

HBM = 8000.0
FETCH_UNIT, WRITE_UNIT = 2048, 1024


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("nsp::spgemm::", "")


def main():
    src, tag = sys.argv[1], sys.argv[2]
    cases = sys.argv[3:] or sorted(os.path.basename(f)[:-len(".call.json")] for f in glob.glob(os.path.join(src, "*.call.json")))
    os.makedirs("profiles", exist_ok=True)
    lines = []
    for c in cases:
        try:
            call = json.loads(open(os.path.join(src, c + ".call.json")).read().strip().splitlines()[-1])
            serial = json.loads(open(os.path.join(src, c + ".serial.json")).read().strip().splitlines()[-1])
        except (OSError, IndexError, ValueError) as e:
            print(f"{c}: missing ({e})")
            continue
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(src, c + ".pmc.json")))
        except (OSError, ValueError):
            pass
        rows = []
        try:
            rows = list(csv.DictReader(open(os.path.join(src, c + ".stats.csv"))))
        except OSError:
            pass
        w = 4 if call.get("prec") == "s" else 8
        kern = []
        for r in rows:
            k = short(r["Name"])
            if not k.startswith("k_") and "rocprim" not in k:
                continue
            d = pmc.get(k, {})
            avg_us = float(r["AverageNs"]) / 1e3
            fb = d.get("FETCH_SIZE", 0.0) * FETCH_UNIT
            wb = d.get("WRITE_SIZE", 0.0) * WRITE_UNIT
            wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            kern.append(dict(kernel=k, calls=int(r["Calls"]), avg_us=round(avg_us, 1), fetch_mb=round(fb / 1e6, 1),
                             write_mb=round(wb / 1e6, 1),
                             hbm_gbs=round((fb + wb) / (avg_us * 1e-6) / 1e9, 1) if avg_us > 0 else 0.0,
                             wait_any=round(d.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                             active_any=round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                             valu_m=round(d.get("SQ_INSTS_VALU", 0.0) / 1e6, 1), salu_m=round(d.get("SQ_INSTS_SALU", 0.0) / 1e6, 1),
                             lds_m=round(d.get("SQ_INSTS_LDS", 0.0) / 1e6, 1), vmem_rd_m=round(d.get("SQ_INSTS_VMEM_RD", 0.0) / 1e6, 2),
                             vmem_wr_m=round(d.get("SQ_INSTS_VMEM_WR", 0.0) / 1e6, 2)))
        kern.sort(key=lambda x: -x["avg_us"] * x["calls"])
        with open(os.path.join("profiles", f"{tag}_{c}_kernels.csv"), "w") as f:
            cols = list(kern[0].keys()) if kern else ["kernel"]
            f.write(",".join(cols) + "\n")
            for k in kern[:24]:
                f.write(",".join(f'"{k[q]}"' if q == "kernel" else str(k[q]) for q in cols) + "\n")
        comp = (4 + w) * (2 * call["nnzA"] + call["nnzC"]) + 12 * call["M"]
        ms = call["ms"]
        dom = kern[0] if kern else None
        roof = {"bound": "hbm", "peak": HBM, "unit": "GB/s",
                "whole_call": {"compulsory_bytes": comp, "ms": ms, "achieved": round(comp / (ms * 1e-3) / 1e9, 1),
                               "frac": round(comp / (ms * 1e-3) / 1e9 / HBM, 4),
                               "model": "(4+w)(nnz A + nnz B + nnz C) + 12 M: every array once"},
                "dominant_kernel": (dict(kernel=dom["kernel"], avg_us=dom["avg_us"],
                                         measured_hbm_bytes=int((dom["fetch_mb"] + dom["write_mb"]) * 1e6),
                                         measured_hbm_gbs=dom["hbm_gbs"], measured_frac=round(dom["hbm_gbs"] / HBM, 4),
                                         wait_any=dom["wait_any"], active_any=dom["active_any"]) if dom else None)}
        lines.append(dict(case=c, prec=call.get("prec"), M=call["M"], nnzA=call["nnzA"], n_prod=call["n_prod"],
                          nnzC=call["nnzC"], ms_overlapped=ms, gflops=call["gflops"],
                          structure_ok=bool(call.get("rpt_ok")) and bool(call.get("col_ok")), val_fails=call.get("val_fails"),
                          serial_ms_total=serial["ms_total"], serial_phase_ms=serial["phase"],
                          sym_bins=serial["sym_bins"], num_bins=serial["num_bins"], sym_ms=serial["sym_ms"],
                          num_ms=serial["num_ms"], roofline=roof))
        print(c, ms, "ms", roof["whole_call"]["frac"], "|", dom["kernel"] if dom else None, dom["avg_us"] if dom else None)
    with open(os.path.join("profiles", f"{tag}_configs.jsonl"), "w") as f:
        for ln in lines:
            f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
