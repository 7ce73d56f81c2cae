#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of the hash SpGEMM translation unit, from the compiler's
own remarks (-Rpass-analysis=kernel-resource-usage; no GPU needed):
    python tools/kernel_resources.py [d|s] [filter ...]      e.g.  python tools/kernel_resources.py d k_num_tb k_num_block
A kernel with ScratchSize > 0 spills: in the latency-bound row kernels that has always cost more than it bought."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("d", "s") else "d"
    filt = [a for a in sys.argv[1:] if a not in ("d", "s")]
    src = os.path.join(ROOT, "nsparse_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", '-DNSPARSE_SRC_HASH="x"',
           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + src,
           "-DDOUBLE" if prec == "d" else "-DFLOAT", "--cuda-device-only", "-c", os.path.join(src, "spgemm_hash.hip"),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for ln in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            mangled = t.split(":", 1)[1].strip()
            cur = {"name": mangled}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    print("%-64s %5s %5s %7s %4s %7s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS", "vspill", "sspill"))
    for r, n in zip(rows, names):
        short = re.sub(r"\(.*$", "", n).replace("void ", "").replace("nsp::spgemm::", "")
        if filt and not any(f in short for f in filt):
            continue
        print("%-64s %5s %5s %7s %4s %7s %6s %6s" % (short[:64], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                     r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"),
                                                     r.get("VGPRs Spill"), r.get("SGPRs Spill")))


if __name__ == "__main__":
    main()
