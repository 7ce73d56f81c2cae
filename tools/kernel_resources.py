#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of every HIP translation unit of the product, from the
compiler's own remarks (-Rpass-analysis=kernel-resource-usage; no GPU needed):
    python tools/kernel_resources.py [d|s] [--unit spgemm_hash|spmv_amb|amb_convert|dist_spmv|all] [--own] [filter ...]
    python tools/kernel_resources.py [d|s] --lds-table FILE [--extra "-DFLAG ..."]
        "<mangled name> <static LDS bytes> <occupancy, waves/SIMD> <VGPRs>" per kernel of every unit: what tests/emu loads
        (next to its library, <lib>.lds) to police LDS capacity and derive co-residency -- the emulation itself cannot see
        how much static LDS a kernel declares
e.g. python tools/kernel_resources.py d --unit spmv_amb k_spmv_amb_row
A kernel with ScratchSize > 0 spills: in the latency-bound row kernels that has always cost more than it bought, and
an HBM-bound kernel that spills writes and re-reads its own operands.  tests/test_kernel_resources.py gates on it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ("spgemm_hash", "spmv_amb", "amb_convert", "dist_spmv")


def unit_resources(prec, unit, extra=()):
    """[{name (demangled), own, vgpr, agpr, sgpr, scratch, occ, lds, vspill, sspill}] of one unit's device code."""
    src = os.path.join(ROOT, "nsparse_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", '-DNSPARSE_SRC_HASH="x"',
           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + src,
           "-DDOUBLE" if prec == "d" else "-DFLOAT", *extra, "--cuda-device-only", "-c", os.path.join(src, unit + ".hip"),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (unit, p.stderr[-2000:]))
    cur = None
    rows = []
    for ln in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    out = []
    for r, n in zip(rows, names):
        n = n.replace("void ", "", 1) if n.startswith("void ") else n
        # template arguments stay (they tell the instantiations apart), the parameter list goes
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        full = n[:cut]
        own = full.startswith("nsp::") or full.startswith("k_")

        def num(key):
            try:
                return int(r.get(key, "0"))
            except ValueError:
                return 0
        out.append({"name": full, "mangled": r["mangled"], "own": own, "unit": unit, "vgpr": num("VGPRs"), "agpr": num("AGPRs"),
                    "sgpr": num("TotalSGPRs"), "scratch": num("ScratchSize [bytes/lane]"),
                    "occ": num("Occupancy [waves/SIMD]"), "lds": num("LDS Size [bytes/block]"),
                    "vspill": num("VGPRs Spill"), "sspill": num("SGPRs Spill")})
    return out


def main():
    args = sys.argv[1:]
    prec = "d"
    units = ["spgemm_hash"]
    own_only = False
    filt = []
    table = None
    extra = []
    i = 0
    while i < len(args):
        a = args[i]
        if a in ("d", "s"):
            prec = a
        elif a == "--unit":
            i += 1
            units = list(UNITS) if args[i] == "all" else [args[i]]
        elif a == "--own":
            own_only = True
        elif a == "--lds-table":
            i += 1
            table = args[i]
        elif a == "--extra":
            i += 1
            extra = [f for f in args[i].split() if f.startswith("-D")]
        else:
            filt.append(a)
        i += 1
    if table:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as ex:
            res = list(ex.map(lambda u: unit_resources(prec, u, extra), UNITS))
        with open(table + ".tmp", "w") as f:
            for rows in res:
                for r in rows:
                    f.write("%s %d %d %d\n" % (r["mangled"], r["lds"], r["occ"], r["vgpr"] + r["agpr"]))
        os.replace(table + ".tmp", table)
        return
    print("%-88s %5s %5s %7s %4s %7s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS", "vspill", "sspill"))
    n_own = n_scratch = 0
    for u in units:
        print("# unit %s.hip (%s)" % (u, "double" if prec == "d" else "float"))
        for r in unit_resources(prec, u):
            if own_only and not r["own"]:
                continue
            short = r["name"].replace("nsp::spgemm::", "").replace("nsp::spmv::", "").replace("nsp::amb::", "").replace("nsp::", "")
            if filt and not any(f in short for f in filt):
                continue
            n_own += r["own"]
            n_scratch += r["own"] and r["scratch"] > 0
            print("%-88s %5d %5d %7d %4d %7d %6d %6d" % (short[:88], r["vgpr"] + r["agpr"], r["sgpr"], r["scratch"], r["occ"],
                                                         r["lds"], r["vspill"], r["sspill"]))
    print("# own kernels: %d, with scratch: %d" % (n_own, n_scratch))


if __name__ == "__main__":
    main()
