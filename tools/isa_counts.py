#!/usr/bin/env python3
"""Static instruction counts (gfx950 ISA, no GPU needed) of the one-wavefront hash kernels of both families:
   python tools/isa_counts.py  ->  totals per kernel and of the sort of 128 keys.
Compiles a translation unit that instantiates only those kernels (about two seconds)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "spgemm/common.h"
#include "spgemm/symbolic.h"
#include "spgemm/numeric.h"
#include "spgemm/lean.h"
namespace nsp { namespace spgemm {
template __global__ void k_num_lean<64, 256, 2>(const int *, const int *, const real *, const int *, const int *, const real *, const int *, int *, real *, const int *, const int *, const int *, int, int, int, int);
template __global__ void k_sym_lean<64, 1024, 2>(const int *, const int *, const int *, const int *, const int *, const int *, const int *, int *, int, int, int, BinState *, int *, long long *, const int *, int, int);
template __global__ void k_num_tb<64, 256, 256>(const int *, const int *, const real *, const int *, const int *, const real *, const int *, int *, real *, const int *, const int *, const int *, int, int, int, int, unsigned long long *);
template __global__ void k_sym_tb<64, 1024, false>(const int *, const int *, const int *, const int *, const int *, const int *, const int *, int *, int, int, int, BinState *, int *, int, int *, long long *, const int *, int, int);
}}
'''


def main():
    td = tempfile.mkdtemp(prefix="isa_")
    src, asm = os.path.join(td, "k.hip"), os.path.join(td, "k.s")
    open(src, "w").write(SRC)
    csrc = os.path.join(ROOT, "nsparse_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", '-DNSPARSE_SRC_HASH="x"',
                           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-DDOUBLE",
                           "--cuda-device-only", "-S", src, "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    print("%-14s %6s %6s %5s %5s %5s %5s %8s" % ("kernel", "VALU", "SALU", "LDS", "VMEM", "VGPR", "occ", "LDS B"))
    lines = text.splitlines()
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_ZN3nsp6spgemm\d+k_\w+.*:\s", ln + " ")]
    for i in starts:
        mm = re.match(r"^_ZN3nsp6spgemm\d+(k_[a-z_]+?)(I|E)", lines[i]); name = mm.group(1) if mm else lines[i][:24]
        j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        body = lines[i:j]
        cnt = lambda pre: sum(1 for ln in body if re.match(r"^\s+" + pre, ln))
        meta = "\n".join(lines[j:j + 80])
        g = lambda key: (re.search(r"; " + key + r": (\d+)", meta) or [0, "?"])[1]
        print("%-14s %6d %6d %5d %5d %5s %5s %8s" % (name, cnt("v_"), cnt("s_"), cnt("ds_"), cnt("(global|buffer)_"),
                                                   g("NumVgprs"), g("Occupancy"), g("LDSByteSize")))


if __name__ == "__main__":
    main()
