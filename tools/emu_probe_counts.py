#!/usr/bin/env python3
"""Probes per key in the LDS hash tables of the two hash-kernel families, counted on the CPU emulation (tests/emu).

    python tools/emu_probe_counts.py [case ...]      cases: web200k stencil40 rmat14   (default: all three)

The hash bins are issue-bound (HISTORY.md 4.1): what a key costs is instructions, and every probe beyond the first is a
retry round the whole wavefront sits through.  Probes per key is a property of the hash function, the table sizing and
the input -- a CPU can count it (spgemm/common.h: NSP_COUNT in ht_insert_vec / ht_find_or_insert and lean.h:
lean_insert4, compiled in under NSP_EMU only).  Each case runs through the default family (k_sym_tb / k_num_tb: 32-bit
Fibonacci hash) and the lean family (k_sym_lean / k_num_lean: 24-bit hash, NSPARSE_TB_LEAN=3) in the experiments build of
the emulation, window bins off (NSPARSE_DENSE=0) so that every row hashes; each run is checked against the oracle.
Test infrastructure: loads tests/emu/lib_exp, never the product library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tests", "emu", "lib_exp")
CASES = {"web200k": (4, 200000, 630000, 0),   # web graph, 200 K pages (the webbase-1M class generator at a fifth of the size)
         "stencil40": (1, 40, 40, 40),        # 27-point stencil 40^3: every row in the one-wavefront bins
         "rmat14": (3, 14, 16, 0)}            # power-law: keys that are mostly zero bits

CHILD = r'''
import sys, os, json, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, nsparse_amd as ns
from gpu_util import spgemm, synth
from oracle.oracle import Oracle
lib = ns.load("d"); emu = C.CDLL(os.path.join(os.environ["NSPARSE_LIB_DIR"], "libnsparse_d.so"))
emu.emu_get_fetch_counts.argtypes = [C.POINTER(C.c_longlong), C.c_int]
kind, p0, p1, p2 = %(case)r
A = synth(lib, kind, p0, p1, p2, seed=0x5EED0022)
buf = (C.c_longlong * 64)(); emu.emu_get_fetch_counts(buf, 1)
got, st = spgemm(lib, A)
emu.emu_get_fetch_counts(buf, 1)
orc = Oracle("d"); ref = orc.spgemm_omp(A, A)
ok = bool(np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"]) and orc.check_spgemm(got, dict(ref, M=A["M"])) == 0)
print(json.dumps(dict(ok=ok, M=int(A["M"]), nnz=int(len(A["col"])), nnz_c=int(got["nnz"]), tb=[int(buf[32 + i]) for i in range(3)],
                      lean=[int(buf[36 + i]) for i in range(3)], sym=list(st.sym_bin_size)[:6], num=list(st.num_bin_size)[:6])))
'''


def main():
    want = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    if not os.path.exists(os.path.join(EXP, "libnsparse_d.so")):
        sys.exit("no tests/emu/lib_exp (make -C tests/emu EXTRA=-DNSPARSE_EXPERIMENTS OUT=.../lib_exp libs)")
    print("# probes per key in the LDS hash tables, both kernel families, CPU emulation; window bins off (NSPARSE_DENSE=0)")
    print("%-10s %-22s %12s %12s %9s %9s   %s" % ("case", "family", "keys", "CAS probes", "probes/key", "retried", "rows in hash bins 0..5 (numeric)"))
    for name in want:
        for fam, lean in (("default (k_*_tb)", "0"), ("lean (k_*_lean)", "3")):
            env = dict(os.environ, NSPARSE_LIB_DIR=EXP, NSPARSE_DENSE="0", NSPARSE_TB_LEAN=lean, EMU_CLOCK_DIV="2000")
            r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, case=CASES[name])], env=env, capture_output=True, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                print(name, fam, "FAILED", r.stderr[-400:])
                sys.exit(1)
            d = json.loads(lines[-1])
            keys, probes, retried = d["lean"] if lean == "3" and d["lean"][0] else d["tb"]
            other = d["tb"] if lean == "3" else d["lean"]
            print("%-10s %-22s %12d %12d %9.3f %8.1f%%   %s%s" % (name, fam, keys, probes, probes / max(keys, 1), 100.0 * retried / max(keys, 1),
                                                               d["num"], "" if d["ok"] else "  PARITY FAILED"))
            if lean == "3" and other[0]:
                print("%-10s %-22s %12d %12d %9.3f %8.1f%%   (rows the lean family leaves to the default one)" % (
                    "", "  + default kernels", other[0], other[1], other[1] / max(other[0], 1), 100.0 * other[2] / max(other[0], 1)))
            if not d["ok"]:
                sys.exit(1)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
