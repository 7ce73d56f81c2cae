"""Three SpGEMM calls on one context, compared with the oracle, + nsparse_fused_state after each: run by
tests/test_emu_cpu.py under EMU_STALL (a workgroup of a fused tail starts late: its grid barrier times out)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
import nsparse_amd as ns
from oracle.oracle import Oracle
from gpu_util import spgemm, synth
lib, orc = ns.load("d"), Oracle("d")
A = synth(lib, 0, 6, 6, 40, seed=5)
ref = orc.spgemm(A, A)
res = []
for i in range(3):
    got, st = spgemm(lib, A)
    co, fb = C.c_int(), C.c_int()
    ok = lib.nsparse_fused_state(C.byref(co), C.byref(fb))
    good = np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"]) and orc.check_spgemm(got, ref) == 0
    res.append(dict(call=i, equal=bool(good), coresident=co.value, fallbacks=fb.value, fused_ok=ok, err=lib.nsparse_last_error()))
print(json.dumps(res))
