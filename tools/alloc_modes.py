"""Whole-call time of spgemm_kernel_hash under the three workspace modes (nsparse_set_workspace_cache):
1 block cache (default), 0 hipMalloc / hipFree inside the call (reference protocol), 2 hipMallocAsync /
hipFreeAsync inside the call.  python tools/alloc_modes.py [case ...]"""
import ctypes as C, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import nsparse_amd as ns
from gpu_util import synth
from tools.run_configs import CASES
for case in (sys.argv[1:] or ["cant", "cant_irr"]):
    prec, kind, p = CASES[case]
    lib = ns.load(prec)
    A = synth(lib, kind, *p, seed=0x5EED0022)
    a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR()
    out = {"case": case}
    for tag, mode in (("cache", 1), ("malloc", 0), ("async", 2), ("cache_again", 1)):
        lib.nsparse_set_workspace_cache(mode)
        for i in range(3):
            lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
        ts = []
        for i in range(20):
            t = time.perf_counter(); lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); ts.append(time.perf_counter() - t); lib.release_csr(c)
        out[f"{tag}_ms"] = round(float(np.mean(ts)) * 1e3, 4)
        out[f"{tag}_min_ms"] = round(float(np.min(ts)) * 1e3, 4)
    lib.nsparse_set_workspace_cache(1)
    print(json.dumps(out))
