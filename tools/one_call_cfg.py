import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth
from tools.run_configs import CASES
prec, kind, p = CASES[sys.argv[1]]
lib = ns.load(prec); A = synth(lib, kind, *p, seed=0x5EED0022)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR()
for i in range(4):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
