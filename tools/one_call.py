import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth
lib = ns.load("d"); A = synth(lib, 0, 9, 9, 257, seed=0x5EED0022)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR()
for i in range(6):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
