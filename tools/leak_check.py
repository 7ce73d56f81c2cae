"""Device memory after repeated SpGEMM calls and AMB conversions must not drift (block cache), and trim returns it."""
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import nsparse_amd as ns
from gpu_util import synth, DeviceAMB
lib = ns.load("d")
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
A = synth(lib, 3, 14, 16, 0, seed=1)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR()
marks = []
for i in range(120):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
    if i in (5, 60, 119): marks.append(round(free_mb()))
print("free MB after 5/60/120 SpGEMM calls:", marks)
A2 = synth(lib, 1, 60, 60, 60, seed=2)
marks = []
for i in range(30):
    d = DeviceAMB(lib, A2); d.close()
    if i in (2, 15, 29): marks.append(round(free_mb()))
print("free MB after 2/15/30 AMB conversions:", marks)
lib.nsparse_trim_workspace(); print("after trim:", round(free_mb()))
