"""AMB SpMV in both precisions on a 120^3 27-point grid: ms and footprint-model GB/s (GPU box)."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth, DeviceAMB
for prec in ("d", "s"):
    lib = ns.load(prec)
    A = synth(lib, 1, 120, 120, 120, seed=5)
    d = DeviceAMB(lib, A)
    lib.nsparse_set_profiling(1)
    ms = []
    for i in range(30):
        lib.sf_spmv_amb(d.d_y, C.byref(d.amb), d.d_x, C.byref(d.plan)); ms.append(lib.nsparse_last_spmv_ms())
    w = lib.real().itemsize
    byt = lib.nsparse_amb_footprint_bytes(C.byref(d.amb))
    print(prec, "rows", A["M"], "nnz", int(A["rpt"][-1]), "ms %.4f" % np.mean(ms[5:]), "GB/s %.0f" % (byt / np.mean(ms[5:]) / 1e6), "bs", d.amb.block_size, "segs", d.amb.seg_num)
    lib.nsparse_set_profiling(0)
    d.close()
