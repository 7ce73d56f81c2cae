"""One matrix through the bench protocol in its own process (environment switches are read once per process):
python tools/one_gflops.py kind p0 p1 p2 [steps]  ->  one JSON line {gflops, ms, twin_rows}."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth
kind, p0, p1, p2 = (int(v) for v in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
lib = ns.load("d")
A = synth(lib, kind, p0, p1, p2, seed=0x5EED0022)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR(); st = ns.SpgemmStats()
fl = C.c_longlong(); lib.get_spgemm_flop(C.byref(a), C.byref(b), a.M, C.byref(fl))
for _ in range(2):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
t = time.perf_counter()
for _ in range(steps):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.nsparse_get_spgemm_stats(C.byref(st)); lib.release_csr(c)
ms = (time.perf_counter() - t) * 1e3 / steps
print(json.dumps(dict(gflops=round(fl.value / (ms * 1e6), 2), ms=round(ms, 4), twin_rows=int(st.twin_rows))))
