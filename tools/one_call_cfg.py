"""A few SpGEMM calls on one run_configs case (for rocprofv3 / counter runs).
NSPARSE_SERIAL=1: bins back to back on one stream, per-bin times printed."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth
from tools.run_configs import CASES
prec, kind, p = CASES[sys.argv[1]]
lib = ns.load(prec); lib.nsparse_set_bin_timing(1); A = synth(lib, kind, *p, seed=0x5EED0022)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR()
serial = os.environ.get("NSPARSE_SERIAL") == "1"
if serial:
    lib.nsparse_set_profiling(1)
if os.environ.get("NSPARSE_UNSORTED") == "1":
    lib.nsparse_spgemm_set_sorted(0)
for i in range(int(os.environ.get("NSPARSE_CALLS", "4"))):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.release_csr(c)
st = ns.SpgemmStats(); lib.nsparse_get_spgemm_stats(C.byref(st))
print("total %.3f ms  setup %.3f sym %.3f num %.3f" % (st.ms_total, st.ms_setup, st.ms_symbolic, st.ms_numeric))
print("sym rows", list(st.sym_bin_size)[:11]); print("sym ms  ", [round(v, 2) for v in list(st.ms_sym_bin)[:11]])
print("num rows", list(st.num_bin_size)[:11]); print("num ms  ", [round(v, 2) for v in list(st.ms_num_bin)[:11]])
