"""One configs-runner case with the bins SERIALISED (nsparse_set_profiling): per-bin kernel times that
do not include waiting for CUs held by other bins.  python tools/one_call_cfg.py <case> [reps]"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import nsparse_amd as ns
from gpu_util import synth
from tools.run_configs import CASES
prec, kind, p = CASES[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = ns.load(prec); lib.nsparse_set_profiling(int(os.environ.get("NSPARSE_SERIAL", "1"))); lib.nsparse_set_bin_timing(1)
if os.environ.get("NSPARSE_UNSORTED"): lib.nsparse_spgemm_set_sorted(0)
A = synth(lib, kind, *p, seed=0x5EED0022)
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"]); b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR(); st = ns.SpgemmStats()
for i in range(reps):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c)); lib.nsparse_get_spgemm_stats(C.byref(st)); lib.release_csr(c)
print(json.dumps(dict(case=sys.argv[1], serial=True, ms_total=round(st.ms_total, 3),
                      phase=[round(v, 3) for v in (st.ms_setup, st.ms_symbolic, st.ms_numeric)],
                      sym_bins=list(st.sym_bin_size)[:11], num_bins=list(st.num_bin_size)[:11],
                      sym_ms=[round(v, 3) for v in list(st.ms_sym_bin)[:11]],
                      num_ms=[round(v, 3) for v in list(st.ms_num_bin)[:11]])))
