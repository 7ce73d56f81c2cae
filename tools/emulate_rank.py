#!/usr/bin/env python3
"""One rank of the N-rank weak-scaling SpGEMM run of bench.py on ONE GPU: A = row block `rank` of
the N-times-longer cant-class brick, B = the whole matrix.  Shows what grows with N on a rank
(set-up over B) without needing N GPUs.   python tools/emulate_rank.py 8 [rank]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsparse_amd as ns  # noqa: E402
from gpu_util import synth  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 5  # 5: the bench's headline stand-in; 0: the regular brick
lib = ns.load("d")
lib.nsparse_set_bin_timing(1)  # the phase times of the statistics are recorded only on request
rows = 62451
nz = 257 * world
B = synth(lib, kind, 9, 9, nz, seed=0x5EED0022)
lo, hi = rank * rows, (rank + 1) * rows
b0, b1 = int(B["rpt"][lo]), int(B["rpt"][hi])
A = dict(M=rows, N=B["N"], rpt=(B["rpt"][lo:hi + 1] - b0).astype(np.int32), col=B["col"][b0:b1], val=B["val"][b0:b1])
a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
b = lib.csr_from_numpy(B["rpt"], B["col"], B["val"], B["N"])
lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b))
c = ns.sfCSR(); st = ns.SpgemmStats()
tot, setup = [], []
for i in range(12):
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
    lib.nsparse_get_spgemm_stats(C.byref(st)); lib.release_csr(c)
    if i >= 2:
        tot.append(st.ms_total); setup.append(st.ms_setup)
print("world %d rank %d: B rows %d  | total %.4f ms  setup %.4f ms  (%.1f GFLOPS per rank)" %
      (world, rank, B["M"], np.mean(tot), np.mean(setup), 2 * st.n_prod / np.mean(tot) / 1e6))
