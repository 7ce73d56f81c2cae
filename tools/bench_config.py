#!/usr/bin/env python3
"""One BASELINE config through the reference's SpGEMM protocol (spgemm_hash.cu:35-54: one warm-up, then the mean of
the timed whole calls), in its own torch-free process: bench.py's `configs` block runs this once per config.

    python tools/bench_config.py <case> [--steps K]      -> JSON lines on stdout, the LAST one is the record
    python tools/bench_config.py <case> --pmc-child      -> two calls and out (wrapped in rocprofv3 --pmc by bench.py)

Cases (stand-ins of tools/run_configs.py; $NSPARSE_DATA/<file>.mtx is used when present):
    webbase1m   config 3: webbase-1M class, fp32 (libnsparse_s.so), C = A^2
    rmat22      config 5: R-MAT scale 22, 7,340,032 edges (the largest edge count whose nnz(C) fits int), fp64
    stencil     27-point stencil 100^3, fp64 (every row in the one-wavefront hash bins)
The record: ms, GFLOPS, nnz(C), phase and per-bin times (separate pass with per-bin events), the whole-call compulsory
HBM fraction ((4+w)(nnz A + nnz B + nnz C) + 12 M bytes, every array once, over 8 TB/s), the dominant kernel, and
`structure_check`: nnz(C) and C.rpt against rocSPARSE csrgemm (libnsparse_vendor_*.so) -- the first JSON line is
printed BEFORE that check, so a vendor library that fails on a huge product costs the check, not the numbers.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsparse_amd as ns  # noqa: E402

HBM_PEAK_GBS = 8000.0
CASES = {  # name -> (precision, synth kind, params, file under $NSPARSE_DATA, BASELINE config)
    "webbase1m": ("s", 4, (1000005, 3105536, 0), "webbase-1M", 3),
    "rmat22": ("d", 3, (22, 0, 7340032), None, 5),
    "stencil": ("d", 1, (100, 100, 100), None, None),
    "rmat18": ("d", 3, (18, 16, 0), None, None),
}
# bin -> kernel family the library launches for it (spgemm_hash.hip: symbolic_phase / numeric_phase)
SYM_KERNEL = {0: "k_sym_small", 1: "k_sym_tb / k_sym_lean <64, 1024>", 2: "k_sym_tb / k_sym_lean <128, 2048>",
              3: "k_sym_tb / k_sym_lean <512, 8192>", 4: "k_sym_tb / k_sym_lean <1024, 32768>", 5: "k_sym_tb<1024, LARGE> + k_sym_global",
              6: "k_sym_dense<.., 4096>", 7: "k_sym_dense<.., 16384>", 8: "k_sym_dense<.., 65536>", 9: "k_sym_bits<512, 8192>",
              10: "k_sym_bits<1024, 32768> / k_num_ranked<.., SYM>"}
NUM_KERNEL = {0: "k_num_small", 1: "k_num_tb / k_num_lean <64, 256>", 2: "k_num_tb / k_num_lean <256, 1024>",
              3: "k_num_tb / k_num_lean <512, 4096>", 4: "k_num_tb / k_num_lean <1024, 8192>", 5: "k_num_tiled + k_num_ranked (heavy rows)",
              6: "k_num_block / k_num_dense <.., 1536>", 7: "k_num_block / k_num_dense <.., 4096>",
              8: "k_num_block / k_num_dense <.., 12288>", 9: "k_num_block<128, 65536> (ranked window)"}


def matrix(lib, name):
    prec, kind, p, fname, _ = CASES[name]
    data = os.environ.get("NSPARSE_DATA")
    m = ns.sfCSR()
    if data and fname and os.path.exists(os.path.join(data, fname + ".mtx")):
        lib.init_csr_matrix_from_file(C.byref(m), os.path.join(data, fname + ".mtx").encode())
        src = fname + ".mtx"
    else:
        lib.nsparse_synth_csr(C.byref(m), kind, p[0], p[1], p[2], 0x5EED0022, 0, 0)
        src = f"synthetic {name}-class (nsparse_synth_csr kind {kind} {p})"
    A = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    return A, src


def main():
    name = sys.argv[1]
    pmc_child = "--pmc-child" in sys.argv
    if os.environ.get("NSPARSE_BENCH_DRYRUN") == "1":
        # bench.py's dry run (tools/bench_dry.py): the sub-process plumbing with a canned record, no device
        from tools.bench_dry import CANNED_CONFIG
        if not pmc_child:
            rec = dict(CANNED_CONFIG, case=name, baseline_config=CASES[name][4], workload=f"dry run {name}",
                       dtype="f64" if CASES[name][0] == "d" else "f32")
            print(json.dumps(dict(rec, structure_check=None)), flush=True)
            print(json.dumps(rec), flush=True)
        return
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
    prec = CASES[name][0]
    w = 8 if prec == "d" else 4
    lib = ns.load(prec)
    lib.nsparse_set_bin_timing(0)
    t0 = time.time()
    A, src = matrix(lib, name)
    gen_s = time.time() - t0
    a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    c = ns.sfCSR()
    st = ns.SpgemmStats()
    if pmc_child:
        for _ in range(2):
            lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
            lib.release_csr(c)
        return
    flop = C.c_longlong()
    lib.get_spgemm_flop(C.byref(a), C.byref(b), a.M, C.byref(flop))
    t = time.perf_counter()
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))  # the warm-up of the protocol
    ms_first = (time.perf_counter() - t) * 1e3
    lib.release_csr(c)
    if steps <= 0:  # about a second of timed calls, 3 .. 10 of them
        steps = int(min(10, max(3, 1000.0 / max(ms_first, 1e-3))))
    lib.hip.hipDeviceSynchronize()
    t = time.perf_counter()
    for _ in range(steps):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))  # synchronous on return
        lib.release_csr(c)
    ms = (time.perf_counter() - t) * 1e3 / steps
    # separate pass: per-bin events on
    lib.nsparse_set_bin_timing(1)
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
    lib.nsparse_get_spgemm_stats(C.byref(st))
    lib.nsparse_set_bin_timing(0)
    nnz_a, nnz_c, M = int(A["rpt"][-1]), int(c.nnz), int(A["M"])
    sym_ms, num_ms = list(st.ms_sym_bin)[:11], list(st.ms_num_bin)[:11]
    ds, dn = int(np.argmax(sym_ms)), int(np.argmax(num_ms))
    dominant = ({"phase": "numeric", "bin": dn, "kernel": NUM_KERNEL.get(dn, "?"), "ms": round(num_ms[dn], 4)}
                if num_ms[dn] >= sym_ms[ds] else
                {"phase": "symbolic", "bin": ds, "kernel": SYM_KERNEL.get(ds, "?"), "ms": round(sym_ms[ds], 4)})
    comp = (4 + w) * (2 * nnz_a + nnz_c) + 12 * M
    rec = {
        "case": name, "baseline_config": CASES[name][4], "workload": src, "dtype": "f64" if prec == "d" else "f32",
        "library": os.path.basename(lib.path), "M": M, "nnz_A": nnz_a, "n_prod": int(flop.value // 2), "nnz_C": nnz_c,
        "steps": steps, "ms": round(ms, 4), "ms_first_call": round(ms_first, 3),
        "gflops": round(flop.value / (ms * 1e6), 2),
        "phase_ms": {"setup": round(st.ms_setup, 4), "symbolic": round(st.ms_symbolic, 4), "numeric": round(st.ms_numeric, 4)},
        "sym_bin_rows": list(st.sym_bin_size)[:11], "num_bin_rows": list(st.num_bin_size)[:11],
        "sym_bins_ms": [round(v, 4) for v in sym_ms], "num_bins_ms": [round(v, 4) for v in num_ms],
        "dominant": dominant,
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "compulsory_bytes": int(comp),
                     "achieved": round(comp / (ms * 1e-3) / 1e9, 1), "frac": round(comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "model": "(4+w)(nnz A + nnz B + nnz C) + 12 M: every array of the call once"},
        "gen_s": round(gen_s, 1), "structure_check": None,
    }
    print(json.dumps(rec), flush=True)
    # ---- nnz(C) and C.rpt against rocSPARSE (the role cuSPARSE plays in spgemm_hash.cu:60-68) ----
    try:
        crpt = lib.d2h(c.d_rpt, (c.M + 1,), np.int32)
        lib.release_csr(c)
        vl = ns.load_vendor(prec)
        cv = ns.sfCSR()
        msd = C.c_float()
        t = time.perf_counter()
        vl.nsparse_vendor_spgemm(C.byref(a), C.byref(b), C.byref(cv), C.byref(msd))
        v_ms = (time.perf_counter() - t) * 1e3
        err = int(vl.nsparse_vendor_last_error())
        if err:
            rec["structure_check"] = {"against": "rocSPARSE csrgemm", "error": err}
        else:
            v_rpt = lib.d2h(cv.d_rpt, (cv.M + 1,), np.int32)
            rec["structure_check"] = {"against": "rocSPARSE csrgemm (libnsparse_vendor_%s.so)" % prec,
                                      "nnz_equal": bool(cv.nnz == nnz_c), "rpt_equal": bool(np.array_equal(v_rpt, crpt)),
                                      "vendor_ms_first_call": round(v_ms, 2)}
            vl.nsparse_vendor_release_csr(cv)
    except Exception as e:  # the numbers above stand; the check is reported as failed
        rec["structure_check"] = {"against": "rocSPARSE csrgemm", "error": repr(e)[:160]}
    print(json.dumps(rec), flush=True)
    lib.release_csr(a)
    lib.release_csr(b)


if __name__ == "__main__":
    main()
