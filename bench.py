#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X (BASELINE.json / BASELINE.md).

metric   "SpGEMM GFLOPS (C=A^2) and SpMV achieved HBM GB/s, fp64, per GPU"
value    SpGEMM GFLOPS = 2 * n_prod / t, the reference's definition (spgemm_hash.cu:35-54):
         t = mean over the K timed calls of the WHOLE spgemm_kernel_hash (binning, symbolic,
         scan, numeric, every allocation), inputs resident in HBM.
         The SpMV half of the metric is in "spmv" (same JSON line): achieved GB/s of sf_spmv_amb
         = reference footprint model bytes / t (spmv_amb.cu:46-62 protocol, 100 runs after 1).
step     one spgemm_kernel_hash call on this rank's batch (see workloads below).

Workloads (SuiteSparse files cannot be fetched: no network; $NSPARSE_DATA/<name>.mtx is used
when present, otherwise the deterministic synthetic stand-in of the same class):
  N = 1  configs[1]: cant class -- 3-dof 27-point FEM brick 9x9x257 = 62,451 rows (cant: 62,451),
         4.33 M nnz (cant: 4.0 M), fp64, C = A^2 and y = A x.
  N > 1  weak scaling of the same path by 1-D row partition: the brick is N times
         longer (9x9x257N), rank r owns row block r (62,451 rows) and computes
         C[rows_r,:] = A[rows_r,:] * A with B = A replicated -- no data-path collective.
         SpMV: y[rows_r] = A[rows_r,:] x, then ONE RCCL all-gather of y (the real exchange).
  always (secondary, "spmv_hbm"): nlpkkt120 class 27-point grid 160x164x135 = 3,542,400 rows,
         ~94 M nnz (1.2 GB per SpMV: out of the 256 MiB Infinity Cache, so GB/s means HBM),
         row-partitioned over the N ranks (strong scaling, configs[3]).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 through
python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synth(lib, kind, p0, p1, p2, seed, rows=(0, 0)):
    import nsparse_amd as ns
    m = ns.sfCSR()
    lib.nsparse_synth_csr(C.byref(m), kind, p0, p1, p2, seed, rows[0], rows[1])
    A = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    return A


def load_or_synth(lib, name, kind, dims, seed, rows=(0, 0)):
    data = os.environ.get("NSPARSE_DATA")
    if data and rows == (0, 0):
        path = os.path.join(data, name + ".mtx")
        if os.path.exists(path):
            import nsparse_amd as ns
            m = ns.sfCSR()
            lib.init_csr_matrix_from_file(C.byref(m), path.encode())
            A = lib.csr_host_to_numpy(m)
            lib.release_cpu_csr(m)
            return A, f"{name}.mtx"
    return synth(lib, kind, dims[0], dims[1], dims[2], seed, rows), f"synthetic {name}-class"


def numeric_bin_bytes(A, B, crpt, sym_ladder, ladder, w):
    """Algorithmic bytes of each numeric-bin launch (SURVEY 8d numeric term, restricted to the
    rows of the bin): per row 12 B (C.rpt pair + permutation entry) + (12+w) per A entry
    (col, val, two B.rpt gathers) + (4+w) per intermediate product (B col, val) + (4+w) per
    C entry written.  Rows are assigned to bins with the library's own rule (bins_of)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import numeric_bins, row_windows
    row_prod, span = row_windows(A, B)
    alen = np.diff(A["rpt"]).astype(np.int64)
    nzc = np.diff(crpt).astype(np.int64)
    bins = numeric_bins(nzc, row_prod, span, sym_ladder, ladder)
    per_row = 12 + (12 + w) * alen + (4 + w) * (row_prod + nzc)
    out = np.zeros(12)
    prods = np.zeros(12)
    for b in range(12):
        sel = bins == b
        out[b] = per_row[sel].sum()
        prods[b] = row_prod[sel].sum()
    return out, prods, row_prod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)   # SPGEMM_TRI_NUM - 1
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spmv-steps", type=int, default=100)  # TRI_NUM - 1
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-large", action="store_true", help="skip the nlpkkt-class SpMV")
    ap.add_argument("--no-vendor", action="store_true", help="skip the rocSPARSE (torch.sparse) baseline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import torch
    import torch.distributed as dist
    # NSPARSE_BENCH_BACKEND=gloo: smoke-test the multi-rank path on a box with fewer GPUs than
    # ranks (ranks share devices; RCCL refuses that).  Never used for reported numbers.
    backend = os.environ.get("NSPARSE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import nsparse_amd as ns
    from nsparse_amd.dist import make_gpu_sharded_spmv, row_partition
    lib = ns.load("d")
    lib.nsparse_set_bin_timing(1)  # the roofline leg needs the kernel time of the dominant bin
    w = 8

    # ------------------------------------------------------------------ workload ----
    nz = 257 * world
    M_glob = 9 * 9 * nz * 3
    rows = (rank * 62451, (rank + 1) * 62451)
    t0 = time.time()
    if world == 1:
        A_full, src = load_or_synth(lib, "cant", 0, (9, 9, nz), 0x5EED0022)
        A_loc = A_full
    else:
        A_full, src = load_or_synth(lib, "cant", 0, (9, 9, nz), 0x5EED0022)
        from nsparse_amd.dist import csr_row_block
        A_loc = csr_row_block(A_full, rows[0], rows[1])
    log(f"[rank {rank}] workload {src}: local {A_loc['M']} x {A_full['N']}, nnz local {A_loc['nnz'] if 'nnz' in A_loc else A_loc['rpt'][-1]}, "
        f"B nnz {A_full['rpt'][-1]} ({time.time() - t0:.1f}s)")

    a = lib.csr_from_numpy(A_loc["rpt"], A_loc["col"], A_loc["val"], A_full["N"])
    b = lib.csr_from_numpy(A_full["rpt"], A_full["col"], A_full["val"], A_full["N"])
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    flop = C.c_longlong()
    lib.get_spgemm_flop(C.byref(a), C.byref(b), a.M, C.byref(flop))

    # ------------------------------------------------------------- SpGEMM: timed ----
    c = ns.sfCSR()
    st = ns.SpgemmStats()
    for _ in range(args.warmup):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
        lib.release_csr(c)
    bin_ms = np.zeros(12)
    sym_ms = np.zeros(12)
    phase = np.zeros(4)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))  # synchronous on return
        lib.nsparse_get_spgemm_stats(C.byref(st))
        bin_ms += np.array(list(st.ms_num_bin))
        sym_ms += np.array(list(st.ms_sym_bin))
        phase += np.array([st.ms_setup, st.ms_symbolic, st.ms_numeric, st.ms_total])
        lib.release_csr(c)
    barrier()
    elapsed = time.perf_counter() - t_start
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    flops_all = torch.tensor([float(flop.value)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(flops_all, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / args.steps
    gflops = float(flops_all.item()) / (ms_per_step * 1e6)
    bin_ms /= args.steps
    sym_ms /= args.steps
    phase /= args.steps

    # one more call to keep C for the roofline byte counts
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
    crpt = lib.d2h(c.d_rpt, (c.M + 1,), np.int32)
    nnz_c = c.nnz
    lib.release_csr(c)
    sym_thr = (C.c_int * 15)()
    num_thr = (C.c_int * 15)()
    lib.nsparse_get_spgemm_bins(sym_thr, num_thr)
    bytes_bin, prods_bin, row_prod = numeric_bin_bytes(A_loc, A_full, crpt, list(sym_thr), list(num_thr), w)
    dom = int(np.argmax(bin_ms))
    achieved = bytes_bin[dom] / (bin_ms[dom] * 1e-3) / 1e9 if bin_ms[dom] > 0 else 0.0
    dom_kernel = {1: "k_num_tb<64,256,256>", 2: "k_num_tb<256,1024,1024>", 3: "k_num_tb<512,4096,4096>",
                  4: "k_num_tb<1024,8192,8192>", 5: "k_num_global<512>", 6: "k_num_dense<256,1536>",
                  7: "k_num_dense<256,4096>", 8: "k_num_dense<512,12288>"}.get(dom, "k_num_small")
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            key = "spgemm_" + dom_kernel.rstrip(">")
            hits = [v for k, v in pm.items() if k.startswith(key)]
            traffic = hits[0] if hits else None
        except Exception:
            traffic = None
    n_prod = int(flop.value // 2)
    nnz_a = int(A_loc["rpt"][-1])
    nnz_b = int(A_full["rpt"][-1])
    b_spgemm = (8 + w) * n_prod + (36 + w) * nnz_a + (4 + w) * nnz_c + 40 * a.M
    # compulsory traffic of the dominant launch: its A rows once, all of B once, its C rows once
    rows_dom = int(st.num_bin_size[dom])
    frac_rows = rows_dom / max(a.M, 1)
    b_comp = (4 + w) * nnz_a * frac_rows + (4 + w) * nnz_b + 4 * (A_full["M"] + 1) + (4 + w) * nnz_c * frac_rows
    roofline = {
        "bound": "hbm", "kernel": f"{dom_kernel} (numeric bin {dom}, {rows_dom} rows)",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "bytes_per_launch": int(bytes_bin[dom]), "ms_per_launch": round(float(bin_ms[dom]), 4),
        "products_per_launch": int(prods_bin[dom]),
        "note": "achieved = REQUESTED bytes (SURVEY 8d: every product re-reads its B entry) / kernel time; "
                "B rows are re-served by L2, so this can exceed the HBM peak. traffic = measured "
                "FETCH_SIZE*2048 + WRITE_SIZE*1024 per launch (profiles/). compulsory = each array once.",
        "compulsory": {"bytes": int(b_comp),
                       "achieved": round(b_comp / (bin_ms[dom] * 1e-3) / 1e9, 1) if bin_ms[dom] > 0 else 0.0,
                       "frac": round(b_comp / (bin_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if bin_ms[dom] > 0 else 0.0},
        "measured_hbm": ({"achieved": round(traffic / (bin_ms[dom] * 1e-3) / 1e9, 1),
                          "frac": round(traffic / (bin_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                         if traffic and bin_ms[dom] > 0 else None),
        # what actually limits a window-bin kernel: one LDS fp64 atomic per product, at the rate the SQ
        # counters show on gfx950 (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS: ~1.9 lanes per clock per CU)
        "lds_atomic_ceiling": ({"lanes_per_clk_per_cu": 1.94, "cus": 256, "clock_ghz": 2.4,
                                "floor_ms": round(prods_bin[dom] / (1.94 * 256 * 2.4e9) * 1e3, 4),
                                "frac": round(prods_bin[dom] / (1.94 * 256 * 2.4e9) * 1e3 / bin_ms[dom], 4)}
                               if dom >= 6 and bin_ms[dom] > 0 else None),
        "whole_call": {"bytes_model": int(b_spgemm),
                       "achieved": round(b_spgemm / (ms_per_step * 1e-3) / 1e9, 1),
                       "frac": round(b_spgemm / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }

    # ------------------------------------------------------------------- SpMV ----
    def time_spmv(op, x, steps, gather):
        for _ in range(2):
            op(x, gather=gather)
        barrier()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        for _ in range(steps):
            op(x, gather=gather)
        e1.record()
        barrier()
        el = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()) * 1e3 / steps, e0.elapsed_time(e1) / steps

    def spmv_report(A_rows, M_global, nnz_global, label, N_cols):
        op = make_gpu_sharded_spmv(lib, A_rows, M_global, rank, world, dev)
        x = torch.rand(N_cols + 20, dtype=torch.float64, device=dev)
        fp = int(lib.nsparse_amb_footprint_bytes(C.byref(op.amb)))
        # x is counted once over N instead of the reference's second M*w term
        b_amb = fp - A_rows["M"] * w + N_cols * w
        fp_all = torch.tensor([float(b_amb)], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(fp_all, op=dist.ReduceOp.SUM)
        ms_c, ms_c_ev = time_spmv(op, x, args.spmv_steps, gather=False)
        ms_g = time_spmv(op, x, args.spmv_steps, gather=True)[0] if world > 1 else ms_c
        b_csr = nnz_global * (w + 4) + 4 * (M_global + 1) + N_cols * w + M_global * w
        rep = {
            "workload": label, "M": M_global, "nnz": int(nnz_global),
            "plan": {"seg_size": int(op.plan.seg_size), "block_size": int(op.plan.block_size),
                     "thread_block": int(op.plan.thread_block), "chunk": int(op.amb.chunk)},
            "ms_per_spmv": round(ms_g, 5), "ms_compute_only": round(ms_c, 5),
            "ms_kernel_events": round(ms_c_ev, 5),
            "value": round(float(fp_all.item()) / (ms_g * 1e-3) / 1e9, 1), "unit": "GB/s",
            "gbs_compute_only": round(float(fp_all.item()) / (ms_c * 1e-3) / 1e9, 1),
            "frac_hbm_peak": round(float(fp_all.item()) / (ms_g * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "gbs_csr_model": round(b_csr / (ms_g * 1e-3) / 1e9, 1),
            "gflops_ref": round(2.0 * nnz_global / (ms_g * 1e6), 2),
            "bytes_amb_model": int(fp_all.item()),
        }
        # parity spot check against the library's own CPU path (csr_kernel) on rank rows
        y = op(x, gather=False)[:A_rows["M"]].cpu().numpy()
        m = lib.csr_from_numpy(A_rows["rpt"], A_rows["col"], A_rows["val"], N_cols)
        xh = x[:N_cols].cpu().numpy()
        yr = np.zeros(A_rows["M"])
        lib.csr_kernel(yr.ctypes.data_as(C.c_void_p), C.byref(m), xh.ctypes.data_as(C.c_void_p))
        rep["ans_check_fails"] = int(lib.nsparse_ans_check_count(
            yr.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), A_rows["M"]))
        lib.release_amb(op.amb)
        lib.release_csr(op.csr)
        return rep

    nnz_glob = int(A_full["rpt"][-1])
    from nsparse_amd.dist import csr_row_block as _blk
    _, blocks1 = row_partition(A_full["M"], world)
    A_spmv = A_full if world == 1 else _blk(A_full, *blocks1[rank])
    spmv = spmv_report(A_spmv, A_full["M"], nnz_glob, f"{src} (same matrix as SpGEMM)", A_full["N"])
    spmv_hbm = None
    if not args.no_large:
        gx, gy, gz = 160, 164, 135
        M2 = gx * gy * gz
        rpr, blocks = row_partition(M2, world)
        t0 = time.time()
        A2, src2 = load_or_synth(lib, "nlpkkt120", 1, (gx, gy, gz), 0x5EED0044, rows=blocks[rank] if world > 1 else (0, 0))
        nnz2 = torch.tensor([float(A2["rpt"][-1])], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(nnz2, op=dist.ReduceOp.SUM)
        log(f"[rank {rank}] {src2}: rows {A2['M']} nnz {A2['rpt'][-1]} ({time.time() - t0:.1f}s)")
        spmv_hbm = spmv_report(A2, M2, int(nnz2.item()), src2, M2)
        spmv_hbm["scaling"] = "strong"
        A2_host = A2 if world == 1 else None
    else:
        A2_host = None

    # ------------------------------------------------ vendor baseline (rocSPARSE) ----
    # The reference samples print their numbers next to cuSPARSE (spgemm_cu_csr / spmv_cu_csr,
    # SURVEY 8f rank 3).  rocSPARSE is reached through torch.sparse: CSR @ CSR is
    # rocsparse_spgemm, CSR @ vector is rocsparse_spmv.  Informational only.
    vendor = None
    if rank == 0 and world == 1 and not args.no_vendor:
        try:
            import warnings
            warnings.filterwarnings("ignore")
            crow = torch.from_numpy(A_loc["rpt"].astype(np.int32)).to(dev)
            ccol = torch.from_numpy(A_loc["col"].astype(np.int32)).to(dev)
            cval = torch.from_numpy(A_loc["val"].astype(np.float64)).to(dev)
            At = torch.sparse_csr_tensor(crow, ccol, cval, size=(A_loc["M"], A_full["N"]))
            for _ in range(2):
                Ct = torch.sparse.mm(At, At)
            torch.cuda.synchronize()
            t = time.perf_counter()
            reps_v = 5
            for _ in range(reps_v):
                Ct = torch.sparse.mm(At, At)
            torch.cuda.synchronize()
            ms_v = (time.perf_counter() - t) * 1e3 / reps_v
            xv = torch.rand(A_full["N"], dtype=torch.float64, device=dev)
            for _ in range(3):
                yv = At @ xv
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(50):
                yv = At @ xv
            torch.cuda.synchronize()
            ms_s = (time.perf_counter() - t) * 1e3 / 50
            vendor = {"library": "rocSPARSE via torch.sparse (torch %s)" % torch.__version__,
                      "spgemm_ms": round(ms_v, 3), "spgemm_gflops": round(flop.value / (ms_v * 1e6), 1),
                      "spgemm_nnz_c": int(Ct._nnz()), "spmv_ms": round(ms_s, 4),
                      "spmv_gbs_csr_model": round((nnz_a * 12 + 4 * (a.M + 1) + 16 * a.M) / (ms_s * 1e-3) / 1e9, 1)}
            del At, Ct
        except Exception as e:  # torch build without sparse CSR matmul
            vendor = {"error": repr(e)[:200]}

    # ----------------------------------------------------------- CPU baseline ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle.oracle import Oracle  # checker / baseline leg only
        orc = Oracle("d")
        cores = os.cpu_count() or 1
        t = time.perf_counter()
        ref = orc.spgemm(A_loc, A_full)
        t1 = time.perf_counter() - t
        t = time.perf_counter()
        orc.spgemm_omp(A_loc, A_full)
        tn = time.perf_counter() - t
        assert ref["nnz"] == nnz_c and np.array_equal(ref["rpt"], crpt), "GPU structure != oracle"
        xs = np.random.default_rng(1).random(A_full["N"])
        reps = 20
        orc.csr_spmv(A_loc["rpt"], A_loc["col"], A_loc["val"], xs)
        t = time.perf_counter()
        for _ in range(reps):
            orc.csr_spmv(A_loc["rpt"], A_loc["col"], A_loc["val"], xs)
        ts1 = (time.perf_counter() - t) / reps
        t = time.perf_counter()
        for _ in range(reps):
            orc.csr_spmv(A_loc["rpt"], A_loc["col"], A_loc["val"], xs, omp=True)
        tsn = (time.perf_counter() - t) / reps
        b_csr = nnz_a * (w + 4) + 4 * (a.M + 1) + A_full["N"] * w + a.M * w
        cpu = {
            "value": round(flop.value / t1 / 1e9, 3), "unit": "GFLOPS", "cores": 1, "kind": "port",
            "sample": f"whole {src} matrix, C=A^2 once, oracle/nsparse_oracle.c (the reference has no CPU SpGEMM)",
            "all_cores": {"value": round(flop.value / tn / 1e9, 3), "cores": cores},
            "spmv": {"value": round(b_csr / ts1 / 1e9, 2), "unit": "GB/s", "cores": 1,
                     "kind": "port", "sample": f"{reps} x csr_kernel loop order (nsparse.cu:240-259) on {src}",
                     "all_cores": {"value": round(b_csr / tsn / 1e9, 2), "cores": cores}},
        }

    lib.release_csr(a)
    lib.release_csr(b)
    if rank == 0:
        out = {
            "metric": "SpGEMM GFLOPS (C=A^2) and SpMV achieved HBM GB/s, fp64, per GPU",
            "value": round(gflops, 2), "unit": "GFLOPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if "synthetic" in src else "file",
            **({"backend_override": backend} if backend != "nccl" else {}),
            "config": {"workload": f"{src}: 3-dof 27-pt FEM brick 9x9x{nz}, {M_glob} rows, C=A^2 by 1-D row blocks of 62451 rows"
                                   if "synthetic" in src else src,
                       "rows_per_gpu": int(a.M), "nnz_A_per_gpu": nnz_a, "n_prod_per_gpu": n_prod,
                       "nnz_C_per_gpu": int(nnz_c), "parallelism": f"row-partition x{world}, B replicated",
                       "timing": "whole spgemm_kernel_hash call incl. allocation (block cache on)"},
            "phase_ms": {"setup": round(float(phase[0]), 4), "symbolic": round(float(phase[1]), 4),
                         "numeric": round(float(phase[2]), 4), "total_events": round(float(phase[3]), 4),
                         "numeric_bins": [round(float(v), 4) for v in bin_ms[:11]],
                         "symbolic_bins": [round(float(v), 4) for v in sym_ms[:11]],
                         "sym_bin_rows": list(st.sym_bin_size)[:11], "num_bin_rows": list(st.num_bin_size)[:11]},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "vendor_baseline": vendor,
            "spmv": spmv,
            "spmv_hbm": spmv_hbm,
        }
        C.CDLL(None).fflush(None)  # the library's own stdio lines ("Read mtx file: ...") go out first
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
