#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X (BASELINE.json / BASELINE.md).

metric   "SpGEMM GFLOPS (C=A^2) and SpMV achieved HBM GB/s, fp64, per GPU"
value    SpGEMM GFLOPS = 2 * n_prod / t, the reference's definition (spgemm_hash.cu:35-54):
         t = mean over the K timed calls of the WHOLE spgemm_kernel_hash (binning, symbolic, scan,
         numeric, every allocation the call makes), inputs resident in HBM, per-bin event timing OFF.
         Allocations come from the library's block cache (its default); "timing" in the line also
         holds the same loop with nsparse_set_workspace_cache(0), i.e. hipMalloc / hipFree inside
         every call like the reference ("reference_compatible").
         The SpMV half of the metric is in "spmv" / "spmv_hbm" (same JSON line): achieved GB/s of
         sf_spmv_amb = reference footprint-model bytes / t (spmv_amb.cu:46-62: 100 runs after 1).
step     one spgemm_kernel_hash call on this rank's batch.

Workloads (SuiteSparse files cannot be fetched: no network; $NSPARSE_DATA/<name>.mtx is used when
present, otherwise the deterministic stand-in of the same class, nsparse_synth_csr):
  N = 1  configs[1]: cant class, fp64, C = A^2 and y = A x.  cant has 62,451 = 3 * 9 * 9 * 257 rows.
         `value` is measured on $NSPARSE_DATA/cant.mtx when that file exists, else on the stand-in that
         matches cant's SuiteSparse STATISTICS within 1 %: a 9 x 9 x 257 brick of 3-dof nodes renumbered
         inside bands of three mesh planes with 7.4 % of the node couplings dropped (kind 5: 4.02 M nnz,
         0.271 G products, 17.3 M nnz(C)).  Round 2 put the REGULAR brick (kind 0: natural numbering,
         16 % more products, every row in the narrowest window bin) there; it is now "regular_brick" in
         the line, beside the headline.  "structure_sweep" bounds how much of the speed hangs on the
         node structure: kind 6 = kind 5 with scalar perturbations (a dof constrained / one scalar
         coupling dropped) on 0 / 10 / 30 / 100 % of the nodes, and the NSPARSE_TWINS=0 floor.
  N > 1  weak scaling by 1-D row partition: the brick is N times longer (9x9x257N),
         rank r owns row block r (62,451 rows) and computes C[rows_r,:] = A[rows_r,:] * A with
         B = A replicated -- no data-path collective.  SpMV: y[rows_r] = A[rows_r,:] x, then ONE
         RCCL all-gather of y (the real exchange).
  always (secondary, "spmv_hbm"): nlpkkt120 class 27-point grid 160x164x135 = 3,542,400 rows,
         94.4 M nnz (1.0 GB per SpMV: out of the 256 MiB Infinity Cache, so GB/s means HBM),
         row-partitioned over the N ranks by nnz (strong scaling, configs[3]).

roofline (dominant kernel of the headline workload, from a SEPARATE pass with per-bin HIP events on
the bin's own stream):
  frac / achieved   COMPULSORY HBM bytes of that launch (its A rows, all of B once, its C rows) /
                    kernel time / 8 TB/s -- a lower bound of the traffic, never above 1
  traffic           HBM bytes per launch from rocprofv3 PMC passes made by THIS run on THIS input
                    (FETCH_SIZE * 2048 + WRITE_SIZE * 1024, the calibration of
                    profiles/r01_pmc_calibration.txt = the guide's KiB unit and gfx950 x2 read
                    correction); null when rocprofv3 is not usable
  l2_requested      SURVEY 8d's requested-bytes model (every product re-reads its B entry) against
                    the 34.5 TB/s aggregate L2 ceiling: what the cache hierarchy serves
  lds_atomic        products / (measured ds_add_f64 rate): tools/lds_atomic/, profiles/r02_lds_atomic.json

configs  (N = 1) BASELINE configs 3 and 5 and the 27-point stencil through the same protocol, each in its own
         torch-free process (tools/bench_config.py): webbase-1M class in fp32 (libnsparse_s.so), R-MAT scale 22 with
         7,340,032 edges, stencil 100^3 -- ms, GFLOPS, nnz(C) and C.rpt checked against rocSPARSE, whole-call
         compulsory HBM fraction, dominant kernel, PMC traffic of the call.

No torch anywhere in the measuring processes (round 4): the library binds to the system ROCm runtime
(/opt/rocm/lib/libamdhip64.so.7, librccl.so.1) -- with torch imported first it bound to torch's bundled HIP 7.0 /
RCCL 2.26, which is not the runtime the GPU tests validate and doubled the allocate-inside-the-call timing.
Ranks: one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT from the environment (the driver's
`python -m torch.distributed.run ... bench.py` sets them; only ITS agent process imports torch).  The ncclUniqueId
travels through nsparse_amd/rendezvous.py (a directory + a localhost socket), the barriers and max / sum
reductions around the timed loops are nsparse_dist_barrier / nsparse_dist_allreduce_f64 of the native library
(RCCL on the rank's stream).  Every wait has a deadline: more ranks than GPUs, or a rank that never starts, is an
error message and a non-zero exit code, not a hang.

Launch: python bench.py [--gpus N --steps K --warmup W]; N > 1 without WORLD_SIZE in the environment spawns its N
ranks itself as plain subprocesses.  NSPARSE_BENCH_EMULATE=1 (tests, never a measurement): the ranks share the
GPUs that exist, no communicator; barriers / reductions through the rendezvous socket, the all-gather replaced by
the library's own staging + gap-closing copy (nsparse_dist_close_gaps) -- the whole multi-rank control flow on a
one-GPU box.
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md: aggregate L2, 8 XCDs
FETCH_UNIT, WRITE_UNIT = 2048, 1024  # bytes per PMC unit, profiles/r01_pmc_calibration.txt

# numeric bin -> kernel the library launches for it (csrc/spgemm_hash.hip: numeric_phase)
# (the window bins 6-8 run k_num_block<128, span, ...> on matrices with twin rows, else k_num_dense<BS, span, ...>)
NUM_KERNEL = {0: ["k_num_small<256, 4, 32"], 1: ["k_num_tb<64, 256, 256"], 2: ["k_num_tb<256, 1024, 1024"],
              3: ["k_num_tb<512, 4096, 4096"], 4: ["k_num_tb<1024, 8192, 8192"],
              5: ["k_num_tiled<1024, 12288", "k_num_ranked"],
              6: ["k_num_block<128, 1536, 1", "k_num_dense<256, 1536, 1"],
              7: ["k_num_block<128, 4096, 1", "k_num_dense<256, 4096, 1"],
              8: ["k_num_block<128, 12288, 1", "k_num_dense<512, 12288, 1"],
              9: ["k_num_block<128, 65536, 1"]}

STANDINS = {  # name -> (kind, params, seed)
    "cant": (5, (9, 9, 257), 0x5EED0022),        # statistics of SuiteSparse cant within 1 % (the headline)
    "cant_brick": (0, (9, 9, 257), 0x5EED0022),  # regular brick, natural numbering (round 2's headline)
    "nlpkkt120": (1, (160, 164, 135), 0x5EED0044),
}


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


def numeric_bin_models(A, B, crpt, sym_ladder, ladder, w):
    """Per numeric bin: rows, nnz(A) of its rows, products, nnz(C) of its rows, and the two byte
    models -- requested (SURVEY 8d numeric term: 12 B per row + (12+w) per A entry + (4+w) per
    product + (4+w) per C entry) and compulsory (its A rows + its C rows; B is added once by the
    caller).  Rows are assigned to bins with the library's own rule (tests/gpu_util.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import numeric_bins, row_windows
    row_prod, span = row_windows(A, B)
    alen = np.diff(A["rpt"]).astype(np.int64)
    nzc = np.diff(crpt).astype(np.int64)
    bins = numeric_bins(nzc, row_prod, span, sym_ladder, ladder)
    out = {}
    for b in range(12):
        sel = bins == b
        if not sel.any():
            continue
        na, npr, nc, nr = int(alen[sel].sum()), int(row_prod[sel].sum()), int(nzc[sel].sum()), int(sel.sum())
        out[b] = dict(rows=nr, nnz_a=na, products=npr, nnz_c=nc,
                      requested=12 * nr + (12 + w) * na + (4 + w) * (npr + nc),
                      compulsory_rows=12 * nr + (4 + w) * na + (4 + w) * nc)
    return out


def spgemm_loop(lib, a, b, steps, warmup, barrier, with_stats=False):
    """The reference's timed loop (spgemm_hash.cu:35-54): W untimed calls, then K timed ones between
    two barriers.  Returns (seconds, per-bin numeric ms, per-bin symbolic ms, phase ms, last stats)."""
    import nsparse_amd as ns
    c = ns.sfCSR()
    st = ns.SpgemmStats()
    for _ in range(warmup):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
        lib.release_csr(c)
    num_ms, sym_ms, phase = np.zeros(12), np.zeros(12), np.zeros(4)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))  # synchronous on return
        if with_stats:
            lib.nsparse_get_spgemm_stats(C.byref(st))
            num_ms += np.array(list(st.ms_num_bin))
            sym_ms += np.array(list(st.ms_sym_bin))
            phase += np.array([st.ms_setup, st.ms_symbolic, st.ms_numeric, st.ms_total])
        lib.release_csr(c)
    barrier()
    el = time.perf_counter() - t0
    lib.nsparse_get_spgemm_stats(C.byref(st))
    return el, num_ms / max(steps, 1), sym_ms / max(steps, 1), phase / max(steps, 1), st


# ------------------------------------------------------------------------------------ PMC ----
def pmc_pass(counter, workload, timeout_s=240):
    """One rocprofv3 --pmc pass over tools/pmc_one.py (torch-free: library + generator only).
    Returns {kernel name: mean counter value per launch} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    td = tempfile.mkdtemp(prefix="nsp_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_one.py"), workload]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                           timeout=timeout_s)
        if r.returncode != 0:
            log(f"[pmc] {counter} pass failed rc={r.returncode}: {r.stderr[-300:]}")
            return None
        import csv
        agg = {}
        for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != counter:
                    continue
                agg.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        # the first launches of a kernel include cold caches: keep the later half
        return {k: float(np.mean(v[len(v) // 2:])) for k, v in agg.items()}
    except Exception as e:  # timeout, missing tool, parse error: traffic stays null
        log(f"[pmc] {counter} pass: {e!r}")
        return None
    finally:
        shutil.rmtree(td, ignore_errors=True)


def pmc_traffic(workload):
    """HBM bytes per launch for every kernel of the workload: separate FETCH_SIZE / WRITE_SIZE passes."""
    f = pmc_pass("FETCH_SIZE", workload)
    wr = pmc_pass("WRITE_SIZE", workload) if f is not None else None
    if f is None or wr is None:
        return None
    return {k: {"fetch_bytes": f[k] * FETCH_UNIT, "write_bytes": wr.get(k, 0.0) * WRITE_UNIT,
                "hbm_bytes": f[k] * FETCH_UNIT + wr.get(k, 0.0) * WRITE_UNIT} for k in f}


def find_kernel(traffic, patterns):
    """First kernel of the PMC pass whose name contains one of the patterns -> (name, counters)."""
    if not traffic:
        return None, None
    for pattern in ([patterns] if isinstance(patterns, str) else patterns):
        pat = pattern.replace(" ", "")
        for k, v in traffic.items():
            if pat in k.replace(" ", ""):
                short = k.split("(")[0].replace("void ", "").replace("nsp::spgemm::", "").replace("nsp::spmv::", "")
                return short, v
    return None, None


def runtime_report():
    """Which HIP / RCCL / rocSPARSE shared objects this process has mapped (from /proc/self/maps): the evidence that
    the library ran on the system ROCm runtime and not on a copy some Python package brought along."""
    libs = {}
    try:
        for ln in open("/proc/self/maps"):
            path = ln.split()[-1] if "/" in ln else ""
            base = os.path.basename(path)
            for key in ("libamdhip64", "librccl", "librocsparse", "libnsparse_", "libtorch", "libc10"):
                if base.startswith(key):
                    libs[base] = path
    except OSError:
        pass
    rocm = [v for k, v in libs.items() if k.startswith(("libamdhip64", "librccl", "librocsparse"))]
    # no HIP runtime mapped at all (dry run, CPU emulation of the device) is NOT "on the system runtime"
    return {"mapped": libs, "torch_imported": "torch" in sys.modules,
            "system_rocm_runtime": any(os.path.basename(v).startswith("libamdhip64") for v in rocm)
                                   and all(v.startswith("/opt/rocm") for v in rocm)}


def run_config(case, pmc, deadline):
    """One BASELINE config in its own process (tools/bench_config.py); with `pmc`, two more runs of it under
    rocprofv3 (FETCH_SIZE, WRITE_SIZE: separate passes) for the HBM traffic of one whole call."""
    script = os.path.join(ROOT, "tools", "bench_config.py")
    left = deadline - time.time()
    if left < 20:
        return {"case": case, "skipped": "configs time budget spent"}
    try:
        r = subprocess.run([sys.executable, script, case], capture_output=True, text=True, cwd=ROOT,
                           timeout=min(left, 100.0))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            return {"case": case, "error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        rec = json.loads(lines[-1])
    except subprocess.TimeoutExpired as e:
        so = e.stdout or ""
        so = so.decode(errors="replace") if isinstance(so, bytes) else so
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        if not lines:
            return {"case": case, "error": "timed out before the first record"}
        rec = json.loads(lines[-1])
        rec["structure_check"] = {"against": "rocSPARSE csrgemm", "error": "timed out"}
    if not pmc:
        return rec
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = deadline - time.time()
        if left < 25 or not (os.path.exists(exe) or os.environ.get("NSPARSE_BENCH_DRYRUN") == "1"):
            rec["traffic"] = None
            rec["traffic_note"] = "PMC pass skipped: " + ("time budget" if left < 25 else "no rocprofv3")
            return rec
        td = tempfile.mkdtemp(prefix="nsp_pmc_", dir="/tmp")
        try:
            if os.environ.get("NSPARSE_BENCH_DRYRUN") == "1":
                # what rocprofv3 leaves behind, in its layout: two launches per kernel (two calls were made)
                os.makedirs(os.path.join(td, "host", "1"))
                with open(os.path.join(td, "host", "1", "p_counter_collection.csv"), "w") as f:
                    f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size",'
                            '"Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count",'
                            '"Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n')
                    for i, kn in enumerate(("void nsp::spgemm::k_num_tb<64, 256, 256, 0>(int const*, int)",
                                            "void nsp::spgemm::k_sym_tb<64, 1024, 0>(int const*, int)") * 2):
                        f.write(f'{i},{i},1,1,1,1,4096,{i % 2},"{kn}",64,0,0,32,0,16,"{counter}",{1000.0 + i},1,2\n')
            else:
                rr = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "p", "--",
                                     sys.executable, script, case, "--pmc-child"], cwd="/tmp",
                                    env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=min(left, 90.0))
                if rr.returncode != 0:
                    raise RuntimeError(f"rc {rr.returncode}: {rr.stderr[-200:]}")
            import csv
            per = {}
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            # two calls were made; the second half of every kernel's launches belongs to the second (warm) call
            tot[counter] = {k: float(np.sum(v[len(v) // 2:])) for k, v in per.items()}
        except Exception as e:
            rec["traffic"] = None
            rec["traffic_note"] = f"{counter} pass: {e!r}"[:200]
            return rec
        finally:
            shutil.rmtree(td, ignore_errors=True)
    kern = {k: tot["FETCH_SIZE"].get(k, 0.0) * FETCH_UNIT + tot["WRITE_SIZE"].get(k, 0.0) * WRITE_UNIT
            for k in set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])}
    call = float(sum(kern.values()))
    top = sorted(kern.items(), key=lambda kv: -kv[1])[:3]

    def short(k):
        return k.split("(")[0].replace("void ", "").replace("nsp::spgemm::", "")[:80]
    rec["traffic"] = int(call)
    rec["traffic_how"] = ("HBM bytes of ONE whole call: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), summed over "
                          "the kernels of the second of two calls, x 2048 / x 1024 B per unit (profiles/r01_pmc_calibration.txt)")
    rec["traffic_over_compulsory"] = round(call / max(rec["roofline"]["compulsory_bytes"], 1), 2)
    rec["traffic_top_kernels"] = [{"kernel": short(k), "hbm_bytes": int(v)} for k, v in top]
    rec["roofline"]["measured_achieved"] = round(call / (rec["ms"] * 1e-3) / 1e9, 1)
    rec["roofline"]["measured_frac"] = round(call / (rec["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return rec


def spawn_ranks(args):
    """--gpus N without WORLD_SIZE: N plain subprocesses of this script, one per GPU (no torch, no torchrun).  The
    first rank that fails takes the others down; the whole run has a deadline."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdv = tempfile.mkdtemp(prefix="nsparse_rdv_", dir="/tmp")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NSPARSE_RDV=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: spawned {args.gpus} rank processes (rendezvous {rdv})")
    deadline = time.time() + float(os.environ.get("NSPARSE_BENCH_DEADLINE_S", "1500"))
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0].returncode if bad else 124
            log(f"[bench] {'a rank exited with ' + str(rc) if bad else 'deadline passed'}: stopping the other ranks")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    shutil.rmtree(rdv, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)   # SPGEMM_TRI_NUM - 1
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spmv-steps", type=int, default=100)  # TRI_NUM - 1
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-large", action="store_true", help="skip the nlpkkt-class SpMV")
    ap.add_argument("--no-vendor", action="store_true", help="skip the rocSPARSE baseline")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-irregular", action="store_true", help="skip the regular brick and the structure sweep")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 3 / 5 and the stencil (the `configs` block)")
    ap.add_argument("--configs-budget", type=float, default=110.0, help="seconds the `configs` block may take")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries ONE line, the JSON record: whatever libraries print there on the way (RCCL's version banner at
    # communicator creation, the loader's "Read mtx file" lines) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    t_start = time.time()
    import nsparse_amd as ns
    from nsparse_amd.dist import csr_row_block, row_partition, row_partition_nnz
    from nsparse_amd.rendezvous import Rendezvous
    assert "torch" not in sys.modules, "the measuring process must not import torch (it brings its own HIP / RCCL)"
    # NSPARSE_BENCH_DRYRUN=1 (tests, never a measurement): the whole control flow of this script on a box without a
    # GPU -- tools/bench_dry.py stands in for everything that needs the device, on small matrices
    dry = os.environ.get("NSPARSE_BENCH_DRYRUN") == "1"
    if dry:
        from tools import bench_dry
        lib = bench_dry.DryLib("d")
        dl = bench_dry.DryDist(lib)
        args.no_vendor = True
        log("[bench] NSPARSE_BENCH_DRYRUN=1: no device, made-up times -- a test of the harness, not a measurement")
    else:
        lib = ns.load("d")
        dl = ns.load_dist("d")
    # a library built by tests/emu (the kernels compiled for the HOST against a lane-by-lane emulation of the device
    # model) says so in its build string: whatever it produces is a test of control flow and answers, never a number
    emulated_device = (not dry) and lib.nsparse_build_info().decode().split()[2:3] == ["emu"]
    if emulated_device:
        log("[bench] the loaded libnsparse is the CPU emulation build (tests/emu): this line is NOT a measurement")
    w = 8
    planes = 5 if dry else 257            # mesh planes of the cant-class brick per rank (cant: 62,451 = 3 * 9 * 9 * 257 rows)
    rows_rank = 3 * 9 * 9 * planes
    # NSPARSE_BENCH_EMULATE=1: the multi-rank control flow on fewer GPUs than ranks (tests; never a measurement)
    emulate = os.environ.get("NSPARSE_BENCH_EMULATE") == "1"
    ndev = int(dl.nsparse_dist_device_count())
    if ndev < 1:
        log(f"[rank {rank}] no GPU visible to this process")
        sys.exit(3)
    if world > ndev and not emulate:
        log(f"[rank {rank}] --gpus {world} but this box has {ndev} GPU(s): one rank per GPU is the only measured "
            "configuration (RCCL refuses two ranks on one device).  NSPARSE_BENCH_EMULATE=1 runs the control flow "
            "of the multi-rank path on the GPUs that exist, as a test, without a communicator.")
        sys.exit(3)
    if not dry:
        lib.hip.hipSetDevice.argtypes = [C.c_int]
    assert lib.hip.hipSetDevice(local_rank % ndev) == 0
    lib.nsparse_set_bin_timing(0)
    dl.nsparse_dist_set_timeout(float(os.environ.get("NSPARSE_DIST_TIMEOUT_S", "90")))

    # ---- ranks: rendezvous (host), then ONE communicator for the whole run -------------------------------------
    rdv = Rendezvous(rank, world, timeout=float(os.environ.get("NSPARSE_RDV_TIMEOUT_S", "120")))
    if dry:
        dl.attach(rdv)
    h = C.c_void_p()
    if world > 1 and not emulate:
        idb = C.create_string_buffer(ns.DIST_ID_BYTES)
        if rank == 0:
            assert dl.nsparse_dist_unique_id(idb) == 0, "ncclGetUniqueId failed"
        idb = C.create_string_buffer(rdv.bcast(idb.raw if rank == 0 else None, "ncclUniqueId"), ns.DIST_ID_BYTES)
        rc = dl.nsparse_dist_init(C.byref(h), idb, rank, world)
        if not rdv.all_ok(rc == 0, "communicator"):
            log(f"[rank {rank}] RCCL communicator not created on every rank (this rank: {rc}); no measurement")
            sys.exit(4)
    else:
        rc = dl.nsparse_dist_init(C.byref(h), None, rank, world)
        assert rc == 0, f"nsparse_dist_init -> {rc}"
    native_coll = world > 1 and not emulate

    def check(rc, what):
        if rc != 0:
            log(f"[rank {rank}] {what} -> {rc} (nsparse_dist_last_error {dl.nsparse_dist_last_error()})")
            sys.exit(5)

    def barrier():
        if native_coll:
            check(dl.nsparse_dist_barrier(h), "nsparse_dist_barrier")
        elif world > 1:
            lib.hip.hipDeviceSynchronize()
            rdv.barrier()
        lib.hip.hipDeviceSynchronize()

    def reduce_ranks(x, op):
        if world == 1:
            return float(x)
        if not native_coll:
            return rdv.allreduce([x], "sum" if op == 0 else "max")[0]
        v = (C.c_double * 1)(float(x))
        check(dl.nsparse_dist_allreduce_f64(h, v, 1, op), "nsparse_dist_allreduce_f64")
        return float(v[0])

    def max_over_ranks(x):
        return reduce_ranks(x, 1)

    def sum_over_ranks(x):
        return reduce_ranks(x, 0)

    # ------------------------------------------------------------------ workload ----
    nz = planes * world
    M_glob = 9 * 9 * nz * 3
    rows = (rank * rows_rank, (rank + 1) * rows_rank)
    t0 = time.time()
    kind, _, seed = STANDINS["cant"]
    A_full, src = load_or_synth(lib, "cant", kind, (9, 9, nz), seed) if world == 1 else \
        (synth(lib, kind, 9, 9, nz, seed), "synthetic cant-class")
    A_loc = A_full if world == 1 else csr_row_block(A_full, rows[0], rows[1])
    log(f"[rank {rank}] workload {src}: local {A_loc['M']} x {A_full['N']}, nnz local {int(A_loc['rpt'][-1])}, "
        f"B nnz {int(A_full['rpt'][-1])} ({time.time() - t0:.1f}s)")

    a = lib.csr_from_numpy(A_loc["rpt"], A_loc["col"], A_loc["val"], A_full["N"])
    b = lib.csr_from_numpy(A_full["rpt"], A_full["col"], A_full["val"], A_full["N"])
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    flop = C.c_longlong()
    lib.get_spgemm_flop(C.byref(a), C.byref(b), a.M, C.byref(flop))
    flops_all = sum_over_ranks(flop.value)

    # --------------------------------------------- SpGEMM: the timed K steps (headline) ----
    el, _, _, _, st = spgemm_loop(lib, a, b, args.steps, args.warmup, barrier)
    elapsed = max_over_ranks(el)
    ms_per_step = elapsed * 1e3 / args.steps
    gflops = flops_all / (ms_per_step * 1e6)

    # the same loop, allocating inside the call like the reference (block cache off)
    lib.nsparse_set_workspace_cache(0)
    el_ref, _, _, _, _ = spgemm_loop(lib, a, b, args.steps, 1, barrier)
    # ... and with the runtime's stream-ordered allocator (hipMallocAsync / hipFreeAsync inside the call)
    lib.nsparse_set_workspace_cache(2)
    el_async, _, _, _, _ = spgemm_loop(lib, a, b, args.steps, 2, barrier)
    lib.nsparse_set_workspace_cache(1)
    ms_ref = max_over_ranks(el_ref) * 1e3 / args.steps
    ms_async = max_over_ranks(el_async) * 1e3 / args.steps

    # ------------------------------------- roofline pass: per-bin events, separate loop ----
    lib.nsparse_set_bin_timing(1)
    _, bin_ms, sym_ms, phase, st = spgemm_loop(lib, a, b, args.steps, 1, barrier, with_stats=True)
    lib.nsparse_set_bin_timing(0)
    c = ns.sfCSR()
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))  # keep C for the byte models
    crpt = lib.d2h(c.d_rpt, (c.M + 1,), np.int32)
    nnz_c = c.nnz
    lib.release_csr(c)
    sym_thr = (C.c_int * 18)()
    num_thr = (C.c_int * 18)()
    lib.nsparse_get_spgemm_bins(sym_thr, num_thr)
    models = numeric_bin_models(A_loc, A_full, crpt, list(sym_thr), list(num_thr), w)
    # the library folds window bin 6 into bin 7's launch when both hold rows (spgemm_hash.hip: fold6): one kernel,
    # timed as bin 7 -- its byte models are the two bins' together
    if 6 in models and 7 in models and float(bin_ms[6]) == 0.0 and float(bin_ms[7]) > 0.0:
        models[7] = {k: models[6][k] + models[7][k] for k in models[7]}
        del models[6]
    dom = int(np.argmax(bin_ms))
    t_dom = float(bin_ms[dom]) * 1e-3
    mdl = models.get(dom, dict(rows=0, nnz_a=0, products=0, nnz_c=0, requested=0, compulsory_rows=0))
    n_prod = int(flop.value // 2)
    nnz_a = int(A_loc["rpt"][-1])
    nnz_b = int(A_full["rpt"][-1])
    # compulsory HBM bytes of the dominant launch: its A rows and C rows, plus the part of B the rank's
    # rows reach, once (a row block of a banded matrix reaches its own stretch of B, not all of it)
    reach = float(nnz_a) / max(nnz_b, 1) if world > 1 else 1.0
    b_once = ((4 + w) * nnz_b + 4 * (A_full["M"] + 1)) * min(1.0, reach * 1.1)
    b_comp = mdl["compulsory_rows"] + b_once
    b_spgemm = (8 + w) * n_prod + (36 + w) * nnz_a + (4 + w) * nnz_c + 40 * a.M  # SURVEY 8d, whole call

    traffic_all = None
    if rank == 0 and world == 1 and not args.no_pmc:
        t0 = time.time()
        traffic_all = bench_dry.pmc_traffic("bench") if dry else pmc_traffic("bench")
        log(f"[pmc] two passes in {time.time() - t0:.0f}s: {'ok' if traffic_all else 'unavailable'}")
    pats = NUM_KERNEL.get(dom, [f"numeric bin {dom}"])
    kname, tr = find_kernel(traffic_all, pats)
    # without a PMC pass the name is the first candidate: the node-block kernel when the matrix has twin rows
    dom_kernel = kname if kname else (pats[0] if st.twin_rows * 8 >= a.M or dom < 6 else pats[-1]) + ", ...>"

    lds = None
    lds_path = os.path.join(ROOT, "profiles", "r02_lds_atomic.json")
    if dom >= 6 and os.path.exists(lds_path) and t_dom > 0:
        try:
            rowsj = json.load(open(lds_path))["rows"]
            best = max(r["lanes_per_clk_per_cu"] for r in rowsj if r["type"] == "ds_add_f64" and r["pattern"] == "consecutive")
            fem = [r for r in rowsj if r["type"] == "ds_add_f64" and r["pattern"] == "random_1536" and r["active_lanes"] == 32]
            clk = 2.4e9
            lds = {"source": "profiles/r02_lds_atomic.json (tools/lds_atomic/lds_atomic_bench.hip on this GPU model)",
                   "ds_add_f64_lanes_per_clk_per_cu_peak": best,
                   "floor_ms_at_peak": round(mdl["products"] / (best * 256 * clk) * 1e3, 4),
                   "frac_of_peak": round(mdl["products"] / (best * 256 * clk) / t_dom, 4)}
            if fem:
                r24 = fem[0]["lanes_per_clk_per_cu"]
                lds.update({"lanes_per_clk_per_cu_32_lanes_random_window": r24,
                            "floor_ms_at_32_lanes_random": round(mdl["products"] / (r24 * 256 * clk) * 1e3, 4),
                            "needed_lanes_per_clk_per_cu": round(mdl["products"] / (t_dom * 256 * clk), 3)})
        except Exception as e:
            lds = {"error": repr(e)[:120]}

    def gbs(nbytes, t):
        return round(nbytes / t / 1e9, 1) if t > 0 else 0.0

    roofline = {
        "bound": "hbm", "kernel": f"{dom_kernel} (numeric bin {dom}, {mdl['rows']} rows, {mdl['products']} products)",
        "achieved": gbs(b_comp, t_dom), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(gbs(b_comp, t_dom) / HBM_PEAK_GBS, 4),
        "traffic": int(tr["hbm_bytes"]) if tr else None,
        "bytes_per_launch": int(b_comp), "ms_per_launch": round(float(bin_ms[dom]), 4),
        "products_per_launch": int(mdl["products"]),
        "model": "compulsory HBM bytes: (4+w) per entry of the launch's A rows and C rows + 12 per row + all of B once",
        "measured_hbm": ({"fetch_bytes": int(tr["fetch_bytes"]), "write_bytes": int(tr["write_bytes"]),
                          "achieved": gbs(tr["hbm_bytes"], t_dom),
                          "frac": round(gbs(tr["hbm_bytes"], t_dom) / HBM_PEAK_GBS, 4),
                          "over_compulsory": round(tr["hbm_bytes"] / max(b_comp, 1), 3),
                          "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of tools/pmc_one.py made by this run"}
                         if tr else None),
        "l2_requested": {"bytes": int(mdl["requested"]), "achieved": gbs(mdl["requested"], t_dom), "peak": L2_PEAK_GBS,
                         "frac": round(gbs(mdl["requested"], t_dom) / L2_PEAK_GBS, 4),
                         "model": "SURVEY 8d requested bytes: 12/row + (12+w)/A entry + (4+w)/product + (4+w)/C entry"},
        "lds_atomic": lds,
        "whole_call": {"bytes_requested_model": int(b_spgemm), "ms": round(ms_per_step, 4),
                       "l2_frac": round(gbs(b_spgemm, ms_per_step * 1e-3) / L2_PEAK_GBS, 4),
                       "compulsory_hbm_frac": round(gbs((4 + w) * (nnz_a + nnz_b + nnz_c) + 8 * a.M, ms_per_step * 1e-3) / HBM_PEAK_GBS, 4)},
        "note": "frac is the physical HBM fraction.  The numeric window kernels are bound by the latency of the "
                "dependent loads of a row group (row list -> row words -> A entries -> B extents -> B entries) and by "
                "instruction issue, not by HBM and not by the LDS atomics (ablations in DESIGN 4.1)",
    }

    # ------------------------------ regular brick (round 2's headline) + structure sweep ----
    def one_matrix(kind_i, dims_i, seed_i, phases=True):
        """The headline protocol on another matrix: warm loop, allocate-inside loop, phase pass."""
        Ai = synth(lib, kind_i, *dims_i, seed_i)
        ai = lib.csr_from_numpy(Ai["rpt"], Ai["col"], Ai["val"], Ai["N"])
        bi = lib.csr_from_numpy(Ai["rpt"], Ai["col"], Ai["val"], Ai["N"])
        lib.csr_memcpy(C.byref(ai))
        lib.csr_memcpy(C.byref(bi))
        fl_i = C.c_longlong()
        lib.get_spgemm_flop(C.byref(ai), C.byref(bi), ai.M, C.byref(fl_i))
        el_i, _, _, _, st_i = spgemm_loop(lib, ai, bi, args.steps, args.warmup, barrier)
        ms_i = el_i * 1e3 / args.steps
        rep = {"M": int(Ai["M"]), "nnz_A": int(Ai["rpt"][-1]), "n_prod": int(fl_i.value // 2), "nnz_C": int(st_i.nnz_c),
               "value": round(fl_i.value / (ms_i * 1e6), 2), "unit": "GFLOPS", "ms_per_step": round(ms_i, 4),
               "twin_rows": int(st_i.twin_rows)}
        if phases:
            lib.nsparse_set_workspace_cache(0)
            el_ir, _, _, _, _ = spgemm_loop(lib, ai, bi, args.steps, 1, barrier)
            lib.nsparse_set_workspace_cache(1)
            lib.nsparse_set_bin_timing(1)
            _, bin_i, sym_i, ph_i, st_i = spgemm_loop(lib, ai, bi, args.steps, 1, barrier, with_stats=True)
            lib.nsparse_set_bin_timing(0)
            rep.update({
                "reference_compatible_ms": round(el_ir * 1e3 / args.steps, 4),
                "reference_compatible_gflops": round(fl_i.value / (el_ir * 1e3 / args.steps * 1e6), 2),
                "phase_ms": {"setup": round(float(ph_i[0]), 4), "symbolic": round(float(ph_i[1]), 4),
                             "numeric": round(float(ph_i[2]), 4)},
                "numeric_bins_ms": [round(float(v), 4) for v in bin_i[:11]],
                "symbolic_bins_ms": [round(float(v), 4) for v in sym_i[:11]],
                "sym_bin_rows": list(st_i.sym_bin_size)[:11], "num_bin_rows": list(st_i.num_bin_size)[:11]})
        lib.release_csr(ai)
        lib.release_csr(bi)
        return rep

    regular = sweep = None
    if world == 1 and rank == 0 and not args.no_irregular:
        kind_b, dims_b, seed_b = STANDINS["cant_brick"]
        dims_b = (9, 9, planes)
        regular = one_matrix(kind_b, dims_b, seed_b)
        regular["workload"] = ("synthetic cant-class, REGULAR: 9x9x257 brick of 3-dof nodes, natural numbering "
                               "(nsparse_synth_csr kind 0) -- round 2's headline matrix: every row in the narrowest "
                               "window bin, twin rows neighbours")
        regular["suitesparse_cant"] = {"M": 62451, "nnz_A": 4007383, "n_prod": "~269.5 M", "nnz_C": "~17.4 M"}
        # how much of the speed hangs on rows that share a column pattern (the dof of a mesh node)
        sweep = {"what": "nsparse_synth_csr kind 6 = the headline stand-in with SCALAR perturbations on a share p of "
                         "the nodes (one dof constrained: row = diagonal, column gone; or one scalar coupling "
                         "dropped), which break the node's common column pattern; twins_off = the headline matrix with "
                         "NSPARSE_TWINS=0 (no twin detection, no node-block kernel: the floor)",
                 "points": []}
        for pm in (0, 100, 300, 1000):
            r6 = one_matrix(6, (9, 9, planes + (pm << 32)), 0x5EED0022, phases=False)
            sweep["points"].append({"p": pm / 1000.0, "gflops": r6["value"], "ms": r6["ms_per_step"],
                                    "twin_rows": r6["twin_rows"], "nnz_A": r6["nnz_A"], "n_prod": r6["n_prod"]})
        try:
            if dry:
                raise RuntimeError("dry run: tools/one_gflops.py needs the device")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_gflops.py"), "5", "9", "9", "257",
                                str(args.steps)], env=dict(os.environ, NSPARSE_TWINS="0"), capture_output=True,
                               text=True, timeout=300, cwd=ROOT)
            sweep["twins_off"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            sweep["twins_off"] = {"error": repr(e)[:160]}

    # ------------------------------------------------------------------- SpMV ----
    # Row-sharded AMB SpMV through the NATIVE library (include/nsparse_dist.h, libnsparse_dist_d.so): partition,
    # conversion, the per-iteration sequence [memset] -> kernel -> ncclAllGather -> [gap closing] and the timed loop
    # itself are C; so are the barriers and reductions around the loop (nsparse_dist_barrier / _allreduce_f64).  One
    # communicator for the whole run: the handle drops its matrix between the two workloads.
    def native_loop(d_y, d_x, gather, steps):
        check(dl.nsparse_dist_spmv_loop(h, d_y, d_x, gather, 2, None, None, None), "warm-up SpMV loop")
        mw, me, us = C.c_double(), C.c_double(), C.c_double()
        barrier()
        t = time.perf_counter()
        rc = dl.nsparse_dist_spmv_loop(h, d_y, d_x, gather, steps, C.byref(mw), C.byref(me), C.byref(us))
        barrier()
        check(rc, "nsparse_dist_spmv_loop")
        return max_over_ranks(time.perf_counter() - t) * 1e3 / steps, max_over_ranks(me.value), max_over_ranks(us.value)

    def emulated_gather(d_y, cuts, m_loc):
        """NSPARSE_BENCH_EMULATE: what the all-gather + gap closing do to THIS rank's rows, with the library's own copy
        kernel: the local rows go to this rank's share of a staging buffer (equal shares of the longest block) and
        nsparse_dist_close_gaps puts them where they belong in a second y.  Returns the rows it landed."""
        M = int(cuts[-1])
        rpr = max(1, int(np.max(np.diff(cuts))))
        staged = lib.dmalloc(world * rpr * w)
        lib.hip.hipMemset(staged, 0, world * rpr * w)
        d_y2 = lib.dmalloc((M + 64) * w)
        lib.hip.hipMemset(d_y2, 0, (M + 64) * w)
        d_cuts = lib.dmalloc(4 * (world + 1))
        lib.h2d(d_cuts, np.ascontiguousarray(cuts, dtype=np.int32))
        if m_loc > 0:
            assert lib.hip.hipMemcpy(C.c_void_p(staged.value + rank * rpr * w), C.c_void_p(d_y.value + int(cuts[rank]) * w),
                                     m_loc * w, 3) == 0  # device to device
        check(dl.nsparse_dist_close_gaps(d_y2, staged, d_cuts, world, rpr, M, dl.nsparse_dist_stream(h)), "nsparse_dist_close_gaps")
        check(dl.nsparse_dist_sync(h), "nsparse_dist_sync")
        out = lib.d2h(C.c_void_p(d_y2.value + int(cuts[rank]) * w), (m_loc,), np.float64)
        for p_ in (staged, d_y2, d_cuts):
            lib.dfree(p_)
        return out

    def spmv_report(A_rows, M_global, nnz_global, label, N_cols, blocks, traffic=None):
        xh = np.random.default_rng(1).random(N_cols + 20)
        cuts = np.array([blocks[0][0]] + [e for _, e in blocks], dtype=np.int32)
        # communicator-wide agreement after the set-up and ONE gathered SpMV: at N > 1 this is the first time the
        # RCCL path runs on real links -- a rank that failed must not leave the others in the collective
        err = ""
        csr = lib.csr_from_numpy(A_rows["rpt"], A_rows["col"], A_rows["val"], N_cols)
        lib.csr_memcpy(C.byref(csr))
        d_x = lib.dmalloc(xh.nbytes)
        lib.h2d(d_x, xh)
        plan = ns.sfPlan()
        lib.init_plan(C.byref(plan))
        rc = dl.nsparse_dist_spmv_setup(h, C.byref(csr), cuts.ctypes.data_as(ns.capi.c_int_p), d_x, C.byref(plan))
        if rc != 0:
            err = f"nsparse_dist_spmv_setup -> {rc}"
        ny = int(dl.nsparse_dist_y_elems(h)) if not err else 1
        d_y = lib.dmalloc((ny + 64) * w)
        lib.hip.hipMemset(d_y, 0, (ny + 64) * w)
        if world > 1 and not rdv.all_ok(not err, "SpMV set-up"):
            log(f"[rank {rank}] SpMV set-up failed on some rank ({err or 'another rank'}); no measurement")
            sys.exit(6)
        assert not err, err
        gather_flag = 1 if native_coll else 0
        check(dl.nsparse_dist_spmv(h, d_y, d_x, gather_flag), "first SpMV")
        check(dl.nsparse_dist_sync(h), "first SpMV (sync)")
        amb = dl.nsparse_dist_amb(h).contents
        fp = int(lib.nsparse_amb_footprint_bytes(C.byref(amb))) if A_rows["M"] > 0 else 0
        ms_c, ms_c_ev, us_c = native_loop(d_y, d_x, 0, args.spmv_steps)
        ms_g, ms_g_ev, us_g = native_loop(d_y, d_x, 1, args.spmv_steps) if native_coll else (ms_c, ms_c_ev, us_c)
        extra = {"driver": "native: libnsparse_dist_d.so (C loop, RCCL all-gather, no Python per iteration)",
                 "host_us_per_spmv": round(us_g, 2), "ms_events_with_gather": round(ms_g_ev, 5)}
        # the same sequence replayed from a hipGraph (one hipGraphLaunch per SpMV)
        if world == 1 or os.environ.get("NSPARSE_DIST_GRAPH") == "1":
            if dl.nsparse_dist_capture(h, d_y, d_x, gather_flag) == 0:
                ms_gr, ms_gr_ev, us_gr = native_loop(d_y, d_x, gather_flag, args.spmv_steps)
                extra["hipgraph"] = {"ms_per_spmv": round(ms_gr, 5), "ms_events": round(ms_gr_ev, 5),
                                     "host_us_per_spmv": round(us_gr, 2)}
            else:
                extra["hipgraph"] = {"error": int(dl.nsparse_dist_last_error())}

        def local_rows():
            check(dl.nsparse_dist_spmv(h, d_y, d_x, 0), "SpMV (check)")
            check(dl.nsparse_dist_sync(h), "SpMV (check, sync)")
            return lib.d2h(C.c_void_p(d_y.value + int(cuts[rank]) * w), (A_rows["M"],), np.float64)
        plan_o, amb_o = plan, amb
        if emulate and world > 1:
            y_loc = local_rows()
            landed = emulated_gather(d_y, cuts, A_rows["M"])
            extra["emulated_gather"] = {"rows": int(A_rows["M"]), "landed_equal": bool(np.array_equal(landed, y_loc))}
            assert extra["emulated_gather"]["landed_equal"], "staging + gap closing moved a row to the wrong place"
        # x is counted once over N instead of the reference's second M*w term
        b_amb = fp - A_rows["M"] * w + N_cols * w
        fp_all = sum_over_ranks(b_amb)
        b_csr = nnz_global * (w + 4) + 4 * (M_global + 1) + N_cols * w + M_global * w
        atom = "true" if amb_o.seg_num > 1 else "false"
        _, tr_s = find_kernel(traffic, [f"k_spmv_amb_row<{int(plan_o.block_size)}, {atom}", f"k_spmv_amb_pipe<{int(plan_o.block_size)}, {atom}",
                                        f"k_spmv_amb<{int(plan_o.block_size)}, {int(amb_o.chunk)}, {atom}"])
        rep = {
            "workload": label, "M": M_global, "nnz": int(nnz_global),
            "plan": {"seg_size": int(plan_o.seg_size), "block_size": int(plan_o.block_size),
                     "thread_block": int(plan_o.thread_block), "chunk": int(amb_o.chunk),
                     "seg_num": int(amb_o.seg_num)},
            "ms_per_spmv": round(ms_g, 5), "ms_compute_only": round(ms_c, 5),
            "ms_kernel_events": round(ms_c_ev, 5),
            "value": round(fp_all / (ms_g * 1e-3) / 1e9, 1), "unit": "GB/s",
            "gbs_compute_only": round(fp_all / (ms_c * 1e-3) / 1e9, 1),
            "frac_hbm_peak": round(fp_all / (ms_g * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "frac_hbm_peak_kernel_events": round(b_amb / (ms_c_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "gbs_csr_model": round(b_csr / (ms_g * 1e-3) / 1e9, 1),
            "gflops_ref": round(2.0 * nnz_global / (ms_g * 1e6), 2),
            "bytes_amb_model": int(fp_all),
            "traffic": int(tr_s["hbm_bytes"]) if tr_s else None,
            **extra,
        }
        # parity spot check against the library's own CPU path (csr_kernel) on rank rows
        if A_rows["M"] > 0:
            y = local_rows()
            m = lib.csr_from_numpy(A_rows["rpt"], A_rows["col"], A_rows["val"], N_cols)
            xs = np.ascontiguousarray(xh[:N_cols])
            yr = np.zeros(A_rows["M"])
            lib.csr_kernel(yr.ctypes.data_as(C.c_void_p), C.byref(m), xs.ctypes.data_as(C.c_void_p))
            rep["ans_check_fails"] = int(lib.nsparse_ans_check_count(
                yr.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), A_rows["M"]))
            assert rep["ans_check_fails"] == 0 or os.environ.get("NSPARSE_SPMV_ABL"), "AMB SpMV differs from csr_kernel beyond the reference tolerance"
        # vendor csrmv on the same device arrays (N = 1 only: informational)
        if world == 1 and not args.no_vendor:
            try:
                vl = ns.load_vendor("d")
                d_yv = lib.dmalloc((A_rows["M"] + 64) * w)
                ms_v = float(vl.nsparse_vendor_spmv_csr(d_yv, C.byref(csr), d_x, 50))
                lib.dfree(d_yv)
                rep["vendor_csrmv"] = {"library": "rocSPARSE csrmv (adaptive), C API", "ms": round(ms_v, 5),
                                       "gbs_csr_model": round(b_csr / (ms_v * 1e-3) / 1e9, 1),
                                       "err": int(vl.nsparse_vendor_last_error())}
            except Exception as e:
                rep["vendor_csrmv"] = {"error": repr(e)[:160]}
        check(dl.nsparse_dist_release_matrix(h), "nsparse_dist_release_matrix")  # the communicator stays
        lib.release_csr(csr)
        lib.dfree(d_x)
        lib.dfree(d_y)
        return rep

    nnz_glob = int(A_full["rpt"][-1])
    blocks1 = row_partition_nnz(A_full["rpt"], world)
    A_spmv = A_full if world == 1 else csr_row_block(A_full, *blocks1[rank])
    spmv = spmv_report(A_spmv, A_full["M"], nnz_glob, f"{src} (same matrix as SpGEMM)", A_full["N"], blocks1, traffic_all)
    spmv_hbm = None
    if not args.no_large:
        kind2, (gx, gy, gz), seed2 = STANDINS["nlpkkt120"]
        if dry:
            gx, gy, gz = 16, 16, 8
        M2 = gx * gy * gz
        # the stand-in has the same 27 entries in every interior row, so equal row counts ARE the
        # nnz-balanced cut; a file would be cut by its row pointers
        _, blocks = row_partition(M2, world)
        t0 = time.time()
        A2, src2 = load_or_synth(lib, "nlpkkt120", kind2, (gx, gy, gz), seed2, rows=blocks[rank] if world > 1 else (0, 0))
        if world > 1 and A2["M"] == M2:   # a file was loaded whole: cut it by nnz
            blocks = row_partition_nnz(A2["rpt"], world)
            A2 = csr_row_block(A2, *blocks[rank])
        nnz2 = sum_over_ranks(int(A2["rpt"][-1]))
        log(f"[rank {rank}] {src2}: rows {A2['M']} nnz {int(A2['rpt'][-1])} ({time.time() - t0:.1f}s)")
        spmv_hbm = spmv_report(A2, M2, int(nnz2), src2, M2, blocks, traffic_all)
        spmv_hbm["scaling"] = "strong"
        del A2

    # ------------------------------------------------ vendor baseline (rocSPARSE, C API) ----
    # The reference samples print their numbers next to cuSPARSE (spgemm_cu_csr / spmv_cu_csr,
    # SURVEY 8f rank 3); here libnsparse_vendor_d.so: rocsparse_csrgemm_nnz + rocsparse_dcsrgemm.
    vendor = None
    if rank == 0 and world == 1 and not args.no_vendor:
        try:
            vl = ns.load_vendor("d")
            cv = ns.sfCSR()
            msd = C.c_float()
            vl.nsparse_vendor_spgemm(C.byref(a), C.byref(b), C.byref(cv), C.byref(msd))  # warm-up
            assert vl.nsparse_vendor_last_error() == 0, f"rocSPARSE error {vl.nsparse_vendor_last_error()}"
            assert cv.nnz == nnz_c, f"rocSPARSE nnz(C) {cv.nnz} != {nnz_c}"
            v_rpt = lib.d2h(cv.d_rpt, (cv.M + 1,), np.int32)
            assert np.array_equal(v_rpt, crpt), "rocSPARSE C.rpt differs from the library's"
            vl.nsparse_vendor_release_csr(cv)
            reps_v, dev_ms = 5, 0.0
            lib.hip.hipDeviceSynchronize()
            t = time.perf_counter()
            for _ in range(reps_v):
                vl.nsparse_vendor_spgemm(C.byref(a), C.byref(b), C.byref(cv), C.byref(msd))
                dev_ms += msd.value
                vl.nsparse_vendor_release_csr(cv)
            ms_v = (time.perf_counter() - t) * 1e3 / reps_v
            vendor = {"library": "rocSPARSE csrgemm through its C API (libnsparse_vendor_d.so, no torch)",
                      "spgemm_ms_whole_call": round(ms_v, 3), "spgemm_gflops": round(flop.value / (ms_v * 1e6), 1),
                      "spgemm_ms_device_stages": round(dev_ms / reps_v, 3),
                      "spgemm_gflops_device_stages": round(flop.value / (dev_ms / reps_v * 1e6), 1),
                      "spgemm_nnz_c": int(nnz_c), "structure_equal": True,
                      "speedup_whole_call": round(ms_v / ms_per_step, 2)}
        except AssertionError:
            raise
        except Exception as e:
            vendor = {"error": repr(e)[:200]}

    # ------------------------------------------ BASELINE configs 3 and 5 + the stencil ----
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        t0 = time.time()
        lib.nsparse_trim_workspace()  # the children want the memory this process has cached
        deadline = t0 + args.configs_budget
        configs = {"protocol": "spgemm_hash.cu:35-54: one warm-up call, then the mean of the timed whole calls; every "
                               "config in its own torch-free process (tools/bench_config.py)",
                   "cases": [run_config(cs, not args.no_pmc, deadline) for cs in ("webbase1m", "stencil", "rmat22")]}
        configs["seconds"] = round(time.time() - t0, 1)
        log(f"[configs] {configs['seconds']} s")

    # ----------------------------------------------------------- CPU baseline ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # OpenMP placement is read when the runtime starts: one thread per core, spread over the sockets
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
        from oracle.oracle import Oracle  # checker / baseline leg only
        orc = Oracle("d")
        cores = os.cpu_count() or 1
        # a bounded sample of single-core work: the whole product, repeated until about 10 s have gone by
        ref, t1, _, _ = orc.spgemm_omp_timed(A_loc, A_full, reps=1, threads=1)
        n1 = int(min(12, max(1, round(10.0 / max(t1, 1e-3))))) if t1 < 5.0 else 0
        t_all = t1
        if n1:
            _, tb, tm, _ = orc.spgemm_omp_timed(A_loc, A_full, reps=n1, threads=1)
            t_all += tm * n1
            t1 = min(t1, tb)
        ref_n, tn, tn_mean, nth = orc.spgemm_omp_timed(A_loc, A_full, reps=5, threads=0)
        assert ref["nnz"] == nnz_c and np.array_equal(ref["rpt"], crpt), "GPU structure != oracle"
        assert np.array_equal(ref_n["rpt"], crpt) and np.array_equal(ref_n["col"], ref["col"]), "all-cores oracle != 1-core oracle"
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
            "sample": f"whole {src} matrix, C=A^2 {n1 + 1} times on one core ({t_all:.1f} s, best time), oracle/nsparse_oracle.c: "
                      "orc_spgemm_omp_timed -- marker-array symbolic + numeric, ascending window sweep instead of a sort for "
                      "narrow rows (the reference has no CPU SpGEMM)",
            "all_cores": {"value": round(flop.value / tn / 1e9, 3), "cores": int(nth), "host_cpus": cores,
                          "speedup_over_one_core": round(t1 / tn, 1), "mean_value": round(flop.value / tn_mean / 1e9, 3),
                          "how": "the same function on all host cores: rows in dynamic chunks of M / (8 threads), threads "
                                 "bound to cores (OMP_PROC_BIND=spread, OMP_PLACES=cores), 5 repetitions after a warm-up, best"},
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
            **({"emulated_ranks": "NSPARSE_BENCH_EMULATE=1: ranks share the GPUs that exist, no communicator -- a "
                                  "control-flow test, NOT a measurement"} if emulate and world > 1 else {}),
            **({"emulated_device": "libnsparse was built by tests/emu (NSPARSE_LIB_DIR points at it): the kernels ran on "
                                   "host threads -- answers are checked, every time is host time, NOT a measurement"}
               if emulated_device else {}),
            **({"dry_run": "NSPARSE_BENCH_DRYRUN=1: no device, every time and counter is made up (tools/bench_dry.py) -- "
                           "a test of this script's control flow, NOT a measurement"} if dry else {}),
            "runtime": runtime_report(),
            "config": {"workload": f"{src}: 3-dof 27-pt FEM brick 9x9x{nz}, {M_glob} rows, C=A^2 by 1-D row blocks of 62451 rows"
                                   if "synthetic" in src else src,
                       "rows_per_gpu": int(a.M), "nnz_A_per_gpu": nnz_a, "n_prod_per_gpu": n_prod,
                       "nnz_C_per_gpu": int(nnz_c), "parallelism": f"row-partition x{world}, B replicated",
                       "timing": "whole spgemm_kernel_hash call, workspace from the library's block cache, per-bin events off",
                       "reference_compatible_gflops": round(flops_all / (ms_ref * 1e6), 2),
                       "reference_compatible_ms": round(ms_ref, 4)},
            "timing": {"warm_ms": round(ms_per_step, 4), "warm_gflops": round(gflops, 2),
                       "reference_compatible_ms": round(ms_ref, 4),
                       "reference_compatible_gflops": round(flops_all / (ms_ref * 1e6), 2),
                       "reference_compatible": "nsparse_set_workspace_cache(0): every call hipMalloc / hipFree's its "
                                               "workspaces and C like spgemm_hash.cu:35-54",
                       "alloc_async_ms": round(ms_async, 4),
                       "alloc_async_gflops": round(flops_all / (ms_async * 1e6), 2),
                       "alloc_async": "nsparse_set_workspace_cache(2): no cache either; every array of the call comes from "
                                      "hipMallocAsync and goes back with hipFreeAsync (default pool keeps freed memory)"},
            "phase_ms": {"setup": round(float(phase[0]), 4), "symbolic": round(float(phase[1]), 4),
                         "numeric": round(float(phase[2]), 4), "total_events": round(float(phase[3]), 4),
                         "numeric_bins": [round(float(v), 4) for v in bin_ms[:11]],
                         "symbolic_bins": [round(float(v), 4) for v in sym_ms[:11]],
                         "sym_bin_rows": list(st.sym_bin_size)[:11], "num_bin_rows": list(st.num_bin_size)[:11],
                         "note": "separate pass with per-bin events on"},
            "roofline": roofline,
            "configs": configs,
            "regular_brick": regular,
            "structure_sweep": sweep,
            "cpu_baseline": cpu,
            "vendor_baseline": vendor,
            "spmv": spmv,
            "spmv_hbm": spmv_hbm,
            "wall_s": round(time.time() - t_start, 1),
        }
        C.CDLL(None).fflush(None)  # the library's own stdio lines ("Read mtx file: ...") go out first
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    barrier()
    dl.nsparse_dist_destroy(h)
    rdv.close()


if __name__ == "__main__":
    main()
