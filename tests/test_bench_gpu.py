"""bench.py's multi-rank control flow on ONE GPU (verdict r03 item 3): the N = 2 path -- rank processes, rendezvous,
partition, per-rank set-up, native loops, result assembly, JSON -- with both ranks on the one device and the
collective replaced by the library's own staging + gap-closing copy (NSPARSE_BENCH_EMULATE=1); `--gpus 2` WITHOUT the
emulation must fail fast with a message; the thread-per-GPU sample at world 1; the watchdog of the native library."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import nsparse_amd as ns
from gpu_util import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "2", "--warmup", "1", "--spmv-steps", "5", "--no-cpu", "--no-pmc", "--no-vendor", "--no-irregular",
         "--no-configs"]


def test_two_emulated_ranks_run_the_whole_flow():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + QUICK, capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, NSPARSE_BENCH_EMULATE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "emulated_ranks" in d and d["value"] > 0
    assert d["runtime"]["torch_imported"] is False
    assert d["config"]["rows_per_gpu"] == 62451
    for leg in ("spmv", "spmv_hbm"):
        assert d[leg]["ans_check_fails"] == 0
        assert d[leg]["emulated_gather"]["landed_equal"] is True
        assert d[leg]["driver"].startswith("native")
    # the nlpkkt-class rows were cut in two
    assert d["spmv_hbm"]["M"] == 3542400 and d["spmv_hbm"]["nnz"] > 9e7


def test_more_ranks_than_gpus_fails_fast_with_a_message():
    ndev = ns.load_dist("d").nsparse_dist_device_count()
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1)] + QUICK,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "one rank per GPU" in r.stderr and f"{ndev} GPU" in r.stderr
    assert time.time() - t0 < 120, "the refusal must not wait for any time-out"
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_single_rank_line_is_torch_free_and_on_the_system_runtime():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + QUICK + ["--no-large"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rt = d["runtime"]
    assert rt["torch_imported"] is False and rt["system_rocm_runtime"] is True, rt
    assert any(k.startswith("libamdhip64") for k in rt["mapped"])
    assert d["roofline"]["frac"] > 0 and d["spmv"]["ans_check_fails"] == 0


def test_thread_per_gpu_sample_at_world_one(tmp_path):
    lib = ns.load("d")
    A = synth(lib, 0, 5, 5, 12, seed=3)
    m = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    path = str(tmp_path / "brick.mtx")
    assert lib.nsparse_write_mtx(C.byref(m), path.encode(), 0) == 0
    exe = os.path.join(os.path.dirname(lib.path), "amb_dist_d")
    r = subprocess.run([exe, path, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert "on 1 GPUs" in r.stdout and "Calculation Result is Correct" in r.stdout
    # more GPUs than the box has: a message and a code, nothing hangs
    ndev = ns.load_dist("d").nsparse_dist_device_count()
    r = subprocess.run([exe, path, str(ndev + 1)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "visible" in r.stderr


def test_watchdog_and_reductions_of_the_native_library():
    lib, dl = ns.load("d"), ns.load_dist("d")
    assert dl.nsparse_dist_device_count() >= 1
    old = dl.nsparse_dist_set_timeout(5.0)
    assert old > 0 and dl.nsparse_dist_set_timeout(old) == 5.0
    # one-rank communicator: barrier and reductions are real RCCL calls
    ident = C.create_string_buffer(ns.DIST_ID_BYTES)
    assert dl.nsparse_dist_unique_id(ident) == 0
    h = C.c_void_p()
    assert dl.nsparse_dist_init(C.byref(h), ident, 0, 1) == 0
    assert dl.nsparse_dist_barrier(h) == 0
    v = (C.c_double * 3)(1.5, -2.0, 7.0)
    assert dl.nsparse_dist_allreduce_f64(h, v, 3, 0) == 0 and list(v) == [1.5, -2.0, 7.0]
    assert dl.nsparse_dist_allreduce_f64(h, v, 3, 1) == 0 and list(v) == [1.5, -2.0, 7.0]
    assert dl.nsparse_dist_allreduce_f64(h, v, 65, 0) == -1
    # a matrix, released, and the next one on the same handle
    for kind, dims in ((0, (4, 4, 8)), (3, (10, 8, 0))):
        A = synth(lib, kind, *dims, seed=11)
        full = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
        lib.csr_memcpy(C.byref(full))
        x = np.random.default_rng(5).random(A["N"] + 20)
        d_x = lib.dmalloc(x.nbytes)
        lib.h2d(d_x, x)
        plan = ns.sfPlan()
        lib.init_plan(C.byref(plan))
        cuts = np.array([0, A["M"]], dtype=np.int32)
        assert dl.nsparse_dist_spmv_setup(h, C.byref(full), cuts.ctypes.data_as(ns.capi.c_int_p), d_x, C.byref(plan)) == 0
        # a second set-up without a release is refused and leaves the handle as it was
        assert dl.nsparse_dist_spmv_setup(h, C.byref(full), cuts.ctypes.data_as(ns.capi.c_int_p), d_x, C.byref(plan)) == -3
        d_y = lib.dmalloc((int(dl.nsparse_dist_y_elems(h)) + 64) * 8)
        assert dl.nsparse_dist_spmv(h, d_y, d_x, 1) == 0 and dl.nsparse_dist_sync(h) == 0
        y = lib.d2h(d_y, (A["M"],), np.float64)
        ref = np.zeros(A["M"])
        lib.csr_kernel(ref.ctypes.data_as(C.c_void_p), C.byref(full), np.ascontiguousarray(x[:A["N"]]).ctypes.data_as(C.c_void_p))
        assert lib.nsparse_ans_check_count(ref.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), A["M"]) == 0
        assert dl.nsparse_dist_release_matrix(h) == 0
        lib.release_csr(full)
        lib.dfree(d_x)
        lib.dfree(d_y)
    # bad cuts are refused before anything changes
    A = synth(lib, 0, 3, 3, 4, seed=1)
    full = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(full))
    plan = ns.sfPlan()
    lib.init_plan(C.byref(plan))
    bad = np.array([0, A["M"] - 1], dtype=np.int32)
    assert dl.nsparse_dist_spmv_setup(h, C.byref(full), bad.ctypes.data_as(ns.capi.c_int_p), None, C.byref(plan)) == -2
    assert dl.nsparse_dist_y_elems(h) == 0
    lib.release_csr(full)
    dl.nsparse_dist_destroy(h)


def test_a_peer_that_never_joins_is_a_timeout_not_a_hang():
    """Rank 0 of a two-rank communicator whose rank 1 never starts: ncclCommInitRank is given the watchdog's time and the
    call comes back with -8."""
    code = ("import ctypes as C, sys, time; sys.path.insert(0, %r); import nsparse_amd as ns\n"
            "dl = ns.load_dist('d'); dl.nsparse_dist_set_timeout(4.0)\n"
            "ident = C.create_string_buffer(ns.DIST_ID_BYTES); assert dl.nsparse_dist_unique_id(ident) == 0\n"
            "h = C.c_void_p(); t = time.time(); rc = dl.nsparse_dist_init(C.byref(h), ident, 0, 2)\n"
            "print('RC', rc, round(time.time() - t, 1)); sys.stdout.flush(); import os; os._exit(0)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RC")]
    assert line, (r.stdout[-300:], r.stderr[-600:])
    rc, secs = int(line[0].split()[1]), float(line[0].split()[2])
    assert rc != 0 and secs < 60
