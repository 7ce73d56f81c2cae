"""The product's kernels on the CPU emulation of the device model (tests/emu, see its README): the SAME `-m gpu` tests,
the SAME ctypes binding, pointed at tests/emu/lib -- libraries built from nsparse_amd/csrc with the host compiler
against a lane-by-lane emulation of wavefronts, LDS, DPP / swizzle / bpermute, barriers and atomics.

This is how the logic of every kernel is checked against the oracle on a box without a GPU (the device pool was closed
for most of rounds 4 and 5).  It does not replace the device run -- timing, hazards and the memory model are out of its
reach -- and nothing under nsparse_amd/ ever loads these libraries.  The selections below are sized for the CPU suite
(about three minutes together); the whole -m gpu corpus takes ~50 minutes on the emulation:
    NSPARSE_LIB_DIR=$PWD/tests/emu/lib python -m pytest tests -m gpu -q
"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU, "lib")


@pytest.fixture(scope="module")
def emu_lib():
    """tests/emu/lib, built on demand (`make -C tests/emu`, ~4 min on 8 cores; __graft_entry__.build() does it too)."""
    need = ["libnsparse_d.so", "libnsparse_s.so", "libnsparse_dist_d.so", "amb_dist_d", "selftest"]
    r = subprocess.run(["make", "-C", EMU, "-j", str(min(8, os.cpu_count() or 1)), "-s"], capture_output=True, text=True,
                       timeout=3000)
    assert r.returncode == 0, r.stderr[-3000:]
    for n in need:
        assert os.path.exists(os.path.join(LIB, n)), n
    return LIB


def _exp(emu_lib):
    """The emulation build of the -DNSPARSE_EXPERIMENTS variant (tests/emu/lib_exp): the only library that reads the
    measurement switches (NSPARSE_FUSED_FORCE, NSPARSE_TB_LEAN, ...) and carries the opt-in kernel families."""
    d = emu_lib.rstrip("/") + "_exp"
    if not os.path.exists(os.path.join(d, "libnsparse_d.so")):
        pytest.skip("tests/emu/lib_exp not built (make -C tests/emu EXTRA=-DNSPARSE_EXPERIMENTS OUT=.../lib_exp libs)")
    return d


def _gpu_tests_on_emu(emu_lib, args, env=None, timeout=900, expect_min=1):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import EXPERIMENT_SWITCHES
    if any(k in EXPERIMENT_SWITCHES for k in (env or {})):
        emu_lib = _exp(emu_lib)
    e = dict(os.environ, NSPARSE_LIB_DIR=emu_lib, **(env or {}))
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT,
                       env=e, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-4000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert r.returncode == 0 and m and "failed" not in r.stdout.splitlines()[-1], tail
    assert int(m.group(1)) >= expect_min, tail
    return int(m.group(1))


def test_emulated_instructions_against_hand_checked_kernels(emu_lib):
    """tests/emu/selftest.cpp: DPP scans (row_shr / row_bcast with row masks), shuffles of every kind and width, ballot
    under divergence, swizzle, bpermute, readlane, the v_min / v_max_i32_dpp of the asm sorts parsed from their operand
    strings, LDS + barriers, dynamic LDS, global atomics across workgroups, a grid barrier of co-resident workgroups."""
    r = subprocess.run([os.path.join(emu_lib, "selftest")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "selftest: 0 failures" in r.stdout, r.stdout + r.stderr
    # round 6: the pool as 4 compute units of 160 KiB -- a kernel with 96 KiB of (static + dynamic) LDS runs one workgroup per
    # CU, one with 40 KiB up to four (capped by the 8 workers), and 161 KiB is refused at launch (the selftest checks all three)
    r = subprocess.run([os.path.join(emu_lib, "selftest")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, EMU_CUS="4", EMU_WORKERS="8"))
    assert r.returncode == 0 and "selftest: 0 failures" in r.stdout, r.stdout + r.stderr
    import re
    m = re.search(r"selftest: 4 CUs; resident workgroups of a 96 KiB kernel (\d+), of a 40 KiB kernel (\d+)", r.stdout)
    assert m and 1 <= int(m.group(1)) <= 4 and int(m.group(1)) <= int(m.group(2)) <= 8, r.stdout


def test_static_lds_of_the_products_kernels_is_policed(emu_lib):
    """The emulation cannot see how much static LDS a kernel declares (it is thread-local storage there); the table the REAL
    compiler produced for gfx950 sits next to the library (tests/emu/Makefile: <lib>.lds) and is what the launch check and the
    co-residency of every launch are computed from.  It must be there, cover the big-LDS kernels, and stay within a CU."""
    for p in ("d", "s"):
        path = os.path.join(emu_lib, f"libnsparse_{p}.so.lds")
        assert os.path.exists(path), f"{path} missing (make -C tests/emu lds; needs hipcc)"
        rows = [ln.split() for ln in open(path)]
        lds = {r[0]: int(r[1]) for r in rows}
        assert len(lds) > 300
        big = {k: v for k, v in lds.items() if "k_num_ranked" in k or "k_sym_tb" in k or "k_num_tiled" in k}
        assert big and max(big.values()) > 128 * 1024, "the heavy-row kernels' static LDS did not reach the table"
        assert max(lds.values()) <= 160 * 1024


def test_the_library_loaded_is_the_emulation_and_not_the_product(emu_lib):
    """The emulation exports emu_get_stats; the product library must not (nothing CPU-side hides behind the product's
    name in nsparse_amd/lib), and nsparse_amd.load() without NSPARSE_LIB_DIR never looks at tests/emu."""
    emu = C.CDLL(os.path.join(emu_lib, "libnsparse_d.so"), mode=C.RTLD_LOCAL)
    assert hasattr(emu, "emu_get_stats")
    prod = C.CDLL(os.path.join(ROOT, "nsparse_amd", "lib", "libnsparse_d.so"), mode=C.RTLD_LOCAL)
    assert not hasattr(prod, "emu_get_stats") and not hasattr(prod, "emu_reset_stats")
    src = open(os.path.join(ROOT, "nsparse_amd", "capi.py")).read()
    assert "emu" not in src.replace("enumerate", "")


def test_random_products_on_the_emulation(emu_lib):
    """tests/test_fuzz_gpu.py (random rectangular / node-block / general-B products, random AMB plans, both precisions)
    vs the oracle."""
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_fuzz_gpu.py"],
                          env={"NSPARSE_FUZZ_SEEDS": "10", "NSPARSE_FUZZ_SQ_SEEDS": "6", "NSPARSE_FUZZ_AB_SEEDS": "4",
                               "NSPARSE_FUZZ_AMB_SEEDS": "8"})
    assert n == 32  # 10 + 6 + 4 + 8 random cases and the four bricks with 2 / 6 unknowns per node


def test_both_hash_kernel_families_on_the_emulation(emu_lib):
    """The lean kernels of round 4 (owner-array walk, 24-bit hash, in-register DPP sort, flip-form LDS sort) had never
    run anywhere after their hash fix; here every hash bin of both phases goes through them (NSPARSE_TB_LEAN=3) and
    through round 3's kernels: stencil, R-MAT and web-graph rows, one-wavefront rows, clustered columns in the big
    tables.  (First run on the emulation found the cleared table of one-wavefront rows unpublished before the direct
    rounds -- correct on the device by the wavefront's program order, now a wave_lds_sync().)"""
    # ([0], round 3's kernels, is proven on the device and rides in test_spgemm_paths / the fuzz corpus: here the lean ones)
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_spgemm_gpu.py", "-k",
                                    "(both_kernel_families or one_wavefront_bin or big_table_bins) and 3"])
    assert n == 3


def test_spgemm_paths_on_the_emulation(emu_lib):
    """Known answers, golden vectors, fused tails vs kernel chains (a real grid barrier among 8 workgroup threads), twin
    rows and keyed runs of the node-block kernel, non-finite values, rectangular / empty inputs, unsorted B, the
    numeric-only and unsorted-output modes.  (The whole file, with the chained product that found k_num_tiled's sweep
    record and the heavy-row kernels, is part of tools/emu_corpus.sh.)"""
    k = ("known_answer or golden or rectangular or empty or no_rows or unsorted or fused_tails_match or fused_tails_at "
         "or twin_rows or keyed_runs or non_finite or node_block_kernel or workspace_cache or wide_windows "
         "or window_wider")
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_spgemm_gpu.py", "-k", k], timeout=1200)
    assert n >= 24


def test_amb_conversion_and_spmv_on_the_emulation(emu_lib):
    """tests/test_amb_gpu.py: the seven AMB arrays bit for bit against the oracle (both chunk sizes, both precisions),
    the plan search, sigma windows, and the split-row SpMV kernel of round 4 (never run on a device)."""
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_amb_gpu.py"])
    assert n >= 30


def test_aux_modes_on_the_emulation(emu_lib):
    """Deterministic summation (bytes identical to the sequential oracle), stream-ordered workspace, fused-state query."""
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_aux_gpu.py", "-k", "deterministic or stream_ordered or fused_state"])
    assert n >= 8


def test_native_multi_rank_library_on_the_emulation(emu_lib):
    """libnsparse_dist on the in-process RCCL stand-in: one-rank end to end with graph replay, ragged partitions, gap
    closing, the native row-partitioned SpGEMM with its gather (never run on a device), the CLI samples."""
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_dist_native_gpu.py", "tests/test_partition_gpu.py", "tests/test_samples_gpu.py",
                                    "-k", "not cant_class and not webbase_class"])
    assert n >= 25


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_sharded_spmv_at_world_gt_1_one_thread_per_device(world, emu_lib, tmp_path):
    """samples/amb_dist.cpp as it runs on an 8-GPU node -- ncclCommInitAll, one thread per device, nnz-balanced 64-aligned
    row blocks, per-rank conversion, all-gather in place (equal blocks) or staged + gap closing (ragged), rank 0's y
    checked against csr_kernel by the reference's own rule -- on EMU_DEVICES fake devices with threads as ranks.  The
    first executions of the native path at world > 1 anywhere."""
    sys.path.insert(0, ROOT)
    import nsparse_amd as ns
    lib = ns.load("d")  # host entry points only: generator + Matrix Market writer
    m = ns.sfCSR()
    lib.nsparse_synth_csr(C.byref(m), 3, 12, 8, 0, 0x5EED0022, 0, 0)  # R-MAT scale 12: ragged nnz-balanced blocks
    path = str(tmp_path / "rmat12.mtx")
    assert lib.nsparse_write_mtx(C.byref(m), path.encode(), 0) == 0
    lib.release_cpu_csr(m)
    r = subprocess.run([os.path.join(emu_lib, "amb_dist_d"), path, str(world)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"on {world} GPUs" in r.stdout and "Calculation Result is Correct" in r.stdout
    assert len(re.findall(r"^rank \d+: rows", r.stdout, flags=re.M)) == world


@pytest.mark.parametrize("world", [3, 8])
def test_row_partitioned_spgemm_with_gather_at_world_gt_1(world, emu_lib):
    """tests/emu/ranks_spgemm.py: ranks as threads on the fake devices, ONE communicator by unique id, product-balanced
    row blocks through nsparse_dist_spgemm, nsparse_dist_barrier / _allreduce_f64, then nsparse_dist_spgemm_gather (size
    all-reduce, allocation agreed among the ranks, one broadcast per rank and array, row-pointer shift): every rank ends
    up with the whole C = the oracle's, structure bit for bit."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "ranks_spgemm.py"), str(world), "3", "11", "8", "0"],
                       env=dict(os.environ, NSPARSE_LIB_DIR=emu_lib), capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(lines[-1])
    assert not d["errors"] and len(d["ranks"]) == world
    for q in d["ranks"]:
        assert q["rpt_ok"] and q["col_ok"] and q["val_fails"] == 0
        assert q["nnz_sum"] == d["nnz_C"] and q["rank_sum"] == world * (world - 1) // 2
    assert sum(q["block_rows"] for q in d["ranks"]) == d["M"]


@pytest.mark.parametrize("gpus", [int(g) for g in os.environ.get("NSPARSE_EMU_BENCH_GPUS", "2").split(",")])  # (1,2: both)
def test_bench_py_itself_on_the_emulated_device(gpus, emu_lib):
    """The REAL bench.py (not its dry run): every library call of the timed protocol executes, on the emulated device,
    at full size (cant-class 62,451 rows per rank: ~2.5 s per product here).  N = 2 is the command the driver's scaling
    run uses -- `python bench.py --gpus 2` spawns two rank PROCESSES, rank 0's ncclUniqueId travels over the rendezvous,
    `ncclCommInitRank` / barriers / all-reduces / the all-gather of y are the native library's calls into the RCCL
    stand-in, whose ranks meet in POSIX shared memory (tests/emu/emu_rccl.cpp).  `--gpus 8` (8 processes, both SpMV
    workloads) runs the same way in ~5 min: `EMU_WORKERS=2 NSPARSE_LIB_DIR=tests/emu/lib python bench.py --gpus 8 --no-pmc
    --no-vendor --no-configs --no-cpu`.  Numbers from such a run are emulation speed -- what is checked is that the line
    comes out, complete, with the answers right."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--spmv-steps", "2",
           "--no-pmc", "--no-vendor", "--no-configs", "--no-large", "--no-irregular", "--no-cpu"]
    env = dict(os.environ, NSPARSE_LIB_DIR=emu_lib)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NSPARSE_BENCH_EMULATE"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:400]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["value"] > 0 and "dry_run" not in d and "emulated_ranks" not in d
    # ... and the line says what produced it: no HIP runtime is mapped, the library is the emulation build
    assert "NOT a measurement" in d["emulated_device"] and d["runtime"]["system_rocm_runtime"] is False
    assert d["config"]["rows_per_gpu"] == 62451 and d["roofline"]["bytes_per_launch"] > 0
    assert d["spmv"]["ans_check_fails"] == 0 and d["spmv"]["driver"].startswith("native: libnsparse_dist_d.so")
    if gpus > 1:
        assert d["config"]["parallelism"].startswith("row-partition x2") and d["spmv"]["ms_per_spmv"] > 0
    else:
        assert "k_num_block<128, 1536" in d["roofline"]["kernel"] and d["spmv"]["hipgraph"]["ms_per_spmv"] > 0


@pytest.mark.parametrize("stall,tails_run", [("", 6), ("k_setup_tail:2:1500", 1), ("k_numeric_setup:1:1500", 2)])
def test_grid_barrier_that_times_out_falls_back_to_the_chains(stall, tails_run, emu_lib):
    """fused.h: the grid barrier of the fused tails has ONE agreed outcome word (open / passed / failed by compare-and-swap,
    50 ms bound) -- re-written in round 4 without a device.  Here the census is skipped (NSPARSE_FUSED_FORCE=1), the
    tails run as 5 workgroups on 5 threads, and EMU_STALL holds one workgroup of the first or of the second tail back for
    1.5 s: the others time out, the late one READS "failed" and leaves its rows alone, the host sees the flag, repeats
    the call with the kernel chains (same C as the oracle), counts one fallback and stops fusing on that context."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "stalled_workgroup.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT,
                       env=dict(os.environ, NSPARSE_LIB_DIR=_exp(emu_lib), NSPARSE_FUSED_FORCE="1", EMU_STALL=stall, EMU_TRACE="1",
                                EMU_CLOCK_DIV="200"))  # (the 50 ms bound is ~0.5 s of a busy host's time)
    assert r.returncode == 0, r.stderr[-2000:]
    calls = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("[")][-1])
    assert len(calls) == 3 and all(c["equal"] and c["err"] == 0 for c in calls), calls
    fused_launches = sum(("k_setup_tail" in ln or "k_numeric_setup" in ln) for ln in r.stderr.splitlines() if ln.startswith("emu: "))
    assert fused_launches == tails_run, r.stderr[-1500:]
    if stall:
        assert all(c["fallbacks"] == 1 and c["fused_ok"] == 0 for c in calls), calls
    else:
        assert all(c["fallbacks"] == 0 and c["fused_ok"] == 1 for c in calls), calls


def test_fused_tails_with_four_rows_per_thread_and_a_65_way_grid_barrier(emu_lib):
    """Above 256 K rows the fused tails take four rows per thread (k_setup_tail<4>, k_numeric_setup<4>): M = 262,144 is
    65 workgroups of 1024.  72 emulated compute units (72 OS threads), the census skipped and the device clock slowed
    down so that the 50 ms bound of the barrier outlasts an oversubscribed host: both tails must have run, and C must
    be the oracle's (tests/test_spgemm_gpu.py::test_fused_tails_at_their_size_limits[262144])."""
    cov = os.path.join(emu_lib, "cov_fused4.txt")
    if os.path.exists(cov):
        os.remove(cov)
    n = _gpu_tests_on_emu(emu_lib, ["tests/test_spgemm_gpu.py", "-k", "fused_tails_at_their_size_limits and 262144"],
                          env={"NSPARSE_FUSED_FORCE": "1", "EMU_WORKERS": "72", "EMU_CLOCK_DIV": "20000", "EMU_COVERAGE": cov})
    assert n == 1
    launched = set(open(cov).read().split())
    os.remove(cov)
    assert any("k_setup_tailILi4" in k for k in launched) and any("k_numeric_setupILi4" in k for k in launched), launched
