"""bench.py end to end WITHOUT a device (NSPARSE_BENCH_DRYRUN=1, tools/bench_dry.py): the spawn of the ranks, the
rendezvous, the partition helpers, the host half of the library, the `configs` sub-processes, the PMC table parsing
and the assembly of the ONE JSON line all run; everything that needs the GPU is a stand-in with made-up times.  The
values in such a line mean nothing -- its keys, types and the driver's contract are what is asserted here, so that the
first run on the device cannot die in the harness (round-4 verdict, item 2)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = json.load(open(os.path.join(ROOT, "BASELINE.json")))


def _run(cmd, extra_env=None, timeout=600):
    env = dict(os.environ, NSPARSE_BENCH_DRYRUN="1", **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0]), r.stderr


def _contract(d, n, steps, warmup):
    assert d["metric"] == BASELINE["metric"]
    assert d["unit"] == "GFLOPS" and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] in ("synthetic", "file")
    assert isinstance(d["value"], float) and d["value"] > 0 and isinstance(d["ms_per_step"], float) and d["ms_per_step"] > 0
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    assert "dry_run" in d  # a dry line can never be mistaken for a measurement
    rt = d["runtime"]
    assert rt["torch_imported"] is False and any(k.startswith("libnsparse_d") for k in rt["mapped"])


def test_dry_run_world_1_full_line():
    d, err = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--spmv-steps", "3"])
    _contract(d, 1, 2, 1)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    for k in ("kernel", "achieved", "frac", "traffic", "bytes_per_launch", "ms_per_launch", "products_per_launch"):
        assert rf[k] is not None, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["measured_hbm"]["over_compulsory"] > 0 and "k_num_block<128, 1536" in rf["kernel"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert cpu["all_cores"]["cores"] >= 1 and cpu["spmv"]["value"] > 0
    cases = d["configs"]["cases"]
    assert [c["case"] for c in cases] == ["webbase1m", "stencil", "rmat22"]
    for c in cases:
        assert c["structure_check"]["rpt_equal"] is True and c["traffic"] > 0 and c["traffic_over_compulsory"] > 0
        assert c["roofline"]["measured_frac"] >= 0 and len(c["traffic_top_kernels"]) == 2
    assert cases[0]["dtype"] == "f32" and cases[0]["baseline_config"] == 3 and cases[2]["baseline_config"] == 5
    for k in ("spmv", "spmv_hbm"):
        s = d[k]
        assert s["ans_check_fails"] == 0 and s["value"] > 0 and s["unit"] == "GB/s" and 0 < s["frac_hbm_peak"]
        assert s["traffic"] > 0 and "hipgraph" in s and s["plan"]["block_size"] >= 1
    assert d["regular_brick"]["value"] > 0 and len(d["structure_sweep"]["points"]) == 4
    assert "error" in d["structure_sweep"]["twins_off"]  # the one leg a dry run cannot stand in for says so
    assert d["timing"]["reference_compatible_ms"] > 0 and d["timing"]["alloc_async_ms"] > 0
    assert d["wall_s"] < 180


def test_dry_run_configs_budget_spent_is_reported_not_fatal():
    d, _ = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--spmv-steps", "2", "--no-cpu",
                 "--no-irregular", "--no-large", "--configs-budget", "1"])
    assert all("skipped" in c for c in d["configs"]["cases"])
    assert d["cpu_baseline"] is None and d["spmv_hbm"] is None and d["regular_brick"] is None


def test_dry_run_world_2_spawned_ranks():
    d, err = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--spmv-steps", "3"])
    _contract(d, 2, 2, 1)
    assert "spawned 2 rank processes" in err
    assert d["config"]["parallelism"].startswith("row-partition x2")
    # per-N legs are off at N > 1; the two SpMV workloads ran row-sharded on both ranks
    assert d["cpu_baseline"] is None and d["configs"] is None and d["roofline"]["traffic"] is None
    assert d["spmv"]["ans_check_fails"] == 0 and d["spmv_hbm"]["scaling"] == "strong"
    assert d["spmv_hbm"]["M"] == 16 * 16 * 8


def test_dry_run_world_2_under_the_drivers_launcher():
    """The driver's N > 1 command line: python -m torch.distributed.run ... bench.py --gpus N.  Only the agent
    imports torch; the ranks find each other through the default rendezvous directory (MASTER_PORT + parent pid)."""
    pytest.importorskip("torch")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2",
                   "--warmup", "1", "--spmv-steps", "3", "--no-large"], timeout=900)
    _contract(d, 2, 2, 1)
    assert d["runtime"]["torch_imported"] is False
