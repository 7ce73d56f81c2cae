"""The four sample binaries with the reference's names and command lines
(cuda-c/Makefile:99-113; spgemm_hash.cu:79-94; spmv_amb.cu:75-118) on the reference's fixture:
they must print the reference's report lines and the 'Correct' verdict of its check functions."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN
import nsparse_amd as ns

pytestmark = pytest.mark.gpu
MTX = os.path.join(GOLDEN, "test.mtx")


def run(name, *args):
    exe = os.path.join(ns.capi.LIB_DIR, name)
    assert os.path.exists(exe), f"{exe} not built"
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("prec", ["d", "s"])
def test_spgemm_hash_cli(prec):
    out = run(f"spgemm_hash_{prec}", MTX)
    assert out.count("Read mtx file:") == 2
    assert re.search(r"SpGEMM using CSR format \(Hash-based\): .*test\.mtx, [\d.]+\[GFLOPS\], [\d.]+\[ms\]", out)
    assert "(nnz of A): 9 =>" in out and "(Num of intermediate products): 19 =>" in out
    assert "(nnz of C): 11" in out
    assert "Calculation Result is Correct" in out


@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("plan", [(), ("65536", "1"), ("3", "2")])
def test_amb_cli(prec, plan):
    out = run(f"amb_{prec}", MTX, *plan)
    m = re.search(r"Format Conversion Cost \(CSR=>AMB, (\d+)-(\d+)\): [\d.]+\[msec\]", out)
    assert m
    if plan:
        assert (m.group(1), m.group(2)) == plan
    assert re.search(r"SpMV using AMB format: .*test\.mtx, [\d.]+\[GFLOPS\], [\d.]+\[ms\]", out)
    assert "Calculation Result is Correct" in out


def test_amb_cli_keeps_the_plan_beside_the_matrix(tmp_path):
    """NSPARSE_BIN_CACHE=1: the first run writes <file>.plan (and <file>.csr.bin), the second run
    converts with that plan instead of searching; same layout, same verdict."""
    import shutil
    mtx = str(tmp_path / "test.mtx")
    shutil.copy(MTX, mtx)
    exe = os.path.join(ns.capi.LIB_DIR, "amb_d")
    env = dict(os.environ, NSPARSE_BIN_CACHE="1")
    runs = [subprocess.run([exe, mtx], capture_output=True, text=True, timeout=300, env=env) for _ in range(2)]
    assert all(r.returncode == 0 for r in runs), runs[-1].stderr[-2000:]
    assert os.path.exists(mtx + ".plan") and os.path.exists(mtx + ".csr.bin")
    assert "plan:" not in runs[0].stderr and "plan: " + mtx + ".plan" in runs[1].stderr
    conv = [re.search(r"CSR=>AMB, (\d+)-(\d+)\)", r.stdout).groups() for r in runs]
    assert conv[0] == conv[1]
    assert all("Calculation Result is Correct" in r.stdout for r in runs)
