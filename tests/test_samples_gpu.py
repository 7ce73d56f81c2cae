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
