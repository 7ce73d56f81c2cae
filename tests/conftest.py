import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_d():
    from oracle.oracle import Oracle
    return Oracle("d")


@pytest.fixture(scope="session")
def oracle_s():
    from oracle.oracle import Oracle
    return Oracle("s")


@pytest.fixture(scope="session")
def lib_d():
    import nsparse_amd
    return nsparse_amd.load("d")


@pytest.fixture(scope="session")
def lib_s():
    import nsparse_amd
    return nsparse_amd.load("s")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["M"], d["N"] = int(d["M"]), int(d["N"])
    d["nnz"] = int(d["rpt"][-1])
    return d


# known answers for the reference's only fixture, data/test.mtx
TEST_MTX = dict(
    M=5, N=5, nnz=9, nnz_max=3,
    rpt=[0, 2, 3, 6, 7, 9], col=[0, 2, 1, 0, 2, 4, 3, 2, 4],
    val=[10, 1, 20, 1, 30, 2, 40, 2, 50],
    x=[1, 2, 3, 4, 5], y=[13, 40, 101, 160, 256],
    row_prod=[5, 1, 7, 1, 5], n_prod=19,
    c_rpt=[0, 3, 4, 7, 8, 11], c_col=[0, 2, 4, 1, 0, 2, 4, 3, 0, 2, 4],
    c_val=[101, 40, 2, 400, 40, 905, 160, 1600, 2, 160, 2504],
)
