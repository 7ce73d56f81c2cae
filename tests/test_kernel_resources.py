"""Static resource gate (no GPU): every kernel of ours, in every HIP translation unit and both precisions, compiled
for gfx950 with the compiler's resource remarks.  None may use scratch (a spill in an HBM-bound kernel writes and
re-reads its own operands; round 5's review found six default SpMV instantiations doing exactly that), none may ask for
more LDS than a CDNA4 compute unit has (160 KB; static arrays above 64 KB are legal on gfx950 -- the heavy-row kernels
use up to 156 KB and have run on the device since round 3).  The table itself is kept under profiles/
(tools/kernel_resources.py)."""
import os
import shutil
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import kernel_resources as kr  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc on this box")

LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def tables():
    jobs = [(p, u) for p in ("d", "s") for u in kr.UNITS]
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda j: kr.unit_resources(*j), jobs))
    return dict(zip(jobs, res))


def test_every_unit_has_kernels_of_ours(tables):
    for (p, u), rows in tables.items():
        assert any(r["own"] for r in rows), f"{u} ({p}): the remark parser found no kernel of ours"


def test_no_kernel_of_ours_uses_scratch(tables):
    bad = [(p, u, r["name"], r["scratch"], r["vgpr"] + r["agpr"]) for (p, u), rows in tables.items() for r in rows
           if r["own"] and (r["scratch"] > 0 or r["vspill"] > 0)]
    assert not bad, "kernels with scratch / spilled VGPRs:\n" + "\n".join(
        f"  {p} {u}: {n}  scratch {s} B/lane at {v} VGPRs" for p, u, n, s, v in bad)


def test_lds_fits_a_compute_unit(tables):
    bad = [(p, u, r["name"], r["lds"]) for (p, u), rows in tables.items() for r in rows if r["lds"] > LDS_PER_CU]
    assert not bad, f"LDS above the {LDS_PER_CU} B of a compute unit: {bad}"


def test_every_kernel_of_ours_can_be_resident(tables):
    # Occupancy 0 would mean the register allocation cannot host one wavefront of the launch bound
    bad = [(p, u, r["name"]) for (p, u), rows in tables.items() for r in rows if r["own"] and r["occ"] < 1]
    assert not bad, bad


# ---- the -DNSPARSE_EXPERIMENTS variant (nsparse_amd/lib_exp): the opt-in kernel families live only there ----------------
LEGACY_MEASUREMENT_FORMS = ("k_spmv_amb_pipe<",)  # rounds 1-2's SpMV forms, kept for before / after counters (tools/pmc_spmv.sh)


@pytest.fixture(scope="module")
def variant_tables():
    jobs = [(p, u) for p in ("d", "s") for u in ("spgemm_hash", "spmv_amb")]
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda j: kr.unit_resources(j[0], j[1], extra=("-DNSPARSE_EXPERIMENTS",)), jobs))
    return dict(zip(jobs, res))


def test_opt_in_kernel_families_exist_only_in_the_variant(tables, variant_tables):
    """One hash family in the product (k_sym_tb / k_num_tb); the lean family in all four forms, the stateless heavy-row tiles
    and the split-row SpMV are template instantiations of the variant library only."""
    opt_in = ("k_sym_lean<", "k_num_lean<", "k_num_flat<", "k_num_ranked_flat<", "k_panel_", "k_spmv_amb_split<")
    prod = [r["name"] for (p, u), rows in tables.items() for r in rows if r["own"]]
    var = [r["name"] for (p, u), rows in variant_tables.items() for r in rows if r["own"]]
    assert not [n for n in prod if any(k in n for k in opt_in)], "an opt-in kernel family leaked into the product library"
    for k in opt_in:
        assert any(k in n for n in var), f"{k} missing from the variant library"
    forms = {n.split(",")[-1].strip(" >") for n in var if "k_num_lean<" in n}
    assert forms == {"0", "1", "2", "3"}, forms  # retry blocks / branch-free x grouped / pipelined: one build for the A/B
    n_spgemm = len({r["name"] for (p, u), rows in tables.items() if (p, u) == ("d", "spgemm_hash") for r in rows if r["own"]})
    assert n_spgemm <= 80, f"{n_spgemm} kernels of ours in the product's SpGEMM unit"


def test_no_opt_in_kernel_uses_scratch(variant_tables):
    bad = [(p, u, r["name"], r["scratch"]) for (p, u), rows in variant_tables.items() for r in rows
           if r["own"] and (r["scratch"] > 0 or r["vspill"] > 0) and not any(k in r["name"] for k in LEGACY_MEASUREMENT_FORMS)]
    assert not bad, bad
    assert all(r["lds"] <= LDS_PER_CU for rows in variant_tables.values() for r in rows)
