"""Reads an AddressSanitizer report (test infrastructure for tests/test_aux_gpu.py::test_asan_build_runs_clean)."""
import os
import re

ROCM_RUNTIME_MODULES = ("libamdhip64", "libhsa-runtime64", "libamd_comgr", "librocprofiler", "libhsakmt", "libdrm")


def first_module(report):
    """Module (file name) of the first frame of the FIRST stack of an ASan report that is neither the ASan runtime's
    interceptor nor libc; None when the stack has no readable module (unsymbolized / truncated)."""
    in_stack = False
    for ln in report.splitlines():
        m = re.match(r"\s*#(\d+) 0x[0-9a-f]+ .*\(([^()]+?)\+0x[0-9a-f]+\)", ln)
        if not m:
            if in_stack:
                break
            continue
        in_stack = True
        mod = os.path.basename(m.group(2))
        if mod.startswith(("libclang_rt.asan", "libasan", "libc.so", "libc-", "libstdc++", "ld-linux")):
            continue
        return mod
    return None
