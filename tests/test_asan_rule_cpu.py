"""The rule by which test_asan_build_runs_clean tells a report of the ROCm runtime from one of this library (no GPU)."""
from asan_report import ROCM_RUNTIME_MODULES, first_module


def test_asan_skip_rule_only_matches_the_rocm_runtime():
    own = ("==1==ERROR: AddressSanitizer: heap-buffer-overflow\nWRITE of size 8\n"
           "    #0 0x7f0 in __asan_memcpy (/opt/rocm/lib/llvm/lib/clang/20/lib/linux/libclang_rt.asan-x86_64.so+0x1)\n"
           "    #1 0x7f1 in nsp::load (/root/repo/nsparse_amd/lib_asan/libnsparse_d.so+0x2)\n"
           "    #2 0x7f2 in hipMemcpy (/opt/rocm/lib/libamdhip64.so.7+0x3)\n")
    assert first_module(own) == "libnsparse_d.so"
    rt = own.replace("/root/repo/nsparse_amd/lib_asan/libnsparse_d.so", "/opt/rocm/lib/libhsa-runtime64.so.1")
    assert first_module(rt).startswith(ROCM_RUNTIME_MODULES)
    assert first_module("==1==ERROR: AddressSanitizer: SEGV\n    #0 0x7f0  (<unknown module>)\n") is None
