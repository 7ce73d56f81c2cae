#!/bin/bash
# The whole -m gpu corpus through the CPU emulation (tests/emu), minus the tests that are ABOUT the hardware or the vendor
# library; summary -> profiles/<tag>_emu_corpus.txt.  ~50 min on 8 cores.   bash tools/emu_corpus.sh [tag] [EMU_ORDER]
#   deselected, and why:
#     test_vendor_gpu.py                                     rocSPARSE needs the device
#     test_asan_build_runs_clean                             loads the ASan build of the HIP library
#     test_fused_tails_under_a_cu_mask                       HSA_CU_MASK is a property of the real runtime
#     test_single_rank_line_is_torch_free_and_on_the_system_runtime   asserts libamdhip64 is mapped
#     test_cant_class_file_through_{loader_and_spgemm,amb}_sample     assert GFLOPS / GB/s floors of the device
#     test_roctx_ranges_reach_a_marker_trace                 rocprofv3 needs a device (the test skips when the profiler is absent, fails when it cannot trace)
#     test_library_on_the_gpu_box_was_built_from_these_sources   the emulation build carries no source hash
#     test_config5_rmat22                                    needs > 64 GB of host memory here (C alone is 24 GB, twice)
cd "$(dirname "$0")/.."
TAG=${1:-r05}; ORDER=${2:-0}
OUT=profiles/${TAG}_emu_corpus$([ "$ORDER" != 0 ] && echo _order$ORDER).txt
make -C tests/emu -j8 -s || exit 1
LOG=$(mktemp /tmp/emu_corpus.XXXXXX)
EMU_CLOCK_DIV=2000 EMU_ORDER=$ORDER EMU_WATCHDOG_S=3000 NSPARSE_LIB_DIR=$PWD/tests/emu/lib timeout 14000 python -u -m pytest tests -m gpu -q -p no:cacheprovider --timeout 3600 -rf \
  --ignore=tests/test_vendor_gpu.py \
  --deselect tests/test_aux_gpu.py::test_asan_build_runs_clean \
  --deselect tests/test_spgemm_gpu.py::test_fused_tails_under_a_cu_mask \
  --deselect tests/test_bench_gpu.py::test_single_rank_line_is_torch_free_and_on_the_system_runtime \
  --deselect tests/test_samples_gpu.py::test_cant_class_file_through_loader_and_spgemm_sample \
  --deselect tests/test_samples_gpu.py::test_cant_class_file_through_amb_sample \
  --deselect tests/test_configs_gpu.py::test_config5_rmat22 \
  --deselect tests/test_host_abi.py::test_library_on_the_gpu_box_was_built_from_these_sources \
  --durations=25 > $LOG 2>&1
{
  echo "# -m gpu corpus on the CPU emulation (tests/emu), EMU_ORDER=$ORDER, tree $(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty), $(date -u +%FT%TZ)"
  echo "# deselected: see tools/emu_corpus.sh (hardware- / vendor-specific assertions)"
  grep -E "^(FAILED|ERROR)|passed|failed" $LOG | grep -v "^Read mtx"
  echo "# slowest"
  grep -E "^[0-9.]+s (call|setup)" $LOG | head -25
} > $OUT
tail -5 $OUT
