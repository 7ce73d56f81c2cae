// internal.h -- shared plumbing of libnsparse_{d,s}.so (not installed, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <cstdio>
#include <cstdlib>

#include "nsparse.h"

namespace nsp {

// ---- switches ------------------------------------------------------------------------------------------------------
// The PRODUCT library reads nine behaviour switches from the environment (NSPARSE_DENSE, NSPARSE_TWINS, NSPARSE_FUSED,
// NSPARSE_LIST, NSPARSE_AMB_TUNE, NSPARSE_BIN_CACHE, NSPARSE_DIST_TIMEOUT_S, NSPARSE_NO_ABORT, NSPARSE_ROCTX) and nothing
// else.  Everything that exists to MEASURE -- kernel-form selectors, ablations, the opt-in kernel families that have not
// been timed on the device yet -- goes through exp_env: read from the environment only in a -DNSPARSE_EXPERIMENTS build
// (nsparse_amd/lib_exp), a constant in the product, so that the branches and kernel instantiations behind it are not
// part of the product at all (`if constexpr (kExpBuild)` where a kernel template would otherwise be instantiated).
#ifdef NSPARSE_EXPERIMENTS
constexpr bool kExpBuild = true;
#else
constexpr bool kExpBuild = false;
#endif
static inline int exp_env(const char *name, int dflt)
{
#ifdef NSPARSE_EXPERIMENTS
    const char *e = getenv(name);
    return e ? (int)strtol(e, nullptr, 0) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// ---- error channel -----------------------------------------------------------
// The reference aborts through checkCudaErrors on any failure.  We do the
// same by default, but record the code first so that NSPARSE_NO_ABORT=1 callers can
// poll nsparse_last_error().
void set_error(int code, const char *what, const char *file, int line);
void clear_error();

#define NSP_CHECK(expr)                                                          \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) ::nsp::set_error((int)_e, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define NSP_LAUNCH_CHECK() NSP_CHECK(hipGetLastError())

// ---- device block cache ---------------------------------------------------------
// hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises the device; the
// reference performs >= 8 of them inside every timed spgemm_kernel_hash call.
// All device memory handed out by this library goes through this cache so that a steady
// state loop (release_csr(c); spgemm_kernel_hash(a,b,&c);) touches the driver zero times.
void *dev_alloc(size_t bytes);
void dev_free(void *p);
void dev_cache_enable(bool on);
void dev_cache_async(bool on);
bool dev_cache_enabled();
void dev_cache_trim();
// RAII around a public call that queues kernels on the context's (non-blocking) streams: stream-ordered blocks
// (nsparse_set_workspace_cache(2)) released inside it are freed when the call has ended, not under its kernels.
struct CallScope {
    CallScope();
    ~CallScope();
};

// ---- tracing: roctx ranges around the phases of a call (no-ops unless a profiler is in the process) ----
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name);
    ~TraceRange();
    void next(const char *name);  // close the current range, open the next one
};
bool trace_ranges_on();

// ---- threading ---------------------------------------------------------------------
// The reference is not thread-safe (global `memory_access`, default stream).  Here every
// public entry point that touches a Context takes one process-wide lock, so concurrent callers
// serialise instead of sharing the device-side counters of a call in flight; the block cache has
// its own lock.  nsparse_spmv_amb_async touches no shared state and takes no lock.
std::recursive_mutex &api_mutex();
struct ApiLock {
    std::lock_guard<std::recursive_mutex> lk;
    ApiLock() : lk(api_mutex()) {}
};

// ---- per-device context ------------------------------------------------------------
constexpr int kMaxBins = 12;

struct Context {
    hipStream_t stream[kMaxBins] = {};  // one per row bin (the reference uses 7)
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[kMaxBins] = {};
    hipEvent_t ev_t[8] = {};            // phase timing
    hipEvent_t ev_bin[4 * kMaxBins] = {};  // per-bin begin/end: [0,2B) symbolic, [2B,4B) numeric
    int *h_pinned = nullptr;            // 512 ints of pinned host memory for small D2H
    int *d_scratch = nullptr;           // 8192 ints of device scratch (counters; 512.. workgroup records of the fused tails)
    int *h_mapped = nullptr;            // 256 ints of mapped, coherent host memory (GPU writes, host polls)
    int *d_mapped = nullptr;            // device address of h_mapped
    int seq = 0;                        // publish sequence number
    int device = -1;                    // the device this context belongs to
    int num_cus = 0;                    // compute units of the device
    int coresident = -1;                // 1024-thread workgroups resident together as this process sees the device
                                        // (census on first use: CU masks, partitions); grid barriers need every one resident
    bool fused_ok = true;               // false once a grid barrier has timed out: kernel chains from then on
    bool counters_clean = false;        // the SpGEMM counter blocks of d_scratch were zeroed behind the last call
    bool profiling = false;
    bool bin_timing = false;            // per-bin begin / end events in spgemm_kernel_hash (two API calls per bin)
    bool ready = false;
};
Context &ctx();
bool ctx_ready();  // the current device has a context already (a query must not create one)

// Block until the GPU has stored `seq` at h_mapped[slot] (see k_publish); falls back to a stream
// synchronisation after a generous timeout.
void wait_published(int slot, int seq, hipStream_t st);

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace nsp
