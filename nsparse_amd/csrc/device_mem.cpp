// device_mem.cpp -- device allocation cache, process context, H2D / D2H of sfCSR,
// device-side frees.
//
// Replaces (reference file:line):
//   csr_memcpy / csr_memcpyDtH     cuda-c/src/nsparse.cu:146-168
//   release_csr / release_amb      cuda-c/src/nsparse.cu:209-235
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "internal.h"

namespace nsp {

namespace {
constexpr int kMaxDevices = 128;  // an 8-GPU MI300X node in CPX mode exposes 64 devices
// One cache for the process, one idle list per device: a block goes back to the list of the device
// it was allocated on, whatever device is current when it is released.  Guarded by its own mutex
// (csr_memcpy / release_* may be called from any thread).
struct Cache {
    bool enabled = true;
    // nsparse_set_workspace_cache(2): cache off, blocks from the runtime's stream-ordered allocator
    // (hipMallocAsync / hipFreeAsync on the null stream, default pool told to keep what is freed) instead of
    // hipMalloc / hipFree -- still an allocation call per array inside every spgemm_kernel_hash, as in the
    // reference (spgemm_hash.cu:40-44), but served by the runtime's pool and without hipFree's device-wide wait
    bool async = false;
    bool pool_set[kMaxDevices] = {};
    struct Live { size_t bytes; int dev; bool async; };
    std::unordered_map<void *, Live> live;               // blocks handed out
    std::multimap<size_t, void *> idle[kMaxDevices];     // blocks waiting for reuse
    size_t idle_bytes = 0;
    // stream-ordered blocks released while a call is in flight (CallScope): their hipFreeAsync goes to the null
    // stream, which is not ordered against the call's non-blocking streams, so it waits for the end of the call
    std::vector<std::pair<void *, int>> deferred;
    std::mutex mu;
};
Cache &cache()
{
    static Cache c;
    return c;
}
// -1: the current device is beyond the library's per-device tables.  Never aliased to device 0 (its streams and
// scratch live there): dev_alloc refuses, ctx() cannot continue.
int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    if (d < 0 || d >= kMaxDevices) {
        set_error(-50, "device id beyond the library's per-device tables (128)", __FILE__, __LINE__);
        return -1;
    }
    return d;
}
thread_local int t_call_depth = 0;  // > 0: inside a public call that has kernels in flight (CallScope)

// hipFreeAsync on the null stream OF THE DEVICE THE BLOCK CAME FROM (the caller may have another one current)
void free_async_on(void *p, int dev)
{
    int cur = 0;
    const bool sw = hipGetDevice(&cur) == hipSuccess && cur != dev;
    if (sw) (void)hipSetDevice(dev);
    NSP_CHECK(hipFreeAsync(p, 0));
    if (sw) (void)hipSetDevice(cur);
}
void trim_locked(Cache &c)
{
    for (auto &lst : c.idle) {
        for (auto &kv : lst) (void)hipFree(kv.second);
        lst.clear();
    }
    c.idle_bytes = 0;
}
// Round so that near-equal requests of consecutive calls hit the same idle block.
inline size_t round_size(size_t b)
{
    if (b == 0) b = 1;
    const size_t g = b < (1u << 20) ? 256 : (b < (64u << 20) ? (64u << 10) : (2u << 20));
    return (b + g - 1) / g * g;
}
}  // namespace

void *dev_alloc(size_t bytes)
{
    Cache &c = cache();
    const size_t want = round_size(bytes);
    const int dev = current_device();
    if (dev < 0) return nullptr;  // (error -50 is set)
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.enabled) {
        auto &idle = c.idle[dev];
        auto it = idle.lower_bound(want);
        // accept an idle block up to 25 % (+1 MiB) larger than the request
        if (it != idle.end() && it->first <= want + want / 4 + (1u << 20)) {
            void *p = it->second;
            c.live[p] = {it->first, dev, false};
            c.idle_bytes -= it->first;
            idle.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    if (c.async) {
        if (!c.pool_set[dev]) {
            hipMemPool_t pool = nullptr;
            unsigned long long keep = ~0ull;
            if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess)
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            c.pool_set[dev] = true;
        }
        if (hipMallocAsync(&p, want, 0) == hipSuccess) {
            c.live[p] = {want, dev, true};
            return p;
        }
        (void)hipGetLastError();  // no stream-ordered allocator on this runtime: plain hipMalloc below
        p = nullptr;
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && c.enabled && c.idle_bytes > 0) {
        (void)hipGetLastError();
        trim_locked(c);  // give the idle blocks back and retry once
        e = hipMalloc(&p, want);
    }
    NSP_CHECK(e);
    if (c.enabled) c.live[p] = {want, dev, false};
    return p;
}

void dev_free(void *p)
{
    if (!p) return;
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            if (it->second.async) {
                const int dev = it->second.dev;
                c.live.erase(it);
                if (t_call_depth > 0) c.deferred.emplace_back(p, dev);  // kernels of this call may still use it
                else free_async_on(p, dev);
                return;
            }
            if (c.enabled) {
                c.idle[it->second.dev].emplace(it->second.bytes, p);
                c.idle_bytes += it->second.bytes;
                c.live.erase(it);
                return;
            }
            c.live.erase(it);
        }
    }
    NSP_CHECK(hipFree(p));
}

bool dev_cache_enabled()
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    return c.enabled;
}

void dev_cache_enable(bool on)
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!on) trim_locked(c);
    c.enabled = on;
    c.async = false;
}

void dev_cache_async(bool on)
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    c.async = on;
}

CallScope::CallScope() { ++t_call_depth; }
CallScope::~CallScope()
{
    if (--t_call_depth > 0) return;
    Cache &c = cache();
    std::vector<std::pair<void *, int>> todo;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        todo.swap(c.deferred);
    }
    if (todo.empty()) return;
    // the public calls are synchronous on return, so the device is idle here; the wait costs nothing then and
    // makes the order explicit for a caller-supplied future path that is not
    (void)hipDeviceSynchronize();
    for (auto &pd : todo) free_async_on(pd.first, pd.second);
}

void dev_cache_trim()
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    trim_locked(c);
}

std::recursive_mutex &api_mutex()
{
    static std::recursive_mutex m;
    return m;
}

void wait_published(int slot, int seq, hipStream_t st)
{
    Context &c = ctx();
    volatile int *p = c.h_mapped + slot;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(p, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xffff) == 0 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            NSP_CHECK(hipStreamSynchronize(st));  // surfaces a device error, if any
            if (__atomic_load_n(p, __ATOMIC_ACQUIRE) != seq) set_error(-30, "publish flag never arrived", __FILE__, __LINE__);
            return;
        }
    }
}

// One context per device, created on first use with that device current: streams, events and the
// counter / flag blocks belong to the device the caller selected with hipSetDevice (the reference
// builds its sfBIN and streams inside every call, so it follows the current device as well).
namespace {
Context g_per_dev[kMaxDevices];
}
static bool ctx_peek(int d) { return g_per_dev[d].ready; }

Context &ctx()
{
    Context *per_dev = g_per_dev;
    static std::mutex mu;
    const int dev = current_device();
    if (dev < 0) {
        // nothing sensible can run without streams and counters of ITS device: this is fatal whatever
        // NSPARSE_NO_ABORT says (the error word and a message were set by current_device)
        fprintf(stderr, "nsparse: device beyond the per-device tables (%d): cannot continue\n", kMaxDevices);
        abort();
    }
    Context &c = per_dev[dev];
    std::lock_guard<std::mutex> lk(mu);
    if (!c.ready) {
        // the streams of the big-LDS bins (hash bin 4, heavy bin 5, bit-window bin 10) get the highest
        // priority: their few, large workgroups should take CUs as they free up instead of queueing behind a
        // million small rows and running alone at the end (NSPARSE_STREAM_PRIO=0: all equal)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        const bool prio_on = exp_env("NSPARSE_STREAM_PRIO", 1) != 0;
        for (int i = 0; i < kMaxBins; i++) {
            // NSPARSE_PRIO_BINS=<bit mask of bins>: which bins' streams get it (default: 4, 5, 10)
            static const unsigned prio_mask = (unsigned)exp_env("NSPARSE_PRIO_BINS", 0x430);
            const bool big = prio_on && ((prio_mask >> i) & 1u);
            if (big) {
                NSP_CHECK(hipStreamCreateWithPriority(&c.stream[i], hipStreamNonBlocking, prio_hi));
                NSP_CHECK(hipEventCreateWithFlags(&c.ev_join[i], hipEventDisableTiming));
                continue;
            }
            NSP_CHECK(hipStreamCreateWithFlags(&c.stream[i], hipStreamNonBlocking));
            NSP_CHECK(hipEventCreateWithFlags(&c.ev_join[i], hipEventDisableTiming));
        }
        NSP_CHECK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
        for (auto &e : c.ev_t) NSP_CHECK(hipEventCreate(&e));
        for (auto &e : c.ev_bin) NSP_CHECK(hipEventCreate(&e));
        NSP_CHECK(hipHostMalloc((void **)&c.h_pinned, 512 * sizeof(int), hipHostMallocDefault));
        NSP_CHECK(hipMalloc((void **)&c.d_scratch, 8192 * sizeof(int)));
        NSP_CHECK(hipHostMalloc((void **)&c.h_mapped, 256 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
        memset(c.h_mapped, 0, 256 * sizeof(int));
        NSP_CHECK(hipHostGetDevicePointer((void **)&c.d_mapped, c.h_mapped, 0));
        c.device = dev;
        NSP_CHECK(hipDeviceGetAttribute(&c.num_cus, hipDeviceAttributeMultiprocessorCount, dev));
        c.ready = true;
    }
    return c;
}

bool ctx_ready()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return false;
    // (reads a flag of the table ctx() owns: a context that is being created right now counts as "not yet")
    return ctx_peek(d);
}

// ---- roctx ranges (tracing) ------------------------------------------------------------------
// rocprofv3 --marker-trace shows them around the phases of a call, so a kernel trace explains itself.  The
// marker library is looked up at run time (librocprofiler-sdk-roctx, else the older libroctx64): the product
// library does not link it, and without a profiler in the process (NSPARSE_ROCTX=1 forces, =0 forbids) nothing
// is loaded and a range is two predictable branches.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("NSPARSE_ROCTX");
        if (e && atoi(e) == 0) return;
        const char *pre = getenv("LD_PRELOAD");
        const bool profiled = getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") ||
                              (pre && strstr(pre, "rocprofiler")) || (e && atoi(e) == 1);
        if (!profiled) return;
        void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx &roctx()
{
    static Roctx r;
    return r;
}
}  // namespace
TraceRange::TraceRange(const char *name) : on(roctx().push != nullptr)
{
    if (on) roctx().push(name);
}
TraceRange::~TraceRange()
{
    if (on) roctx().pop();
}
void TraceRange::next(const char *name)
{
    if (!on) return;
    roctx().pop();
    roctx().push(name);
}
bool trace_ranges_on() { return roctx().push != nullptr; }

}  // namespace nsp

extern "C" {

int nsparse_trace_ranges(void) { return nsp::trace_ranges_on() ? 1 : 0; }

void nsparse_set_workspace_cache(int on)
{
    nsp::dev_cache_enable(on == 1);
    if (on == 2) nsp::dev_cache_async(true);
}
void nsparse_trim_workspace(void) { nsp::dev_cache_trim(); }
void nsparse_set_profiling(int on)
{
    nsp::ApiLock lk;
    nsp::ctx().profiling = (on != 0);
}
int nsparse_set_bin_timing(int on)
{
    nsp::ApiLock lk;
    const int old = nsp::ctx().bin_timing;
    nsp::ctx().bin_timing = (on != 0);
    return old;
}

void csr_memcpy(sfCSR *mat)
{
    nsp::clear_error();
    mat->d_rpt = (int *)nsp::dev_alloc(sizeof(int) * (size_t)(mat->M + 1));
    mat->d_col = (int *)nsp::dev_alloc(sizeof(int) * (size_t)mat->nnz);
    mat->d_val = (real *)nsp::dev_alloc(sizeof(real) * (size_t)mat->nnz);
    NSP_CHECK(hipMemcpy(mat->d_rpt, mat->rpt, sizeof(int) * (size_t)(mat->M + 1), hipMemcpyHostToDevice));
    NSP_CHECK(hipMemcpy(mat->d_col, mat->col, sizeof(int) * (size_t)mat->nnz, hipMemcpyHostToDevice));
    NSP_CHECK(hipMemcpy(mat->d_val, mat->val, sizeof(real) * (size_t)mat->nnz, hipMemcpyHostToDevice));
}

void csr_memcpyDtH(sfCSR *mat)
{
    nsp::clear_error();
    // allocates the host arrays, caller frees with release_cpu_csr (as upstream)
    mat->rpt = (int *)malloc(sizeof(int) * (size_t)(mat->M + 1));
    mat->col = (int *)malloc(sizeof(int) * (size_t)(mat->nnz > 0 ? mat->nnz : 1));
    mat->val = (real *)malloc(sizeof(real) * (size_t)(mat->nnz > 0 ? mat->nnz : 1));
    NSP_CHECK(hipMemcpy(mat->rpt, mat->d_rpt, sizeof(int) * (size_t)(mat->M + 1), hipMemcpyDeviceToHost));
    if (mat->nnz > 0) {
        NSP_CHECK(hipMemcpy(mat->col, mat->d_col, sizeof(int) * (size_t)mat->nnz, hipMemcpyDeviceToHost));
        NSP_CHECK(hipMemcpy(mat->val, mat->d_val, sizeof(real) * (size_t)mat->nnz, hipMemcpyDeviceToHost));
    }
}

void release_csr(sfCSR mat)
{
    nsp::dev_free(mat.d_rpt);
    nsp::dev_free(mat.d_col);
    nsp::dev_free(mat.d_val);
}

void release_amb(sfAMB mat)
{
    nsp::dev_free(mat.d_cs);
    nsp::dev_free(mat.d_cl);
    nsp::dev_free(mat.d_sellcs_val);
    nsp::dev_free(mat.d_sellcs_col);
    nsp::dev_free(mat.d_write_permutation);
    nsp::dev_free(mat.d_s_write_permutation);
    nsp::dev_free(mat.d_s_write_permutation_offset);
}

}  // extern "C"
