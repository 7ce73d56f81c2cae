// dist_spmv.hip -- libnsparse_dist_{d,s}.so: row-sharded AMB SpMV over RCCL (include/nsparse_dist.h).
//
// New design (the reference is single-GPU); the algorithm is SURVEY 8e / BASELINE.json north_star:
// 1-D row blocks cut by non-zeros, x replicated, one ncclAllGather of y per SpMV.  The SpMV kernel is the
// product library's (nsparse_spmv_amb_async = the launch path of sf_spmv_amb, reference
// kernel_spmv_amb.cu:10-104); this file adds the partition, the communicator and the per-iteration sequence
//     [memset y_local] -> k_spmv_amb_row -> ncclAllGather (in place) -> [k_close_gaps]
// on ONE stream with no host synchronisation, optionally replayed from a hipGraph.
//
// xGMI is point to point (7 links per GPU) and a shard of y is small (3.5 MB per rank for the nlpkkt120
// class at 8 ranks): one collective with the whole shard as the message, nothing to bucket; the collective
// needs the finished shard, so there is nothing to overlap inside one SpMV either.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "nsparse_dist.h"

namespace {

thread_local int g_err = 0;

inline int fail_hip(hipError_t e, const char *what, int line)
{
    g_err = (int)e;
    fprintf(stderr, "nsparse_dist: HIP error %d (%s) at %s:%d\n", (int)e, hipGetErrorString(e), what, line);
    return g_err;
}
inline int fail_nccl(ncclResult_t r, const char *what, int line)
{
    g_err = 2000 + (int)r;
    fprintf(stderr, "nsparse_dist: RCCL error %d (%s) at %s:%d\n", (int)r, ncclGetErrorString(r), what, line);
    return g_err;
}
#define D_HIP(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) return fail_hip(e_, #x, __LINE__);    \
    } while (0)
#define D_NCCL(x)                                                   \
    do {                                                            \
        ncclResult_t r_ = (x);                                      \
        if (r_ != ncclSuccess) return fail_nccl(r_, #x, __LINE__);  \
    } while (0)

constexpr ncclDataType_t kNcclReal = sizeof(real) == 8 ? ncclDouble : ncclFloat;

// Watchdog (round 4): nothing that waits on a peer waits for ever.  A collective whose peer never arrives leaves a
// kernel spinning on the stream; every wait below polls the stream and gives up after this many seconds
// (NSPARSE_DIST_TIMEOUT_S, nsparse_dist_set_timeout), aborts the communicator and returns -7.
double g_timeout_s = [] {
    const char *e = getenv("NSPARSE_DIST_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 60.0;
}();
constexpr int kReduceMax = 64;  // doubles per nsparse_dist_allreduce_f64 call

// Unequal blocks are gathered with equal shares of `rpr` elements (share r holds rows of block r at its
// start); this closes the gaps: y[cuts[r] + i] = staged[r * rpr + i].  One launch, 16 B per row moved.
__global__ __launch_bounds__(256) void k_close_gaps(real *__restrict__ y, const real *__restrict__ staged,
                                                    const int *__restrict__ cuts, int world, int rpr, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    // the block of row i: world <= 64, a short scan of a cached array
    int r = 0;
    while (r + 1 < world && cuts[r + 1] <= i) r++;
    y[i] = staged[(long long)r * rpr + (i - cuts[r])];
}

}  // namespace

struct nsparse_dist {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // matrix
    sfAMB amb;
    sfPlan plan;
    bool have_amb = false;
    std::vector<int> cuts;
    int *d_cuts = nullptr;
    int M = 0, rpr = 0, m_local = 0;
    bool equal_blocks = true;   // cuts[r] == r * rpr for every non-empty block: the gather lands in place
    real *staged = nullptr;     // world * rpr elements, unequal blocks only
    // graph
    hipGraphExec_t gexec = nullptr;
    hipGraph_t graph = nullptr;
    real *g_y = nullptr;
    const real *g_x = nullptr;
    int g_gather = -1;
    // barrier / small reductions among the ranks (device scratch: kReduceMax doubles + one int)
    double *d_red = nullptr;
    int *d_tick = nullptr;
    double *h_red = nullptr;  // pinned twin of d_red (+ one int): copies to and from it are truly asynchronous, so
                              // the host reaches the watchdog instead of blocking inside a pageable-memory copy
};

namespace {

int enqueue(nsparse_dist *h, real *d_y, const real *d_x, int gather)
{
    hipStream_t st = h->stream;
    const bool direct = h->equal_blocks || !gather || h->world == 1;
    real *y_loc = direct ? d_y + h->cuts[h->rank] : h->staged + (long long)h->rank * h->rpr;
    if (h->m_local > 0)
        nsparse_spmv_amb_async(y_loc, &h->amb, const_cast<real *>(d_x), &h->plan, (void *)st);
    if (!gather || (h->world == 1 && !h->comm)) return 0;
    if (!h->comm) return g_err = -4;  // handle made without an id: local rows only
    real *buf = direct ? d_y : h->staged;
    D_NCCL(ncclAllGather(buf + (long long)h->rank * h->rpr, buf, (size_t)h->rpr, kNcclReal, h->comm, st));
    if (!direct) return nsparse_dist_close_gaps(d_y, h->staged, h->d_cuts, h->world, h->rpr, h->M, (void *)st);
    return 0;
}

template <typename T>
int partition(const T *prefix, int M, int world, int align, int *cuts)
{
    // prefix[i] = work before row i, prefix[M] = total
    if (!prefix || !cuts || M < 0 || world < 1 || align < 1) return -1;
    const double total = (double)prefix[M];
    cuts[0] = 0;
    for (int r = 1; r < world; r++) {
        const double target = total * (double)r / (double)world;  // == python's total * r / world below 2^53
        int lo = 0, hi = M + 1;  // first i with prefix[i] >= target
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if ((double)prefix[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        long long row = (long long)std::nearbyint((double)lo / (double)align) * align;  // half to even, like round()
        if (row < cuts[r - 1]) row = cuts[r - 1];
        if (row > M) row = M;
        cuts[r] = (int)row;
    }
    cuts[world] = M;
    return 0;
}

// Wait for everything queued on the handle's stream, but not for ever: a poll of the stream (a few microseconds per
// query, so the timed loops lose nothing against hipStreamSynchronize) that also looks at the communicator's
// asynchronous error state.  On a time-out the communicator is aborted -- which ends the collective kernel that is
// waiting for its peer -- and the handle keeps working for local rows only.
int sync_watch(nsparse_dist *h)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    long polls = 0;
    bool slow = false;  // past 5 ms: one poll per 100 us instead of spinning
    for (;;) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail_hip(q, "hipStreamQuery", __LINE__);
        if (slow) std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (!slow && (++polls & 63) != 0) continue;
        const double el = std::chrono::duration<double>(clk::now() - t0).count();
        slow = el > 0.005;
        if (h->comm && slow) {
            ncclResult_t ar = ncclSuccess;
            if (ncclCommGetAsyncError(h->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                (void)ncclCommAbort(h->comm);
                h->comm = nullptr;
                return fail_nccl(ar, "asynchronous communicator error", __LINE__);
            }
        }
        if (el > g_timeout_s) {
            fprintf(stderr, "nsparse_dist: rank %d of %d waited %.0f s for its stream (a peer that never joined the "
                            "collective?): aborting the communicator\n", h->rank, h->world, el);
            if (h->comm) {
                (void)ncclCommAbort(h->comm);
                h->comm = nullptr;
            }
            return g_err = -7;
        }
    }
}

void free_handle(nsparse_dist *h)
{
    if (h->ev[0]) (void)hipEventDestroy(h->ev[0]);
    if (h->ev[1]) (void)hipEventDestroy(h->ev[1]);
    if (h->d_red) (void)hipFree(h->d_red);
    if (h->h_red) (void)hipHostFree(h->h_red);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// (takes the communicator over: on a failure it is destroyed with the half-built handle)
int new_handle(nsparse_dist **out, ncclComm_t comm, int rank, int world, int device)
{
    nsparse_dist *h = new nsparse_dist();
    h->rank = rank;
    h->world = world;
    h->device = device;
    h->comm = comm;
    memset(&h->amb, 0, sizeof(h->amb));
    memset(&h->plan, 0, sizeof(h->plan));
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev[0]);
    if (e == hipSuccess) e = hipEventCreate(&h->ev[1]);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_red, sizeof(double) * kReduceMax + sizeof(int) * 2);
    if (e == hipSuccess) e = hipMemset(h->d_red, 0, sizeof(double) * kReduceMax + sizeof(int) * 2);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_red, sizeof(double) * (kReduceMax + 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        if (comm) (void)ncclCommDestroy(comm);
        free_handle(h);
        return fail_hip(e, "new_handle", __LINE__);
    }
    h->d_tick = reinterpret_cast<int *>(h->d_red + kReduceMax);
    *out = h;
    return 0;
}

}  // namespace

extern "C" {

int nsparse_dist_last_error(void) { return g_err; }

int nsparse_dist_partition_nnz(const int *rpt, int M, int world, int align, int *cuts)
{
    return partition(rpt, M, world, align, cuts);
}

int nsparse_dist_partition_work(const long long *work_per_row, int M, int world, int align, int *cuts)
{
    if (!work_per_row || M < 0) return -1;
    std::vector<long long> prefix((size_t)M + 1);
    prefix[0] = 0;
    for (int i = 0; i < M; i++) prefix[i + 1] = prefix[i] + work_per_row[i];
    return partition(prefix.data(), M, world, align, cuts);
}

int nsparse_dist_csr_row_block(const sfCSR *full, int begin, int end, sfCSR *block)
{
    if (!full || !block || begin < 0 || end < begin || end > full->M) return -1;
    const int lo = full->rpt[begin], hi = full->rpt[end];
    const int m = end - begin, nnz = hi - lo;
    memset(block, 0, sizeof(*block));
    block->rpt = (int *)malloc(sizeof(int) * ((size_t)m + 1));
    block->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    block->val = (real *)malloc(sizeof(real) * (size_t)(nnz > 0 ? nnz : 1));
    if (!block->rpt || !block->col || !block->val) return -2;
    int longest = 0;
    for (int i = 0; i <= m; i++) block->rpt[i] = full->rpt[begin + i] - lo;
    for (int i = 0; i < m; i++) longest = block->rpt[i + 1] - block->rpt[i] > longest ? block->rpt[i + 1] - block->rpt[i] : longest;
    if (nnz > 0) {
        memcpy(block->col, full->col + lo, sizeof(int) * (size_t)nnz);
        memcpy(block->val, full->val + lo, sizeof(real) * (size_t)nnz);
    }
    block->M = m;
    block->N = full->N;
    block->nnz = nnz;
    block->nnz_max = longest;
    block->matrix_name = full->matrix_name;
    return 0;
}

int nsparse_dist_unique_id(char id[NSPARSE_DIST_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == NSPARSE_DIST_ID_BYTES, "id size");
    ncclUniqueId u;
    D_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

int nsparse_dist_init(nsparse_dist_t *h, const char id[NSPARSE_DIST_ID_BYTES], int rank, int world)
{
    g_err = 0;
    if (!h || world < 1 || rank < 0 || rank >= world) return g_err = -1;
    int device = 0;
    D_HIP(hipGetDevice(&device));
    ncclComm_t comm = nullptr;
    if (id) {
        // ncclCommInitRank returns when ALL ranks have called it: a rank that never starts (fewer GPUs than ranks,
        // a crashed peer) would hold the others for ever, so the call runs on a helper thread and is given
        // g_timeout_s; a thread that does not come back is left behind (the process is about to report an error)
        struct Job {
            std::mutex m;
            std::condition_variable cv;
            bool done = false;
            ncclResult_t res = ncclSuccess;
            ncclComm_t comm = nullptr;
        };
        auto job = std::make_shared<Job>();
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        std::thread([job, u, rank, world, device]() {
            ncclComm_t c = nullptr;
            ncclResult_t r = hipSetDevice(device) == hipSuccess ? ncclCommInitRank(&c, world, u, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> lk(job->m);
            job->res = r;
            job->comm = c;
            job->done = true;
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->m);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(g_timeout_s), [&] { return job->done; })) {
            fprintf(stderr, "nsparse_dist: rank %d of %d: ncclCommInitRank did not return within %.0f s (a rank that never "
                            "started?)\n", rank, world, g_timeout_s);
            return g_err = -8;
        }
        if (job->res != ncclSuccess) return fail_nccl(job->res, "ncclCommInitRank", __LINE__);
        comm = job->comm;
    }
    return new_handle(h, comm, rank, world, device);
}

int nsparse_dist_init_all(nsparse_dist_t *handles, int world)
{
    g_err = 0;
    if (!handles || world < 1) return g_err = -1;
    int before = 0;
    D_HIP(hipGetDevice(&before));  // (before the communicators exist: an early return owns nothing)
    std::vector<ncclComm_t> comms((size_t)world, nullptr);
    if (world > 1) D_NCCL(ncclCommInitAll(comms.data(), world, nullptr));  // devices 0 .. world-1
    int rc = 0;
    for (int r = 0; r < world; r++) handles[r] = nullptr;
    for (int r = 0; r < world && rc == 0; r++) {
        const hipError_t e = hipSetDevice(r);
        if (e != hipSuccess) {
            rc = fail_hip(e, "hipSetDevice", __LINE__);
            break;  // comms[r] is still ours: destroyed below
        }
        rc = new_handle(&handles[r], comms[r], r, world, r);
        comms[r] = nullptr;  // new_handle owns the communicator whatever it returns (it destroys it when it fails)
    }
    if (rc) {  // nothing half-built survives: handles made so far, communicators not yet handed over
        for (int r = 0; r < world; r++) {
            if (handles[r]) nsparse_dist_destroy(handles[r]), handles[r] = nullptr;
            else if (comms[r]) (void)ncclCommDestroy(comms[r]);
        }
    }
    (void)hipSetDevice(before);
    return rc;
}

void nsparse_dist_destroy(nsparse_dist_t h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)sync_watch(h);
    (void)nsparse_dist_release_matrix(h);
    if (h->comm) (void)ncclCommDestroy(h->comm);
    free_handle(h);
}

int nsparse_dist_release_matrix(nsparse_dist_t h)
{
    if (!h) return -1;
    if (h->stream) (void)sync_watch(h);
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->gexec = nullptr;
    h->graph = nullptr;
    h->g_y = nullptr;
    h->g_x = nullptr;
    h->g_gather = -1;
    if (h->have_amb) release_amb(h->amb);
    h->have_amb = false;
    memset(&h->amb, 0, sizeof(h->amb));
    if (h->staged) (void)hipFree(h->staged);
    if (h->d_cuts) (void)hipFree(h->d_cuts);
    h->staged = nullptr;
    h->d_cuts = nullptr;
    h->cuts.clear();
    h->M = h->rpr = h->m_local = 0;
    return 0;
}

int nsparse_dist_spmv_setup(nsparse_dist_t h, sfCSR *a_local, const int *cuts, real *d_x_any, sfPlan *plan)
{
    g_err = 0;
    if (!h || !a_local || !cuts || !plan) return g_err = -1;
    if (h->have_amb || !h->cuts.empty()) return g_err = -3;  // one matrix at a time (nsparse_dist_release_matrix)
    // every argument is checked before the handle changes: a refused call leaves it as it was
    if (cuts[0] != 0) return g_err = -1;
    int rpr = 1;
    for (int r = 0; r < h->world; r++) {
        const int len = cuts[r + 1] - cuts[r];
        if (len < 0) return g_err = -1;
        rpr = len > rpr ? len : rpr;
    }
    const int m_local = cuts[h->rank + 1] - cuts[h->rank];
    if (a_local->M != m_local) return g_err = -2;
    bool equal_blocks = true;
    for (int r = 0; r < h->world; r++)
        if (cuts[r + 1] > cuts[r] && (long long)cuts[r] != (long long)r * rpr) equal_blocks = false;
    real *staged = nullptr;
    int *d_cuts = nullptr;
    if (h->world > 1 && !equal_blocks) {
        hipError_t e = hipMalloc((void **)&staged, sizeof(real) * (size_t)h->world * (size_t)rpr);
        if (e == hipSuccess) e = hipMemset(staged, 0, sizeof(real) * (size_t)h->world * (size_t)rpr);
        if (e == hipSuccess) e = hipMalloc((void **)&d_cuts, sizeof(int) * ((size_t)h->world + 1));
        if (e == hipSuccess) e = hipMemcpy(d_cuts, cuts, sizeof(int) * ((size_t)h->world + 1), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (staged) (void)hipFree(staged);
            if (d_cuts) (void)hipFree(d_cuts);
            return fail_hip(e, "gather staging", __LINE__);
        }
    }
    if (m_local > 0) {
        // synchronous; takes the product library's API lock.  The callee clears its error word (per calling thread)
        // on entry, so anything but 0 afterwards is THIS call's failure -- also when it repeats an earlier code
        sf_csr2amb(&h->amb, a_local, d_x_any, plan);
        const int after = nsparse_last_error();
        if (after != 0) {
            if (staged) (void)hipFree(staged);
            if (d_cuts) (void)hipFree(d_cuts);
            memset(&h->amb, 0, sizeof(h->amb));
            return g_err = after;
        }
        h->have_amb = true;
    } else if (plan->isPlan == FALSE) {
        init_plan(plan);
    }
    h->cuts.assign(cuts, cuts + h->world + 1);
    h->M = cuts[h->world];
    h->rpr = rpr;
    h->m_local = m_local;
    h->equal_blocks = equal_blocks;
    h->staged = staged;
    h->d_cuts = d_cuts;
    h->plan = *plan;
    return 0;
}

long long nsparse_dist_y_elems(nsparse_dist_t h)
{
    const long long need = (long long)h->world * h->rpr;
    return need > h->M ? need : h->M;
}

int nsparse_dist_close_gaps(real *d_y, const real *d_staged, const int *d_cuts, int world, int rpr, int M, void *stream)
{
    if (M <= 0) return 0;
    hipLaunchKernelGGL(k_close_gaps, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_y, d_staged, d_cuts,
                       world, rpr, M);
    D_HIP(hipGetLastError());
    return 0;
}

int nsparse_dist_spmv(nsparse_dist_t h, real *d_y, const real *d_x, int gather)
{
    if (h->gexec && d_y == h->g_y && d_x == h->g_x && gather == h->g_gather) {
        D_HIP(hipGraphLaunch(h->gexec, h->stream));
        return 0;
    }
    return enqueue(h, d_y, d_x, gather);
}

int nsparse_dist_capture(nsparse_dist_t h, real *d_y, const real *d_x, int gather)
{
    g_err = 0;
    if (h->gexec) {
        (void)hipGraphExecDestroy(h->gexec);
        (void)hipGraphDestroy(h->graph);
        h->gexec = nullptr;
        h->graph = nullptr;
    }
    // warm: first-use allocations of RCCL (channels, proxies) must not fall inside the capture
    int rc = enqueue(h, d_y, d_x, gather);
    if (rc) return rc;
    rc = sync_watch(h);
    if (rc) return rc;
    D_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    rc = enqueue(h, d_y, d_x, gather);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return fail_hip(e, "hipStreamEndCapture", __LINE__);
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fail_hip(e, "hipGraphInstantiate", __LINE__);
    }
    h->graph = g;
    h->gexec = ge;
    h->g_y = d_y;
    h->g_x = d_x;
    h->g_gather = gather;
    return 0;
}

int nsparse_dist_sync(nsparse_dist_t h) { return sync_watch(h); }

double nsparse_dist_set_timeout(double seconds)
{
    const double old = g_timeout_s;
    if (seconds > 0.0) g_timeout_s = seconds;
    return old;
}

int nsparse_dist_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// All ranks: returns when every rank has reached the call and the handle's stream is idle.  A one-int ncclAllReduce on
// the handle's stream (world 1, or a handle without a communicator: the stream only).  -7 after the time-out.
int nsparse_dist_barrier(nsparse_dist_t h)
{
    if (!h) return g_err = -1;
    if (h->comm && h->world > 1) D_NCCL(ncclAllReduce(h->d_tick, h->d_tick + 1, 1, ncclInt, ncclSum, h->comm, h->stream));
    return sync_watch(h);
}

// vals[0 .. n) <- sum (op 0) or max (op 1) over the ranks, host memory in and out, n <= 64.  What a launcher needs
// around a timed loop (the slowest rank's time, the total work) without a second communication library.
int nsparse_dist_allreduce_f64(nsparse_dist_t h, double *vals, int n, int op)
{
    if (!h || !vals || n < 0 || n > kReduceMax || (op != 0 && op != 1)) return g_err = -1;
    if (!h->comm || h->world == 1 || n == 0) return h->world == 1 || n == 0 ? 0 : (g_err = -4);
    memcpy(h->h_red, vals, sizeof(double) * (size_t)n);
    D_HIP(hipMemcpyAsync(h->d_red, h->h_red, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    D_NCCL(ncclAllReduce(h->d_red, h->d_red, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, h->comm, h->stream));
    D_HIP(hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    const int rc = sync_watch(h);  // pinned on both sides: nothing above blocks the host, a missing peer ends here (-7)
    if (rc == 0) memcpy(vals, h->h_red, sizeof(double) * (size_t)n);
    return rc;
}

int nsparse_dist_spmv_loop(nsparse_dist_t h, real *d_y, const real *d_x, int gather, int iters,
                           double *ms_wall, double *ms_events, double *us_host)
{
    using clk = std::chrono::steady_clock;
    if (iters < 1) return g_err = -1;
    {
        const int rc = sync_watch(h);
        if (rc) return rc;
    }
    const auto t0 = clk::now();
    D_HIP(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < iters; i++) {
        const int rc = nsparse_dist_spmv(h, d_y, d_x, gather);
        if (rc) return rc;
    }
    D_HIP(hipEventRecord(h->ev[1], h->stream));
    const auto t1 = clk::now();
    {
        const int rc = sync_watch(h);
        if (rc) return rc;
    }
    const auto t2 = clk::now();
    float ms = 0;
    D_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    if (ms_wall) *ms_wall = std::chrono::duration<double, std::milli>(t2 - t0).count() / iters;
    if (ms_events) *ms_events = (double)ms / iters;
    if (us_host) *us_host = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
    return 0;
}

// ---- row-partitioned SpGEMM (stretch row; round 4: native) ------------------------------------------
// C[rows of rank r, :] = A[rows of rank r, :] * B with B whole on every rank: the single-GPU call on a row block (an
// A with fewer rows than B takes the library's column-range set-up), NO collective inside the algorithm.  What is
// new here is the partition (by intermediate products, what the reference's set_intprod_num counts,
// kernel_spgemm_hash_d.cu:70-86) and the optional assembly of the whole C on every rank.
int nsparse_dist_spgemm_row_work(const sfCSR *a, const sfCSR *b, long long *work)
{
    if (!a || !b || !work || !a->rpt || !a->col || !b->rpt) return g_err = -1;
    for (int i = 0; i < a->M; i++) {
        long long w = 0;
        for (int j = a->rpt[i]; j < a->rpt[i + 1]; j++) {
            const int k = a->col[j];
            if (k < 0 || k >= b->M) return g_err = -2;
            w += b->rpt[k + 1] - b->rpt[k];
        }
        work[i] = w;
    }
    return 0;
}

int nsparse_dist_spgemm(nsparse_dist_t h, sfCSR *a_block, sfCSR *b, sfCSR *c_block)
{
    g_err = 0;
    if (!h || !a_block || !b || !c_block) return g_err = -1;
    // c_block is an output: whatever its device pointers held on entry is not ours to free later
    c_block->d_rpt = nullptr;
    c_block->d_col = nullptr;
    c_block->d_val = nullptr;
    spgemm_kernel_hash(a_block, b, c_block);  // synchronous; an empty block gives an empty C of a_block->M rows
    // the callee clears its (per-thread) error word on entry: non-zero now = this call failed, c_block is not a result
    const int after = nsparse_last_error();
    if (after != 0) {
        // what the failed call had already allocated goes back to the library (advisor r05: it used to leak with the
        // memset), then the struct is emptied so that nobody reads a half-built C
        release_csr(*c_block);
        memset(c_block, 0, sizeof(*c_block));
        return g_err = after;
    }
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void k_shift_rpt(int *__restrict__ rpt, int n, int add)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) rpt[i] += add;
}
}  // namespace

// The whole C on every rank (device arrays from hipMalloc: release with nsparse_dist_release_gathered).  One
// all-reduce of the block sizes, then per rank one broadcast each of its rpt / col / val stretch straight to its
// place in the full arrays (world broadcasts of 3 messages: no padding, no staging copy; xGMI is point to point, a
// broadcast from r is r's stretch once over every link).  cuts: the world + 1 row cuts of the partition.
int nsparse_dist_spgemm_gather(nsparse_dist_t h, const int *cuts, const sfCSR *c_block, sfCSR *c_full)
{
    g_err = 0;
    if (!h || !cuts || !c_block || !c_full) return g_err = -1;
    const int world = h->world, rank = h->rank;
    if (world > kReduceMax) return g_err = -1;
    if (c_block->M != cuts[rank + 1] - cuts[rank]) return g_err = -2;
    double sizes[kReduceMax] = {};
    sizes[rank] = (double)c_block->nnz;
    if (world > 1) {
        if (!h->comm) return g_err = -4;
        const int rc = nsparse_dist_allreduce_f64(h, sizes, world, 0);
        if (rc) return rc;
    }
    long long total = 0;
    std::vector<long long> off((size_t)world + 1, 0);
    for (int r = 0; r < world; r++) {
        off[r] = total;
        total += (long long)sizes[r];
    }
    off[world] = total;
    if (total > 0x7fffffffLL) return g_err = -40;  // sfCSR keeps nnz and rpt in int (as the single-GPU call refuses)
    const int M = cuts[world];
    memset(c_full, 0, sizeof(*c_full));
    // Every array exists on every rank BEFORE the first broadcast: the ranks agree on that with one more all-reduce,
    // so an allocation failure anywhere is an error everywhere instead of peers parked inside ncclBroadcast.
    int *f_rpt = nullptr, *f_col = nullptr;
    real *f_val = nullptr;
    hipError_t ae = hipMalloc((void **)&f_rpt, sizeof(int) * ((size_t)M + 1));
    if (ae == hipSuccess) ae = hipMalloc((void **)&f_col, sizeof(int) * (size_t)(total > 0 ? total : 1));
    if (ae == hipSuccess) ae = hipMalloc((void **)&f_val, sizeof(real) * (size_t)(total > 0 ? total : 1));
    auto drop = [&]() {
        if (f_rpt) (void)hipFree(f_rpt);
        if (f_col) (void)hipFree(f_col);
        if (f_val) (void)hipFree(f_val);
        memset(c_full, 0, sizeof(*c_full));
    };
    double bad = ae == hipSuccess ? 0.0 : 1.0;
    if (world > 1) {
        const int rc = nsparse_dist_allreduce_f64(h, &bad, 1, 1);
        if (rc) {
            drop();
            return rc;
        }
    }
    if (bad != 0.0) {
        drop();
        return ae != hipSuccess ? fail_hip(ae, "hipMalloc of the gathered C", __LINE__) : (g_err = -2);  // -2: a peer could not
    }
    hipStream_t st = h->stream;
    int rc = 0;
#define G_HIP(x)                                                              \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (rc == 0 && e_ != hipSuccess) rc = fail_hip(e_, #x, __LINE__);     \
    } while (0)
#define G_NCCL(x)                                                             \
    do {                                                                      \
        ncclResult_t r_ = (x);                                                \
        if (rc == 0 && r_ != ncclSuccess) rc = fail_nccl(r_, #x, __LINE__);   \
    } while (0)
    for (int r = 0; r < world && rc == 0; r++) {
        const int m = cuts[r + 1] - cuts[r];
        const long long z = (long long)sizes[r];
        int *rpt_dst = f_rpt + cuts[r];
        if (world > 1) {
            // rows of rank r: its m row starts (the end of the last row is the start of the next block's first)
            if (m > 0) G_NCCL(ncclBroadcast(c_block->d_rpt, rpt_dst, (size_t)m, ncclInt, r, h->comm, st));
            if (z > 0 && rc == 0) G_NCCL(ncclBroadcast(c_block->d_col, f_col + off[r], (size_t)z, ncclInt, r, h->comm, st));
            if (z > 0 && rc == 0) G_NCCL(ncclBroadcast(c_block->d_val, f_val + off[r], (size_t)z, kNcclReal, r, h->comm, st));
        } else {
            if (m > 0) G_HIP(hipMemcpyAsync(rpt_dst, c_block->d_rpt, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, st));
            if (z > 0) {
                G_HIP(hipMemcpyAsync(f_col, c_block->d_col, sizeof(int) * (size_t)z, hipMemcpyDeviceToDevice, st));
                G_HIP(hipMemcpyAsync(f_val, c_block->d_val, sizeof(real) * (size_t)z, hipMemcpyDeviceToDevice, st));
            }
        }
        if (m > 0 && off[r] > 0 && rc == 0) {
            hipLaunchKernelGGL(k_shift_rpt, dim3((m + 255) / 256), dim3(256), 0, st, rpt_dst, m, (int)off[r]);
            G_HIP(hipGetLastError());
        }
    }
    int *h_last = reinterpret_cast<int *>(h->h_red + kReduceMax);  // pinned: the copy below reads it after we return from the call
    *h_last = (int)total;
    G_HIP(hipMemcpyAsync(f_rpt + M, h_last, sizeof(int), hipMemcpyHostToDevice, st));
#undef G_HIP
#undef G_NCCL
    const int wrc = sync_watch(h);  // also after an enqueue error: what was queued must be off the arrays before they go
    if (rc == 0) rc = wrc;
    if (rc) {  // a broadcast that failed half-way: nothing half-filled is handed out (peers end by their watchdog)
        drop();
        return rc;
    }
    c_full->M = M;
    c_full->N = c_block->N;
    c_full->nnz = (int)total;
    c_full->nnz_max = c_block->nnz_max;  // (a hint: the longest row of THIS rank's block)
    c_full->d_rpt = f_rpt;
    c_full->d_col = f_col;
    c_full->d_val = f_val;
    return 0;
}

void nsparse_dist_release_gathered(sfCSR c_full)
{
    if (c_full.d_rpt) (void)hipFree(c_full.d_rpt);
    if (c_full.d_col) (void)hipFree(c_full.d_col);
    if (c_full.d_val) (void)hipFree(c_full.d_val);
}

const sfAMB *nsparse_dist_amb(nsparse_dist_t h) { return &h->amb; }
const sfPlan *nsparse_dist_plan(nsparse_dist_t h) { return &h->plan; }
void *nsparse_dist_stream(nsparse_dist_t h) { return (void *)h->stream; }

}  // extern "C"
