// tests/emu/emu_rccl.cpp -- see include/rccl/rccl.h.  Linked into the emulation build of libnsparse_dist only.
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace emu {
bool capture_record(hipStream_t st, std::function<void()> fn);  // emu_core.cpp: true when `st` is being captured
}

namespace {
struct Group {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    int joined = 0;
    bool aborted = false;
    std::vector<const void *> send;
    std::vector<void *> recv;
};
double timeout_s()
{
    const char *e = getenv("EMU_NCCL_TIMEOUT_S");
    return e && atof(e) > 0 ? atof(e) : 30.0;
}
// all ranks of the group: returns false when a rank did not arrive in time (or the group was aborted)
bool rendezvous(Group &g)
{
    std::unique_lock<std::mutex> lk(g.m);
    if (g.aborted) return false;
    const unsigned long long my = g.gen;
    if (++g.arrived == g.n) {
        g.arrived = 0;
        g.gen++;
        g.cv.notify_all();
        return true;
    }
    const bool ok = g.cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), [&] { return g.gen != my || g.aborted; });
    if (!ok || g.aborted) {
        g.aborted = true;
        g.cv.notify_all();
        return false;
    }
    return true;
}
size_t width(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}
std::mutex g_reg_mu;
std::map<std::string, std::shared_ptr<Group>> g_by_id;
}  // namespace

struct ncclComm {
    std::shared_ptr<Group> g;
    int rank = 0;
};

template <typename T>
static void reduce_into(T *dst, const std::vector<const void *> &src, size_t count, ncclRedOp_t op)
{
    std::vector<T> out(count);
    for (size_t i = 0; i < count; i++) {
        T acc = static_cast<const T *>(src[0])[i];
        for (size_t r = 1; r < src.size(); r++) {
            const T v = static_cast<const T *>(src[r])[i];
            acc = op == ncclSum ? acc + v : op == ncclProd ? acc * v : op == ncclMax ? (v > acc ? v : acc) : (v < acc ? v : acc);
        }
        out[i] = acc;
    }
    memcpy(dst, out.data(), sizeof(T) * count);
}

// post my buffers, meet, let `work` run on every rank with everybody's buffers visible, meet again (buffers stay
// valid until every rank has read them)
template <typename F>
static ncclResult_t collective(ncclComm_t c, const void *send, void *recv, F work)
{
    if (!c || !c->g) return ncclInvalidArgument;
    Group &g = *c->g;
    if (g.n == 1) {
        g.send[0] = send;
        g.recv[0] = recv;
        work(g);
        return ncclSuccess;
    }
    {
        std::lock_guard<std::mutex> lk(g.m);
        g.send[c->rank] = send;
        g.recv[c->rank] = recv;
    }
    if (!rendezvous(g)) return ncclSystemError;
    std::vector<unsigned char> stage;
    work(g);
    if (!rendezvous(g)) return ncclSystemError;
    return ncclSuccess;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::mutex m;
    static unsigned long long counter = 0;
    std::lock_guard<std::mutex> lk(m);
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "emu-rccl-%d-%llu", (int)getpid(), ++counter);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto &slot = g_by_id[std::string(id.internal, strnlen(id.internal, sizeof(id.internal)))];
        if (!slot) {
            slot = std::make_shared<Group>();
            slot->n = nranks;
            slot->send.resize((size_t)nranks);
            slot->recv.resize((size_t)nranks);
        }
        g = slot;
    }
    if (g->n != nranks) return ncclInvalidArgument;
    if (nranks > 1 && !rendezvous(*g)) return ncclSystemError;  // returns when ALL ranks have called it, like the real one
    *comm = new ncclComm{g, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    int have = 0;
    hipGetDeviceCount(&have);
    if (ndev > have) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->n = ndev;
    g->send.resize((size_t)ndev);
    g->recv.resize((size_t)ndev);
    for (int r = 0; r < ndev; r++) comms[r] = new ncclComm{g, r};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    if (comm && comm->g) {
        std::lock_guard<std::mutex> lk(comm->g->m);
        comm->g->aborted = true;
        comm->g->cv.notify_all();
    }
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *e)
{
    *e = comm && comm->g && comm->g->aborted ? ncclSystemError : ncclSuccess;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "a rank did not arrive (emulated communicator)";
    case ncclInvalidArgument: return "invalid argument";
    default: return "error (emulated communicator)";
    }
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)ncclAllGather(send, recv, count, t, c, nullptr); })) return ncclSuccess;
    const size_t nb = count * width(t);
    const int me = c ? c->rank : 0;
    return collective(c, send, recv, [&](Group &g) {
        for (int r = 0; r < g.n; r++) {
            unsigned char *dst = (unsigned char *)g.recv[me] + (size_t)r * nb;
            if (dst != g.send[r] && nb) memmove(dst, g.send[r], nb);
        }
    });
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)ncclAllReduce(send, recv, count, t, op, c, nullptr); })) return ncclSuccess;
    const int me = c ? c->rank : 0;
    // in-place operands: every rank reduces into a private buffer first, the results are stored after the second meeting
    std::vector<unsigned char> out(count * width(t));
    const ncclResult_t rc = collective(c, send, recv, [&](Group &g) {
        switch (t) {
        case ncclInt32: reduce_into((int *)out.data(), g.send, count, op); break;
        case ncclUint32: reduce_into((unsigned *)out.data(), g.send, count, op); break;
        case ncclInt64: reduce_into((long long *)out.data(), g.send, count, op); break;
        case ncclUint64: reduce_into((unsigned long long *)out.data(), g.send, count, op); break;
        case ncclFloat32: reduce_into((float *)out.data(), g.send, count, op); break;
        case ncclFloat64: reduce_into((double *)out.data(), g.send, count, op); break;
        default: fprintf(stderr, "emu rccl: all-reduce of type %d\n", (int)t); abort();
        }
        (void)me;
    });
    if (rc == ncclSuccess && !out.empty()) memcpy(recv, out.data(), out.size());
    return rc;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)ncclBroadcast(send, recv, count, t, root, c, nullptr); })) return ncclSuccess;
    const size_t nb = count * width(t);
    const int me = c ? c->rank : 0;
    return collective(c, send, recv, [&](Group &g) {
        if (g.recv[me] != g.send[root] && nb) memmove(g.recv[me], g.send[root], nb);
    });
}
}
