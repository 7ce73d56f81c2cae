// tests/emu/emu_rccl.cpp -- see include/rccl/rccl.h.  Linked into the emulation build of libnsparse_dist only.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
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

namespace {
// ---- ranks in DIFFERENT processes (bench.py --gpus N spawns one process per rank): the same rendezvous in POSIX shared
// memory.  /dev/shm/emu-rccl-<id>: a control block (process-shared mutex + condition, the lengths the ranks post) and
// /dev/shm/emu-rccl-<id>-data: a staging area that grows to the largest collective -- every rank copies what it sends
// into its stretch, meets the others, and copies what it receives out of theirs.  Threads of one process may use it too.
struct ShmCtl {
    pthread_mutex_t m;
    pthread_cond_t cv;
    volatile int inited;
    int n, arrived, aborted;
    unsigned long long gen;
    unsigned long long len[64];
    unsigned long long data_bytes;
};
struct ShmGroup {
    ShmCtl *ctl = nullptr;
    std::string name;
    int fd_data = -1;
    unsigned char *data = nullptr;
    size_t mapped = 0;
};
bool shm_meet(ShmGroup &g)
{
    ShmCtl *c = g.ctl;
    pthread_mutex_lock(&c->m);
    if (c->aborted) {
        pthread_mutex_unlock(&c->m);
        return false;
    }
    const unsigned long long my = c->gen;
    if (++c->arrived == c->n) {
        c->arrived = 0;
        c->gen++;
        pthread_cond_broadcast(&c->cv);
        pthread_mutex_unlock(&c->m);
        return true;
    }
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const double t = timeout_s();
    ts.tv_sec += (time_t)t;
    ts.tv_nsec += (long)((t - (double)(time_t)t) * 1e9);
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
    int rc = 0;
    while (c->gen == my && !c->aborted && rc == 0) rc = pthread_cond_timedwait(&c->cv, &c->m, &ts);
    const bool ok = c->gen != my && !c->aborted;
    if (!ok) {
        c->aborted = 1;
        pthread_cond_broadcast(&c->cv);
    }
    pthread_mutex_unlock(&c->m);
    return ok;
}
ShmGroup *shm_join(const std::string &id, int nranks)
{
    auto *g = new ShmGroup;
    g->name = "/" + id;
    bool creator = true;
    int fd = shm_open(g->name.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) {
        creator = false;
        for (int spin = 0; fd < 0 && spin < 20000; spin++) {
            fd = shm_open(g->name.c_str(), O_RDWR, 0600);
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) { delete g; return nullptr; }
    }
    if (creator && ftruncate(fd, (off_t)sizeof(ShmCtl)) != 0) { close(fd); delete g; return nullptr; }
    if (!creator) {  // the creator's ftruncate may not have happened yet
        struct stat st;
        for (int spin = 0; spin < 20000; spin++) {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(ShmCtl)) break;
            usleep(1000);
        }
    }
    g->ctl = (ShmCtl *)mmap(nullptr, sizeof(ShmCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (g->ctl == (ShmCtl *)MAP_FAILED) { delete g; return nullptr; }
    if (creator) {
        pthread_mutexattr_t ma;
        pthread_condattr_t ca;
        pthread_mutexattr_init(&ma);
        pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
        pthread_condattr_init(&ca);
        pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
        pthread_mutex_init(&g->ctl->m, &ma);
        pthread_cond_init(&g->ctl->cv, &ca);
        g->ctl->n = nranks;
        g->ctl->arrived = g->ctl->aborted = 0;
        g->ctl->gen = 0;
        g->ctl->data_bytes = 0;
        __atomic_store_n(&g->ctl->inited, 1, __ATOMIC_RELEASE);
    } else {
        for (int spin = 0; __atomic_load_n(&g->ctl->inited, __ATOMIC_ACQUIRE) != 1 && spin < 20000; spin++) usleep(1000);
        if (g->ctl->inited != 1 || g->ctl->n != nranks) { delete g; return nullptr; }
    }
    g->fd_data = shm_open((g->name + "-data").c_str(), O_RDWR | O_CREAT, 0600);
    if (g->fd_data < 0) { delete g; return nullptr; }
    return g;
}
void shm_leave(ShmGroup *g)
{
    if (!g) return;
    if (g->data) munmap(g->data, g->mapped);
    if (g->fd_data >= 0) close(g->fd_data);
    shm_unlink(g->name.c_str());  // (the first rank to leave removes the names; the mappings of the others live on)
    shm_unlink((g->name + "-data").c_str());
    if (g->ctl) munmap(g->ctl, sizeof(ShmCtl));
    delete g;
}
// every rank posts `nb` bytes from `send`; on return data/off describe where each rank's bytes are (until shm_done)
bool shm_exchange(ShmGroup &g, int rank, const void *send, size_t nb, std::vector<size_t> &off)
{
    ShmCtl *c = g.ctl;
    pthread_mutex_lock(&c->m);
    c->len[rank] = nb;
    pthread_mutex_unlock(&c->m);
    if (!shm_meet(g)) return false;  // all lengths posted
    off.assign((size_t)c->n + 1, 0);
    for (int r = 0; r < c->n; r++) off[r + 1] = off[r] + ((c->len[r] + 63) & ~63ull);
    const size_t total = off[c->n] ? off[c->n] : 64;
    if (rank == 0 && c->data_bytes < total) {
        if (ftruncate(g.fd_data, (off_t)total) != 0) return false;
        c->data_bytes = total;
    }
    if (!shm_meet(g)) return false;  // the staging area is large enough
    if (g.mapped < c->data_bytes) {
        if (g.data) munmap(g.data, g.mapped);
        g.mapped = c->data_bytes;
        g.data = (unsigned char *)mmap(nullptr, g.mapped, PROT_READ | PROT_WRITE, MAP_SHARED, g.fd_data, 0);
        if (g.data == (unsigned char *)MAP_FAILED) { g.data = nullptr; g.mapped = 0; return false; }
    }
    if (nb) memcpy(g.data + off[rank], send, nb);
    return shm_meet(g);  // everybody's bytes are there
}
}  // namespace

struct ncclComm {
    std::shared_ptr<Group> g;  // ranks = threads of this process (ncclCommInitAll)
    ShmGroup *shm = nullptr;   // ranks by unique id (ncclCommInitRank): threads or processes
    int rank = 0, n = 1;
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
static void reduce_typed(void *dst, const std::vector<const void *> &src, size_t count, ncclDataType_t t, ncclRedOp_t op)
{
    switch (t) {
    case ncclInt32: reduce_into((int *)dst, src, count, op); break;
    case ncclUint32: reduce_into((unsigned *)dst, src, count, op); break;
    case ncclInt64: reduce_into((long long *)dst, src, count, op); break;
    case ncclUint64: reduce_into((unsigned long long *)dst, src, count, op); break;
    case ncclFloat32: reduce_into((float *)dst, src, count, op); break;
    case ncclFloat64: reduce_into((double *)dst, src, count, op); break;
    default: fprintf(stderr, "emu rccl: all-reduce of type %d\n", (int)t); abort();
    }
}

// One collective: every rank contributes `nb` bytes from `send`; `work(src)` runs on every rank with src[r] = rank r's
// bytes (its own buffer for thread groups, the staging area for shared-memory groups), then the ranks meet once more so
// that nobody's bytes go away while somebody still reads them.
template <typename F>
static ncclResult_t collective(ncclComm_t c, const void *send, size_t nb, F work)
{
    if (!c) return ncclInvalidArgument;
    std::vector<const void *> src((size_t)c->n);
    if (c->n == 1) {
        src[0] = send;
        work(src);
        return ncclSuccess;
    }
    if (c->shm) {
        std::vector<size_t> off;
        if (!shm_exchange(*c->shm, c->rank, send, nb, off)) return ncclSystemError;
        for (int r = 0; r < c->n; r++) src[r] = c->shm->data + off[r];
        work(src);
        return shm_meet(*c->shm) ? ncclSuccess : ncclSystemError;
    }
    Group &g = *c->g;
    {
        std::lock_guard<std::mutex> lk(g.m);
        g.send[c->rank] = send;
    }
    if (!rendezvous(g)) return ncclSystemError;
    for (int r = 0; r < c->n; r++) src[r] = g.send[r];
    work(src);
    return rendezvous(g) ? ncclSuccess : ncclSystemError;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::mutex m;
    static unsigned long long counter = 0;
    std::lock_guard<std::mutex> lk(m);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "emu-rccl-%d-%llu-%lx", (int)getpid(), ++counter, (long)ts.tv_nsec);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    auto *c = new ncclComm;
    c->rank = rank;
    c->n = nranks;
    if (nranks > 1) {
        c->shm = shm_join(std::string(id.internal, strnlen(id.internal, sizeof(id.internal))), nranks);
        if (!c->shm) {
            delete c;
            return ncclSystemError;
        }
        if (!shm_meet(*c->shm)) {  // returns when ALL ranks have called it, like the real one
            shm_leave(c->shm);
            delete c;
            return ncclSystemError;
        }
        if (rank == 0) {  // everybody has both objects open: the names can go now, so that a rank killed later leaks nothing
            shm_unlink(c->shm->name.c_str());
            shm_unlink((c->shm->name + "-data").c_str());
        }
    }
    *comm = c;
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
    for (int r = 0; r < ndev; r++) {
        comms[r] = new ncclComm;
        comms[r]->g = g;
        comms[r]->rank = r;
        comms[r]->n = ndev;
    }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (comm && comm->shm) shm_leave(comm->shm);
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
    if (comm && comm->shm) {
        pthread_mutex_lock(&comm->shm->ctl->m);
        comm->shm->ctl->aborted = 1;
        pthread_cond_broadcast(&comm->shm->ctl->cv);
        pthread_mutex_unlock(&comm->shm->ctl->m);
        shm_leave(comm->shm);
    }
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *e)
{
    const bool bad = comm && ((comm->g && comm->g->aborted) || (comm->shm && comm->shm->ctl->aborted));
    *e = bad ? ncclSystemError : ncclSuccess;
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
    return collective(c, send, nb, [&](const std::vector<const void *> &src) {
        for (size_t r = 0; r < src.size(); r++) {
            unsigned char *dst = (unsigned char *)recv + r * nb;
            if (dst != src[r] && nb) memmove(dst, src[r], nb);
        }
    });
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)ncclAllReduce(send, recv, count, t, op, c, nullptr); })) return ncclSuccess;
    // (operands may be in place: reduce into a private buffer, store after the last meeting)
    std::vector<unsigned char> out(count * width(t));
    const ncclResult_t rc = collective(c, send, out.size(), [&](const std::vector<const void *> &src) {
        if (count) reduce_typed(out.data(), src, count, t, op);
    });
    if (rc == ncclSuccess && !out.empty()) memcpy(recv, out.data(), out.size());
    return rc;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)ncclBroadcast(send, recv, count, t, root, c, nullptr); })) return ncclSuccess;
    const size_t nb = count * width(t);
    // only the root's bytes travel (the others post nothing)
    return collective(c, send, c && c->rank == root ? nb : 0, [&](const std::vector<const void *> &src) {
        if (recv != src[(size_t)root] && nb) memmove(recv, src[(size_t)root], nb);
    });
}
}
