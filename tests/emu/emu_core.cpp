// tests/emu/emu_core.cpp -- the scheduler and the fake runtime of the CPU emulation (see include/hip/hip_runtime.h).
// TEST INFRASTRUCTURE.  One OS thread per resident workgroup (a pool of EMU_WORKERS threads, default 8 = the
// "compute units" the fake device reports), one fiber per work-item inside it, cross-lane instructions evaluated
// over the lanes that reached the same call site.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <set>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <ucontext.h>
#include <unistd.h>
#include <sys/mman.h>

struct ihipStream_t {
    bool capturing = false;
    std::shared_ptr<std::vector<std::function<void()>>> rec;  // operations recorded while capturing
};

namespace emu {

thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local unsigned char *t_dyn_lds = nullptr;

// ---- fibers -------------------------------------------------------------------------------------------------
extern "C" void emu_switch(void **save_sp, void *to_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");

enum State { RUN, WAIT_WAVE, WAIT_BLOCK, YIELD, DONE };
constexpr size_t kStack = 96 * 1024;

struct Lane {
    void *sp = nullptr;
    unsigned char *stack = nullptr;
    State st = DONE;
    const void *site = nullptr;
    Req rq;
    Res rs;
    unsigned tid = 0;
};

struct Worker {
    int index = 0;             // position in the pool: a launch that a CU model limits to n resident workgroups uses workers 0 .. n-1
    std::vector<Lane> lanes;   // stacks are kept between workgroups
    void *sched_sp = nullptr;
    Lane *cur = nullptr;
    const std::function<void()> *body = nullptr;
    std::vector<unsigned char> dyn;
};
thread_local Worker *t_w = nullptr;

std::atomic<long long> g_stats[8];

static void lane_entry()
{
    Worker *w = t_w;
    (*w->body)();
    w = t_w;
    w->cur->st = DONE;
    emu_switch(&w->cur->sp, w->sched_sp);
    abort();  // a finished lane is never resumed
}

static void prepare(Lane &l)
{
    if (!l.stack) {
        l.stack = (unsigned char *)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (l.stack == (unsigned char *)MAP_FAILED) {
            fprintf(stderr, "emu: cannot map a lane stack\n");
            abort();
        }
    }
    // the frame emu_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into lane_entry with rsp = 8 mod 16
    uintptr_t top = ((uintptr_t)l.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)(top - 8);  // after `ret` pops the entry address: rsp = top - 8 (a call's alignment)
    *--sp = (void *)lane_entry;
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    l.sp = sp;
}

static inline void to_scheduler()
{
    Worker *w = t_w;
    emu_switch(&w->cur->sp, w->sched_sp);
}

Res collective(const Req &rq)
{
    Worker *w = t_w;
    Lane *l = w->cur;
    l->site = __builtin_extract_return_addr(__builtin_return_address(0));
    l->rq = rq;
    l->st = WAIT_WAVE;
    to_scheduler();
    return l->rs;
}

void block_barrier()
{
    Lane *l = t_w->cur;
    l->st = WAIT_BLOCK;
    to_scheduler();
}

void yield_lane()
{
    Lane *l = t_w->cur;
    l->st = YIELD;
    to_scheduler();
}

// ---- cross-lane instructions ------------------------------------------------------------------------------
// DPP source lane of `lane` under dpp_ctrl; -1 = invalid (out of row / beyond the wave)
static int dpp_src(int lane, int ctrl)
{
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0x000 && ctrl <= 0x0ff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);  // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = r + (ctrl & 15); return s < 16 ? row + s : -1; }   // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = r - (ctrl & 15); return s >= 0 ? row + s : -1; }   // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((r - (ctrl & 15)) & 15);                            // row_ror
    if (ctrl == 0x130) return lane + 1 < 64 ? lane + 1 : -1;   // wave_shl:1
    if (ctrl == 0x134) return (lane + 1) & 63;                 // wave_rol:1
    if (ctrl == 0x138) return lane - 1 >= 0 ? lane - 1 : -1;   // wave_shr:1
    if (ctrl == 0x13c) return (lane - 1) & 63;                 // wave_ror:1
    if (ctrl == 0x140) return row + (15 - r);                  // row_mirror
    if (ctrl == 0x141) return row + ((r & 8) | (7 - (r & 7))); // row_half_mirror
    if (ctrl == 0x142) return (row >= 16 && r >= 0) ? row - 1 : -2;  // row_bcast:15: lane 15 of the previous row (rows 1..3)
    if (ctrl == 0x143) return lane >= 32 ? 31 : -2;                  // row_bcast:31: lane 31 to rows 2, 3
    fprintf(stderr, "emu: dpp_ctrl 0x%x is not modelled\n", ctrl);
    abort();
}

static void evaluate(Lane **wl, unsigned long long mask)
{
    // wl[0..64): the lanes of the wave (nullptr beyond the workgroup); mask: the lanes this instruction executes on
    auto in = [&](int i) { return i >= 0 && i < 64 && ((mask >> i) & 1ull); };
    const int first = __builtin_ctzll(mask);
    const Req &r0 = wl[first]->rq;
    Res out[64];
    for (int i = 0; i < 64; i++) {
        if (!in(i)) continue;
        const Req &q = wl[i]->rq;
        Res o = {0, 0};
        switch (q.op) {
        case OP_BALLOT: {
            unsigned long long b = 0;
            for (int j = 0; j < 64; j++)
                if (in(j) && wl[j]->rq.a) b |= 1ull << j;
            o.r0 = b;
            break;
        }
        case OP_SHFL: case OP_SHFL_UP: case OP_SHFL_DOWN: case OP_SHFL_XOR: {
            const int wd = q.d;
            int src;
            if (q.op == OP_SHFL) src = (i & ~(wd - 1)) | (q.c & (wd - 1));
            else if (q.op == OP_SHFL_UP) { src = i - q.c; if (src < (i & ~(wd - 1))) src = i; }
            else if (q.op == OP_SHFL_DOWN) { src = i + q.c; if ((i & (wd - 1)) + q.c >= wd) src = i; }
            else { src = i ^ q.c; if (src >= ((i + wd) & ~(wd - 1))) src = i; }
            src &= 63;
            if (in(src)) o.r0 = wl[src]->rq.a;
            else { o.r0 = 0; g_stats[2]++; }
            break;
        }
        case OP_READLANE:
            if (!in(q.c & 63)) {
                fprintf(stderr, "emu: v_readlane of lane %d, which does not execute the instruction (mask %016llx)\n", q.c, mask);
                abort();
            }
            o.r0 = wl[q.c & 63]->rq.a;
            break;
        case OP_READFIRST: o.r0 = r0.a; break;
        case OP_BPERMUTE: {
            const int src = (int)((q.a >> 2) & 63);
            if (in(src)) o.r0 = wl[src]->rq.b;
            else { o.r0 = 0; g_stats[2]++; }
            break;
        }
        case OP_SWIZZLE: {
            int src;
            if (q.c & 0x8000) src = (i & ~3) | ((q.c >> (2 * (i & 3))) & 3);
            else {
                const int am = q.c & 31, om = (q.c >> 5) & 31, xm = (q.c >> 10) & 31;
                src = (i & 32) | ((((i & 31) & am) | om) ^ xm);
            }
            if (in(src)) o.r0 = wl[src]->rq.a;
            else { o.r0 = 0; g_stats[2]++; }
            break;
        }
        case OP_PERMLANE32_SWAP: {
            // v_permlane32_swap vdst, src: vdst[i + 32] <-> src[i] for i < 32
            const int p = i ^ 32;
            if (!in(p)) { o.r0 = q.a; o.r1 = q.b; g_stats[2]++; break; }
            if (i < 32) { o.r0 = q.a; o.r1 = wl[p]->rq.a; }   // my src takes the partner's vdst
            else { o.r0 = wl[p]->rq.b; o.r1 = q.b; }          // my vdst takes the partner's src
            break;
        }
        case OP_WAVE_BARRIER: break;
        case OP_DPP: case OP_DPP_MIN: case OP_DPP_MAX: {
            const int ctrl = q.c, rm = q.d, bm = q.e;
            const bool bc = q.op == OP_DPP ? q.f != 0 : false;
            const bool enabled = ((rm >> (i >> 4)) & 1) && ((bm >> ((i >> 2) & 3)) & 1);
            const int s = dpp_src(i, ctrl);
            o.r0 = (unsigned)q.a;  // old
            if (!enabled || s == -2) break;  // (-2: a broadcast that does not reach this row writes nothing)
            const bool valid = s >= 0 && in(s);
            if (!valid) {
                g_stats[4]++;
                if (q.op == OP_DPP && bc) o.r0 = 0;  // bound_ctrl:0 reads 0 from an invalid lane
                break;                               // else: the lane is not written
            }
            if (q.op == OP_DPP) o.r0 = (unsigned)wl[s]->rq.b;
            else {
                const int x = (int)(unsigned)(wl[s]->rq.b >> 32), y = (int)(unsigned)(q.b & 0xffffffffu);
                o.r0 = (unsigned)(q.op == OP_DPP_MIN ? (x < y ? x : y) : (x > y ? x : y));
            }
            break;
        }
        default: fprintf(stderr, "emu: op %d\n", q.op); abort();
        }
        out[i] = o;
    }
    for (int i = 0; i < 64; i++)
        if (in(i)) { wl[i]->rs = out[i]; wl[i]->st = RUN; }
}

// ---- one workgroup ------------------------------------------------------------------------------------------
static void run_block(Worker *w, const std::function<void()> &body, dim3 grid, dim3 block, unsigned bx, size_t lds)
{
    const unsigned nt = block.x * block.y * block.z;
    if (w->lanes.size() < nt) w->lanes.resize(nt);
    if (w->dyn.size() < lds + 64) w->dyn.resize(lds + 64);
    w->body = &body;
    t_dyn_lds = (unsigned char *)(((uintptr_t)w->dyn.data() + 63) & ~(uintptr_t)63);
    t_blockIdx = {bx, 0, 0};
    t_blockDim = {block.x, block.y, block.z};
    t_gridDim = {grid.x, grid.y, grid.z};
    for (unsigned t = 0; t < nt; t++) {
        Lane &l = w->lanes[t];
        prepare(l);
        l.st = RUN;
        l.tid = t;
    }
    const unsigned nw = (nt + 63) / 64;
    auto resume = [&](Lane &l) {
        w->cur = &l;
        t_threadIdx = {l.tid % block.x, (l.tid / block.x) % block.y, l.tid / (block.x * block.y)};
        emu_switch(&w->sched_sp, l.sp);
    };
    // EMU_ORDER: the order in which the waves of a workgroup, and the lanes of a wave, are run between sync points --
    // 0 ascending (default), 1 descending, 2 a different pseudo-random order per workgroup and scheduling round.  A
    // program without races gives the same results under every order: running the tests under 1 and 2 is how a missing
    // barrier (a wave reading what another has not written yet) shows up here instead of once a week on the device.
    static const int order_mode = getenv("EMU_ORDER") ? atoi(getenv("EMU_ORDER")) : 0;
    unsigned long long rng = 0x9E3779B97F4A7C15ull * (bx + 1) + 12345;
    auto next_rand = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    std::vector<unsigned> worder(nw);
    int lorder[64];
    for (;;) {
        unsigned at_barrier = 0, done = 0;
        for (unsigned i = 0; i < nw; i++) worder[i] = order_mode == 1 ? nw - 1 - i : i;
        for (int i = 0; i < 64; i++) lorder[i] = order_mode == 1 ? 63 - i : i;
        if (order_mode == 2) {
            for (unsigned i = nw; i > 1; i--) std::swap(worder[i - 1], worder[next_rand() % i]);
            for (int i = 64; i > 1; i--) std::swap(lorder[i - 1], lorder[next_rand() % (unsigned)i]);
        }
        for (unsigned wi = 0; wi < nw; wi++) {
            const unsigned wv = worder[wi];
            Lane *wl[64];
            for (int i = 0; i < 64; i++) wl[i] = wv * 64 + i < nt ? &w->lanes[wv * 64 + i] : nullptr;
            for (;;) {
                bool ran = false;
                for (int li = 0, i = lorder[0]; li < 64; li++, i = lorder[li & 63])
                    if (wl[i] && (wl[i]->st == RUN || wl[i]->st == YIELD)) {
                        const bool was_yield = wl[i]->st == YIELD;
                        wl[i]->st = RUN;
                        resume(*wl[i]);
                        ran = true;
                        if (was_yield && wl[i]->st == YIELD) sched_yield();  // a spin-wait on another workgroup
                    }
                // every live lane of the wave is parked now (or yielded again)
                const void *site = nullptr;
                unsigned long long waiting = 0, yielded = 0;
                for (int i = 0; i < 64; i++) {
                    if (!wl[i]) continue;
                    if (wl[i]->st == WAIT_WAVE) {
                        waiting |= 1ull << i;
                        if (!site || (uintptr_t)wl[i]->site < (uintptr_t)site) site = wl[i]->site;
                    } else if (wl[i]->st == YIELD) yielded |= 1ull << i;
                }
                if (yielded) continue;  // spinning lanes: keep polling (the other workgroups run on their own threads)
                if (!waiting) break;
                unsigned long long mask = 0;
                for (int i = 0; i < 64; i++)
                    if (((waiting >> i) & 1ull) && wl[i]->site == site) mask |= 1ull << i;
                if (mask != waiting) g_stats[1]++;
                // one instruction = one opcode for all its lanes
                const int op0 = wl[__builtin_ctzll(mask)]->rq.op;
                for (int i = 0; i < 64; i++)
                    if (((mask >> i) & 1ull) && wl[i]->rq.op != op0) {
                        fprintf(stderr, "emu: lanes at one call site with different operations (%d / %d)\n", op0, wl[i]->rq.op);
                        abort();
                    }
                evaluate(wl, mask);
                (void)ran;
            }
            for (int i = 0; i < 64; i++) {
                if (!wl[i]) continue;
                at_barrier += wl[i]->st == WAIT_BLOCK;
                done += wl[i]->st == DONE;
            }
        }
        if (done == nt) break;
        if (at_barrier + done != nt) {
            fprintf(stderr, "emu: workgroup %u stuck (%u at the barrier, %u done of %u)\n", bx, at_barrier, done, nt);
            abort();
        }
        // s_barrier: every wave that still runs has arrived
        for (unsigned t = 0; t < nt; t++)
            if (w->lanes[t].st == WAIT_BLOCK) w->lanes[t].st = RUN;
    }
    g_stats[3]++;
}

// ---- worker pool -------------------------------------------------------------------------------------------
struct Job {
    const char *name = "?";
    std::function<void()> body;
    dim3 grid, block;
    size_t lds = 0;
    std::atomic<unsigned> next{0};
    std::atomic<unsigned> finished{0};
    unsigned total = 0;
    int active = 1 << 30;  // workgroups of this launch that may be resident at once (compute-unit model, below)
};
// (never destroyed: the detached workers wait on them until the process ends)
static std::mutex &g_mu = *new std::mutex;
static std::condition_variable &g_cv = *new std::condition_variable, &g_cv_done = *new std::condition_variable;
static std::shared_ptr<Job> &g_job = *new std::shared_ptr<Job>;
static unsigned long long g_job_seq = 0;
static std::vector<std::thread> &g_threads = *new std::vector<std::thread>;
static int g_workers = 0;

static thread_local const char *t_kernel = "?";
extern "C" void emu_describe_address(const void *a);
static void on_fault(int sig, siginfo_t *si, void *uc_)
{
    const ucontext_t *uc = (const ucontext_t *)uc_;
    const bool wr = (uc->uc_mcontext.gregs[REG_ERR] & 2) != 0;
    char buf[512];
    const int n = snprintf(buf, sizeof buf, "emu: signal %d (%s) at address %p in kernel %s, workgroup %u, work-item %u\n", sig, wr ? "write" : "read", si->si_addr,
                           t_kernel, t_blockIdx.x, t_threadIdx.x);
    (void)!write(2, buf, (size_t)n);
    emu_describe_address(si->si_addr);
    void *bt[48];
    const int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
    _exit(139);
}
static void on_watchdog(int, siginfo_t *, void *)
{
    char buf[256];
    const int n = snprintf(buf, sizeof buf, "emu: watchdog: kernel %s, workgroup %u, work-item %u is here:\n", t_kernel, t_blockIdx.x,
                           t_threadIdx.x);
    (void)!write(2, buf, (size_t)n);
    void *bt[16];
    const int k = backtrace(bt, 16);
    backtrace_symbols_fd(bt, k, 2);
}
static void install_trap()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("EMU_NO_TRAP")) return;
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_fault;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
        struct sigaction sw;
        memset(&sw, 0, sizeof sw);
        sw.sa_sigaction = on_watchdog;
        sw.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGUSR1, &sw, nullptr);
    });
}

unsigned long long clock_div()
{
    static const unsigned long long d = [] {
        const char *e = getenv("EMU_CLOCK_DIV");
        const long long v = e ? atoll(e) : 20;
        return (unsigned long long)(v > 0 ? v : 20);
    }();
    return d;
}

int workers()
{
    if (g_workers == 0) {
        const char *e = getenv("EMU_WORKERS");
        g_workers = e && atoi(e) > 0 ? atoi(e) : 8;
#ifdef EMU_SHARED_STATIC
        g_workers = 1;  // LDS is one static per kernel in this build: one workgroup at a time
#endif
    }
    return g_workers;
}

// ---- compute-unit model (round 6) ------------------------------------------------------------------------------------
// The pool of EMU_WORKERS threads stands for EMU_CUS compute units (default: as many as workers; tests/test_emu_cpu.py runs the
// selftest with half as many, so that a CU can hold two small workgroups and only one big one) of 160 KiB of LDS and 32
// wavefront slots each.  A launch may have  CUs x min(160 KiB / (static + dynamic LDS), 32 / wavefronts per
// workgroup, register occupancy x 4 / wavefronts per workgroup)  workgroups resident at once, never more than there are
// workers: a kernel that takes more than half a CU's LDS runs one workgroup per CU, and that is what a co-residency census
// (fused.h: k_census) counts and what a grid barrier can rely on -- not "8 workers".  Static LDS and occupancy come from the
// REAL compiler (gfx950): `<library>.lds`, written by tools/kernel_resources.py next to the library (tests/emu/Makefile),
// or emu_set_static_lds() (selftest).  A kernel the table does not know has static LDS 0.
constexpr size_t kLdsPerCu = 160 * 1024;
int cus()
{
    static const int n = [] {
        const char *e = getenv("EMU_CUS");
        const int v = e && atoi(e) > 0 ? atoi(e) : workers();
        return v > 0 ? v : 1;
    }();
    return n;
}
struct KernelRes {
    size_t lds = 0;
    int occ = 8;  // waves per SIMD the register allocation allows
};
static std::mutex &g_res_mu = *new std::mutex;
static std::map<std::string, KernelRes> &g_res_by_name = *new std::map<std::string, KernelRes>;
static std::map<const void *, KernelRes> &g_res_by_func = *new std::map<const void *, KernelRes>;
static std::set<std::string> &g_res_files = *new std::set<std::string>;
void set_static_lds(const void *func, size_t bytes, int occ)
{
    std::lock_guard<std::mutex> lk(g_res_mu);
    g_res_by_func[func] = KernelRes{bytes, occ > 0 ? occ : 8};
}
static KernelRes resources_of(const void *func)
{
    std::lock_guard<std::mutex> lk(g_res_mu);
    auto it = g_res_by_func.find(func);
    if (it != g_res_by_func.end()) return it->second;
    KernelRes r;
    Dl_info di;
    if (dladdr(func, &di) && di.dli_fname) {
        const std::string file = std::string(di.dli_fname) + ".lds";
        if (g_res_files.insert(file).second) {  // first kernel of this library: read its table
            if (FILE *f = fopen(file.c_str(), "r")) {
                char name[4096];
                long lds;
                int occ, vgpr;
                while (fscanf(f, "%4095s %ld %d %d", name, &lds, &occ, &vgpr) == 4) g_res_by_name[name] = KernelRes{(size_t)lds, occ > 0 ? occ : 8};
                fclose(f);
            }
        }
        if (di.dli_sname) {
            auto jt = g_res_by_name.find(di.dli_sname);
            if (jt != g_res_by_name.end()) r = jt->second;
        }
    }
    g_res_by_func[func] = r;
    return r;
}

static std::vector<pthread_t> &g_worker_ids = *new std::vector<pthread_t>;
static void worker_main(int index)
{
    Worker self;
    self.index = index;
    t_w = &self;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_worker_ids.push_back(pthread_self());
    }
    {
        // faults are reported from an alternate stack: the faulting stack may be a lane's, and may be exhausted
        stack_t ss;
        ss.ss_sp = malloc(1 << 16);
        ss.ss_size = 1 << 16;
        ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
    }
    unsigned long long seen = 0;
    for (;;) {
        std::shared_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(g_mu);
            g_cv.wait(lk, [&] { return g_job && g_job_seq != seen; });
            job = g_job;
            seen = g_job_seq;
        }
        if (self.index >= job->active) continue;  // this launch does not fill the device: the "CU" of this worker holds none of it
        for (;;) {
            const unsigned b = job->next.fetch_add(1);
            if (b >= job->total) break;
            t_kernel = job->name;
            // EMU_STALL="<kernel name substring>:<workgroup>:<ms>" (tests): that workgroup starts late -- what a tenant
            // that takes a CU away between the census and the call does to a grid barrier
            {
                static const char *stall = getenv("EMU_STALL");
                if (stall) {
                    char pat[128];
                    unsigned blk = 0, ms = 0;
                    if (sscanf(stall, "%127[^:]:%u:%u", pat, &blk, &ms) == 3 && b == blk && strstr(job->name, pat))
                        std::this_thread::sleep_for(std::chrono::milliseconds(ms));
                }
            }
            run_block(&self, job->body, job->grid, job->block, b, job->lds);
            t_w = &self;
            if (job->finished.fetch_add(1) + 1 == job->total) {
                std::lock_guard<std::mutex> lk(g_mu);
                g_cv_done.notify_all();
            }
        }
    }
}

bool capture_record(hipStream_t st, std::function<void()> fn)
{
    if (!st || !st->capturing) return false;
    st->rec->push_back(std::move(fn));
    return true;
}

// kernels of different host threads (one per fake device in samples/amb_dist.cpp) run one after the other
static std::mutex &g_launch_mu = *new std::mutex;

// what hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) has allowed, per (function, device)
static std::map<std::pair<const void *, int>, int> &g_dyn_lds_allowed = *new std::map<std::pair<const void *, int>, int>;
static std::mutex &g_attr_mu = *new std::mutex;
int current_device();

void launch(const char *name, const void *func, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, std::function<void()> body)
{
    // the launch-time rules a real device enforces and a CPU would not: workgroup size, and dynamic LDS above 64 KiB only
    // for a function that was given the attribute ON THE CURRENT DEVICE (the runtime returns hipErrorInvalidValue)
    const unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || nt > 1024) {
        fprintf(stderr, "emu: kernel %s launched with %u work-items per workgroup\n", name, nt);
        abort();
    }
    if (lds_bytes > 160 * 1024) {
        fprintf(stderr, "emu: kernel %s asks for %zu bytes of dynamic LDS (a CU has 160 KiB)\n", name, lds_bytes);
        abort();
    }
    const KernelRes res = resources_of(func);
    if (res.lds + lds_bytes > kLdsPerCu) {
        fprintf(stderr, "emu: kernel %s needs %zu bytes of static + %zu bytes of dynamic LDS per workgroup (a CU has 160 KiB)\n", name,
                res.lds, lds_bytes);
        abort();
    }
    if (lds_bytes > 64 * 1024) {
        std::lock_guard<std::mutex> lk(g_attr_mu);
        auto it = g_dyn_lds_allowed.find({func, current_device()});
        if (it == g_dyn_lds_allowed.end() || (size_t)it->second < lds_bytes) {
            fprintf(stderr, "emu: kernel %s launched on device %d with %zu bytes of dynamic LDS without "
                            "hipFuncAttributeMaxDynamicSharedMemorySize >= that on THIS device (allowed: %d)\n",
                    name, current_device(), lds_bytes, it == g_dyn_lds_allowed.end() ? 65536 : it->second);
            abort();
        }
    }
    // EMU_COVERAGE=<file>: which kernel instantiations were launched (mangled names, appended at exit) -- compared with
    // the kernels the library contains by tools/emu_kernel_coverage.py
    {
        static const char *cov = getenv("EMU_COVERAGE");
        if (cov) {
            static std::set<const void *> &seen = *new std::set<const void *>;
            static std::mutex &mu = *new std::mutex;
            static bool hooked = false;
            std::lock_guard<std::mutex> lk(mu);
            seen.insert(func);
            if (!hooked) {
                hooked = true;
                atexit([] {
                    FILE *f = fopen(getenv("EMU_COVERAGE"), "a");
                    if (!f) return;
                    for (const void *p : seen) {
                        Dl_info di;
                        if (dladdr(p, &di) && di.dli_sname) fprintf(f, "%s\n", di.dli_sname);
                    }
                    fclose(f);
                });
            }
        }
    }
    if (stream && stream->capturing) {
        stream->rec->push_back([=] { launch(name, func, grid, block, lds_bytes, nullptr, body); });
        return;
    }
    std::lock_guard<std::mutex> launch_lk(g_launch_mu);
    install_trap();
    if (grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "emu: only 1-D grids are modelled\n");
        abort();
    }
    if (t_w) {
        fprintf(stderr, "emu: a kernel launched from a kernel\n");
        abort();
    }
    g_stats[0]++;
    static const bool trace = getenv("EMU_TRACE") != nullptr;
    if (trace) fprintf(stderr, "emu: %s <<<%u, %u, %zu>>>\n", name, grid.x, block.x * block.y * block.z, lds_bytes);
    if (grid.x == 0) return;
    auto job = std::make_shared<Job>();
    job->name = name;
    job->body = std::move(body);
    job->grid = grid;
    job->block = block;
    job->lds = lds_bytes;
    job->total = grid.x;
    {
        // resident workgroups of this launch (compute-unit model above)
        const size_t lds_wg = res.lds + lds_bytes;
        const int waves = (int)((nt + 63) / 64);
        long per_cu = 32 / waves;
        if (lds_wg > 0 && (long)(kLdsPerCu / lds_wg) < per_cu) per_cu = (long)(kLdsPerCu / lds_wg);
        if ((long)res.occ * 4 / waves < per_cu) per_cu = (long)res.occ * 4 / waves;
        if (per_cu < 1) per_cu = 1;
        const long resident = per_cu * cus();
        job->active = resident < workers() ? (int)resident : workers();
        static const bool trace_res = getenv("EMU_TRACE") != nullptr;
        if (trace_res) fprintf(stderr, "emu:   %zu B LDS per workgroup, %ld per CU x %d CUs -> %d resident\n", lds_wg, per_cu, cus(), job->active);
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_threads.empty())
            for (int i = 0; i < workers(); i++) g_threads.emplace_back(worker_main, i).detach();
        g_job = job;
        g_job_seq++;
    }
    g_cv.notify_all();
    std::unique_lock<std::mutex> lk(g_mu);
    // EMU_WATCHDOG_S (default 120): a kernel that runs longer has every worker say where it is (SIGUSR1: kernel,
    // workgroup, work-item, backtrace) and the process ends -- a lane spinning without a sync point cannot be
    // interrupted any other way
    static const double limit = getenv("EMU_WATCHDOG_S") ? atof(getenv("EMU_WATCHDOG_S")) : 120.0;
    if (!g_cv_done.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->finished.load() == job->total; })) {
        fprintf(stderr, "emu: kernel %s <<<%u, %u>>> has been running for %.0f s (%u of %u workgroups done)\n", name, grid.x,
                block.x * block.y * block.z, limit, job->finished.load(), job->total);
        for (pthread_t t : g_worker_ids) {
            pthread_kill(t, SIGUSR1);
            usleep(200000);
        }
        _exit(124);
    }
    g_job.reset();
}

}  // namespace emu

// ---------------------------------------------------------------------------------------------- fake runtime
namespace {
struct Ev {
    std::chrono::steady_clock::time_point t;
};
thread_local int t_device = 0;
}  // namespace
namespace emu {
int current_device() { return t_device; }
}

struct ihipEvent_t { std::chrono::steady_clock::time_point t; };
struct ihipGraph { std::shared_ptr<std::vector<std::function<void()>>> ops; };
struct ihipGraphExec { std::shared_ptr<std::vector<std::function<void()>>> ops; };
struct ihipMemPool { int x; };

// EMU_NO_GUARD=1: no guard bytes -- for the AddressSanitizer build of the emulation (make OUT=lib_asan EXTRA=-fsanitize=address
// ...), whose own redzones then sit right behind the block and catch an out-of-bounds READ as well
static const size_t kGuard = getenv("EMU_NO_GUARD") ? 0 : 256;
static std::mutex &g_blocks_mu = *new std::mutex;
static std::map<uintptr_t, size_t> &g_blocks = *new std::map<uintptr_t, size_t>;
extern "C" void emu_describe_address(const void *a)
{
    // (called from the fault handler: no lock -- the process is about to end)
    const uintptr_t x = (uintptr_t)a;
    auto it = g_blocks.upper_bound(x);
    char buf[256];
    int n = 0;
    if (it != g_blocks.begin()) {
        auto lo = std::prev(it);
        n += snprintf(buf + n, sizeof buf - n, "emu:   %zd bytes from the start of the device block of %zu bytes before it\n",
                      (ptrdiff_t)(x - lo->first), lo->second);
    }
    if (it != g_blocks.end())
        n += snprintf(buf + n, sizeof buf - n, "emu:   %zd bytes before the device block of %zu bytes behind it\n",
                      (ptrdiff_t)(it->first - x), it->second);
    (void)!write(2, buf, (size_t)n);
}
extern "C" {
// Every block carries kGuard bytes of 0xA5 on both sides, checked when it is freed (a kernel that WRITES past its
// array is reported with the block's size), and is registered so that a fault can be reported as "n bytes past block
// of size m" instead of a bare address.
hipError_t hipMalloc(void **p, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 256, n + 2 * kGuard) != 0) return hipErrorOutOfMemory;
    memset(q, 0xA5, n + 2 * kGuard);  // device memory is not zero
    *p = (unsigned char *)q + kGuard;
    std::lock_guard<std::mutex> lk(g_blocks_mu);
    g_blocks[(uintptr_t)*p] = n;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    size_t n;
    {
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        auto it = g_blocks.find((uintptr_t)p);
        if (it == g_blocks.end()) {
            fprintf(stderr, "emu: hipFree(%p) of something hipMalloc did not return\n", p);
            abort();
        }
        n = it->second;
        g_blocks.erase(it);
    }
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < kGuard; i++)
        if (b[-(ptrdiff_t)kGuard + (ptrdiff_t)i] != 0xA5 || b[n + i] != 0xA5) {
            fprintf(stderr, "emu: a kernel wrote outside a device block of %zu bytes (%s it, offset %zd)\n", n,
                    b[n + i] != 0xA5 ? "behind" : "before", b[n + i] != 0xA5 ? (ptrdiff_t)i : (ptrdiff_t)i - (ptrdiff_t)kGuard);
            abort();
        }
    free((unsigned char *)p - kGuard);
    return hipSuccess;
}
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFreeAsync(void *p, hipStream_t) { return hipFree(p); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)hipMemcpy(d, s, n, k); })) return hipSuccess;
    return hipMemcpy(d, s, n, k);
}
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st)
{
    if (emu::capture_record(st, [=] { (void)hipMemset(d, v, n); })) return hipSuccess;
    return hipMemset(d, v, n);
}
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
static int emu_devices()
{
    static const int n = [] {
        const char *e = getenv("EMU_DEVICES");
        const int v = e ? atoi(e) : 8;
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}
// EMU_DEVICES fake devices (default 8): all of them are this host's memory and the one worker pool -- what differs
// is the library's per-device state (contexts, block caches, communicators), which is what the tests are after
hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_devices()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int)
{
    *v = a == hipDeviceAttributeMultiprocessorCount ? emu::cus() : 0;
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t *p, int) { static ihipMemPool pool; *p = &pool; return hipSuccess; }
hipError_t hipMemPoolSetAttribute(hipMemPool_t, hipMemPoolAttr, void *) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// stream capture: what is queued on a capturing stream (kernels, async copies / fills, the collectives of the RCCL
// stand-in) is recorded as closures and replayed in order by hipGraphLaunch
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode)
{
    if (!s || s->capturing) return hipErrorInvalidValue;
    s->capturing = true;
    s->rec = std::make_shared<std::vector<std::function<void()>>>();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g)
{
    *g = nullptr;
    if (!s || !s->capturing) return hipErrorInvalidValue;
    s->capturing = false;
    *g = new ihipGraph{s->rec};
    s->rec.reset();
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t)
{
    if (!g) return hipErrorInvalidValue;
    *e = new ihipGraphExec{g->ops};
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t)
{
    if (!e) return hipErrorInvalidValue;
    for (auto &op : *e->ops) op();
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new ihipEvent_t{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNoDevice: return "no ROCm-capable device is detected";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorNotReady: return "device not ready";
    case hipErrorNotSupported: return "operation not supported (emulation)";
    default: return "unknown error";
    }
}
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v)
{
    if (a == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (v < 0 || v > 160 * 1024) return hipErrorInvalidValue;
        std::lock_guard<std::mutex> lk(emu::g_attr_mu);
        emu::g_dyn_lds_allowed[{f, t_device}] = v;
    }
    return hipSuccess;
}
void emu_get_stats(long long out[8])
{
    for (int i = 0; i < 8; i++) out[i] = emu::g_stats[i].load();
}
void emu_reset_stats(void)
{
    for (int i = 0; i < 8; i++) emu::g_stats[i] = 0;
}
// ---- B-entry fetch census (round 6): the product's heavy-row kernels report, per kernel family, how many column
// indices and values of B they loaded and how many products they accumulated (spgemm/common.h: NSP_COUNT, compiled in
// under NSP_EMU only).  Bytes, not clocks: "B entries fetched per product" is a property of the algorithm that a CPU
// can count.  16 families x {0: B.col loads, 1: B.val loads, 2: products accumulated, 3: tiles / rows (kernel's own)}.
static std::atomic<long long> g_fetch[16][4];
void nsp_emu_count(int family, int what, long long n)
{
    if ((unsigned)family < 16u && (unsigned)what < 4u) g_fetch[family][what].fetch_add(n, std::memory_order_relaxed);
}
// static LDS / occupancy of a kernel the table cannot know (selftest)
void emu_set_static_lds(const void *func, size_t bytes, int occ) { emu::set_static_lds(func, bytes, occ); }
void emu_get_fetch_counts(long long out[64], int reset)
{
    for (int f = 0; f < 16; f++)
        for (int w = 0; w < 4; w++) {
            out[f * 4 + w] = g_fetch[f][w].load();
            if (reset) g_fetch[f][w] = 0;
        }
}
}
