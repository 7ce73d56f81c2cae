// tests/emu: a lane-by-lane CPU EMULATION of the HIP device model, enough to run the kernels of nsparse_amd/csrc
// unchanged on a box without a GPU.  TEST INFRASTRUCTURE -- not a product path, not a fallback: the libraries it
// builds live in tests/emu/lib, nothing under nsparse_amd/ loads them, and they exist to check the LOGIC of the
// wave-level code (lane maps, DPP / swizzle / bpermute patterns, LDS index arithmetic, barrier placement) when no
// device is available.  What it cannot see: timing, occupancy, hazards between instructions, the memory model
// (every access is sequentially consistent here), LDS capacity.
//
// Model: one fiber per work-item.  A wave is 64 consecutive work-items of a workgroup.  A cross-lane operation
// (__shfl*, __ballot, DPP, ds_bpermute, ds_swizzle, readlane, ...) parks the lane until every lane of its wave that
// is still running has parked; the lanes parked at the SAME call site then form the execution mask of that
// instruction (lowest call-site address first when the wave has diverged), the operation is evaluated over them
// with the hardware's rules for inactive / out-of-row source lanes, and they continue.  __syncthreads() parks a
// lane until all running lanes of the workgroup are parked at a barrier.  Workgroups of a launch run one after the
// other -- or all resident together, round-robin at s_sleep, when the launch is small (grid barriers).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

// ---------------------------------------------------------------------------------------------- host runtime API
typedef enum hipError_t {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100,
    hipErrorInvalidDevice = 101, hipErrorNotReady = 600, hipErrorNotSupported = 801, hipErrorUnknown = 999
} hipError_t;
struct ihipStream_t;
struct ihipEvent_t;
struct ihipGraph;
struct ihipGraphExec;
struct ihipMemPool;
typedef ihipStream_t *hipStream_t;
typedef ihipEvent_t *hipEvent_t;
typedef ihipGraph *hipGraph_t;
typedef ihipGraphExec *hipGraphExec_t;
typedef ihipMemPool *hipMemPool_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemPoolAttr { hipMemPoolAttrReleaseThreshold = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2,
                   hipHostMallocCoherent = 0x40000000;
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern "C" {
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t);
hipError_t hipFreeAsync(void *p, hipStream_t);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned flags);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipDeviceSynchronize(void);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int dev);
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi);
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t *p, int dev);
hipError_t hipMemPoolSetAttribute(hipMemPool_t p, hipMemPoolAttr a, void *v);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g);
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
// emulator statistics (tests): [0] kernel launches, [1] wave instructions over a partial wave whose other lanes
// waited at a DIFFERENT call site (divergence resolved by the lowest-address rule), [2] reads of an inactive lane
// through bpermute / swizzle / shfl (0 on the hardware), [3] workgroups run, [4] DPP reads of an invalid lane
void emu_get_stats(long long out[8]);
void emu_reset_stats(void);
void emu_set_static_lds(const void *func, size_t bytes, int occ);  // what <library>.lds says for the product's kernels
void emu_get_fetch_counts(long long out[64], int reset);
}

// ---------------------------------------------------------------------------------------------- device side
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...)          /* __attribute__((amdgpu_waves_per_eu(..))) -> __attribute__(()) */
#define amdgpu_flat_work_group_size(...)
#ifdef EMU_SHARED_STATIC
// LDS as plain statics: AddressSanitizer puts redzones around globals (not around thread-locals), so an out-of-bounds
// LDS index is reported -- at the price of ONE workgroup at a time (EMU_WORKERS is forced to 1: no grid barriers)
#define __shared__ static
#else
#define __shared__ static thread_local
#endif
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace emu {
struct Idx {
    unsigned x, y, z;
};
// the coordinates of the lane that is running (set by the scheduler on every switch)
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local unsigned char *t_dyn_lds;  // dynamic LDS of the running workgroup

enum Op {
    OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_READLANE, OP_READFIRST, OP_DPP, OP_BPERMUTE, OP_SWIZZLE,
    OP_PERMLANE32_SWAP, OP_WAVE_BARRIER, OP_DPP_MIN, OP_DPP_MAX
};
struct Req {
    int op;
    unsigned long long a, b;  // payloads (a: value / old, b: second value)
    int c, d, e, f;           // immediates: lane / delta / mask, width, ...
};
struct Res {
    unsigned long long r0, r1;
};
// parks the calling lane until the instruction has been evaluated over its wave
Res collective(const Req &rq) __attribute__((noinline));  // the call site = its return address
void block_barrier() __attribute__((noinline));
void yield_lane() __attribute__((noinline));
void launch(const char *name, const void *func, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, std::function<void()> body);
bool capture_record(hipStream_t st, std::function<void()> fn);  // true: `st` is being captured, fn was recorded instead of run

template <typename T>
static inline unsigned long long bits_of(T v)
{
    static_assert(sizeof(T) <= 8, "cross-lane payloads are at most 64 bits");
    unsigned long long u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <typename T>
static inline T from_bits(unsigned long long u)
{
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
}  // namespace emu

#define threadIdx (::emu::t_threadIdx)
#define blockIdx (::emu::t_blockIdx)
#define blockDim (::emu::t_blockDim)
#define gridDim (::emu::t_gridDim)

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ::emu::launch(#kernel, reinterpret_cast<const void *>(+(kernel)), (grid), (block), (size_t)(lds), (hipStream_t)(stream), [=]() { (kernel)(__VA_ARGS__); })

static __forceinline__ void __syncthreads() { ::emu::block_barrier(); }
static __forceinline__ void __threadfence() {}
static __forceinline__ void __threadfence_block() {}
static __forceinline__ void __threadfence_system() {}

static __forceinline__ unsigned long long __ballot(int pred)
{
    return ::emu::collective({::emu::OP_BALLOT, (unsigned long long)(pred != 0), 0, 0, 0, 0, 0}).r0;
}
static __forceinline__ int __any(int pred) { return __ballot(pred) != 0; }
static __forceinline__ int __all(int pred) { return __ballot(!pred) == 0; }
template <typename T>
static __forceinline__ T __shfl(T v, int src, int width = 64)
{
    return ::emu::from_bits<T>(::emu::collective({::emu::OP_SHFL, ::emu::bits_of(v), 0, src, width, 0, 0}).r0);
}
template <typename T>
static __forceinline__ T __shfl_up(T v, unsigned delta, int width = 64)
{
    return ::emu::from_bits<T>(::emu::collective({::emu::OP_SHFL_UP, ::emu::bits_of(v), 0, (int)delta, width, 0, 0}).r0);
}
template <typename T>
static __forceinline__ T __shfl_down(T v, unsigned delta, int width = 64)
{
    return ::emu::from_bits<T>(::emu::collective({::emu::OP_SHFL_DOWN, ::emu::bits_of(v), 0, (int)delta, width, 0, 0}).r0);
}
template <typename T>
static __forceinline__ T __shfl_xor(T v, int mask, int width = 64)
{
    return ::emu::from_bits<T>(::emu::collective({::emu::OP_SHFL_XOR, ::emu::bits_of(v), 0, mask, width, 0, 0}).r0);
}
static __forceinline__ int __builtin_amdgcn_readlane(int v, int lane)
{
    return (int)::emu::collective({::emu::OP_READLANE, (unsigned)v, 0, lane, 0, 0, 0}).r0;
}
static __forceinline__ int __builtin_amdgcn_readfirstlane(int v)
{
    return (int)::emu::collective({::emu::OP_READFIRST, (unsigned)v, 0, 0, 0, 0, 0}).r0;
}
static __forceinline__ int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    return (int)::emu::collective({::emu::OP_DPP, (unsigned)old, (unsigned)src, ctrl, row_mask, bank_mask, bound_ctrl}).r0;
}
static __forceinline__ int __builtin_amdgcn_ds_bpermute(int addr, int v)
{
    return (int)::emu::collective({::emu::OP_BPERMUTE, (unsigned)addr, (unsigned)v, 0, 0, 0, 0}).r0;
}
static __forceinline__ int __builtin_amdgcn_ds_swizzle(int v, int pattern)
{
    return (int)::emu::collective({::emu::OP_SWIZZLE, (unsigned)v, 0, pattern, 0, 0, 0}).r0;
}
struct emu_uint2v {
    unsigned int v[2];
    unsigned int operator[](int i) const { return v[i]; }
};
static __forceinline__ emu_uint2v __builtin_amdgcn_permlane32_swap(unsigned old, unsigned src, bool fi, bool bc)
{
    const ::emu::Res r = ::emu::collective({::emu::OP_PERMLANE32_SWAP, old, src, 0, 0, 0, 0});
    return emu_uint2v{{(unsigned)r.r0, (unsigned)r.r1}};
}
// v_min_i32_dpp / v_max_i32_dpp dst, src0(dpp), src1: dst = min(dpp(src0), src1) on the lanes the row / bank masks
// enable and whose DPP source is valid, else dst keeps `old` (tests/emu hook of the inline-asm sorts in spgemm/lean.h)
static __forceinline__ int emu_dpp_minmax(bool is_max, int old, int src0, int src1, int ctrl, int row_mask, int bank_mask)
{
    return (int)::emu::collective({is_max ? ::emu::OP_DPP_MAX : ::emu::OP_DPP_MIN, (unsigned)old,
                                               ((unsigned long long)(unsigned)src0 << 32) | (unsigned)src1, ctrl, row_mask, bank_mask, 0}).r0;
}
// the operand text of a DPP instruction as the inline asm of spgemm/lean.h spells it ("row_mirror row_mask:0x5
// bank_mask:0xf", "quad_perm:[1,0,3,2] ...") -> dpp_ctrl, row mask, bank mask
static inline void emu_parse_dpp(const char *s, int &ctrl, int &rm, int &bm)
{
    ctrl = -1, rm = 0xf, bm = 0xf;
    auto num = [](const char *p) { return (int)strtol(p, nullptr, 0); };
    for (const char *p = s; *p;) {
        while (*p == ' ') p++;
        if (!*p) break;
        const char *e = p;
        while (*e && *e != ' ') e++;
        const size_t n = (size_t)(e - p);
        auto is = [&](const char *k) { return strncmp(p, k, strlen(k)) == 0; };
        if (is("quad_perm:[")) {
            int q[4];
            if (sscanf(p, "quad_perm:[%d,%d,%d,%d]", &q[0], &q[1], &q[2], &q[3]) != 4) abort();
            ctrl = q[0] | (q[1] << 2) | (q[2] << 4) | (q[3] << 6);
        } else if (is("row_shl:")) ctrl = 0x100 + num(p + 8);
        else if (is("row_shr:")) ctrl = 0x110 + num(p + 8);
        else if (is("row_ror:")) ctrl = 0x120 + num(p + 8);
        else if (is("wave_shl:1")) ctrl = 0x130;
        else if (is("wave_rol:1")) ctrl = 0x134;
        else if (is("wave_shr:1")) ctrl = 0x138;
        else if (is("wave_ror:1")) ctrl = 0x13c;
        else if (is("row_mirror")) ctrl = 0x140;
        else if (is("row_half_mirror")) ctrl = 0x141;
        else if (is("row_bcast:15")) ctrl = 0x142;
        else if (is("row_bcast:31")) ctrl = 0x143;
        else if (is("row_mask:")) rm = num(p + 9);
        else if (is("bank_mask:")) bm = num(p + 10);
        else {
            fprintf(stderr, "emu: DPP operand '%.*s' is not modelled\n", (int)n, p);
            abort();
        }
        p = e;
    }
    if (ctrl < 0) {
        fprintf(stderr, "emu: no DPP pattern in '%s'\n", s);
        abort();
    }
}
static __forceinline__ int emu_dpp_asm(bool is_max, int old, int src0, int src1, const char *spec)
{
    int ctrl, rm, bm;
    emu_parse_dpp(spec, ctrl, rm, bm);
    return emu_dpp_minmax(is_max, old, src0, src1, ctrl, rm, bm);
}
static __forceinline__ void __builtin_amdgcn_wave_barrier()
{
    ::emu::collective({::emu::OP_WAVE_BARRIER, 0, 0, 0, 0, 0, 0});
}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static __forceinline__ void __builtin_amdgcn_s_sleep(int) { ::emu::yield_lane(); }
static __forceinline__ unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned width)
{
    off &= 31, width &= 31;
    return width == 0 ? 0u : (v >> off) & ((1u << width) - 1u);
}
static __forceinline__ float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
// ~100 MHz ticks of real time (EMU_CLOCK_DIV = host cycles per tick, default 20).  A larger divisor slows the device's
// clock down: the bounded waits of the census and of the grid barriers (0.2 ms / 50 ms of it) then survive a host whose
// cores are shared with other work -- tools/emu_corpus.sh runs with 2000
namespace emu {
unsigned long long clock_div();
}
static __forceinline__ unsigned long long wall_clock64()
{
    return (unsigned long long)__builtin_readcyclecounter() / ::emu::clock_div();
}
static __forceinline__ long long clock64() { return (long long)__builtin_readcyclecounter(); }

static __forceinline__ unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static __forceinline__ int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
static __forceinline__ int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static __forceinline__ int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
static __forceinline__ int __popc(unsigned v) { return __builtin_popcount(v); }
static __forceinline__ int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static __forceinline__ int __ffs(int v) { return __builtin_ffs(v); }
static __forceinline__ int __ffsll(long long v) { return __builtin_ffsll(v); }

// atomics: workgroups run on their own OS threads, so these are real (a CAS loop on the bit pattern)
template <typename T, typename F>
static __forceinline__ T emu_rmw(T *p, F f)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- and 64-bit atomics");
    using U = typename std::conditional<sizeof(T) == 4, unsigned int, unsigned long long>::type;
    U *up = reinterpret_cast<U *>(p);
    U old = __atomic_load_n(up, __ATOMIC_RELAXED);
    for (;;) {
        T o;
        memcpy(&o, &old, sizeof(T));
        const T n = f(o);
        U nb;
        memcpy(&nb, &n, sizeof(T));
        if (__atomic_compare_exchange_n(up, &old, nb, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return o;
    }
}
#define EMU_ATOMIC_RMW(name, expr)                                              \
    template <typename T, typename U>                                           \
    static __forceinline__ T name(T *p, U v_)                                   \
    {                                                                           \
        const T v = (T)v_;                                                      \
        return emu_rmw(p, [v](T old) -> T { return (T)(expr); });               \
    }
EMU_ATOMIC_RMW(atomicAdd, old + v)
EMU_ATOMIC_RMW(unsafeAtomicAdd, old + v)
EMU_ATOMIC_RMW(atomicSub, old - v)
EMU_ATOMIC_RMW(atomicOr, old | v)
EMU_ATOMIC_RMW(atomicAnd, old & v)
EMU_ATOMIC_RMW(atomicXor, old ^ v)
EMU_ATOMIC_RMW(atomicMax, old > v ? old : v)
EMU_ATOMIC_RMW(atomicMin, old < v ? old : v)
EMU_ATOMIC_RMW(atomicExch, v)
template <typename T, typename U, typename W>
static __forceinline__ T atomicCAS(T *p, U cmp, W val)
{
    const T c = (T)cmp, v = (T)val;
    return emu_rmw(p, [c, v](T old) -> T { return old == c ? v : old; });
}
