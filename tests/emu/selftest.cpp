// tests/emu/selftest.cpp -- the emulator checked against what the INSTRUCTIONS are documented to do, on kernels small
// enough to verify by hand (run by tests/test_emu_cpu.py).  Exit code 0 = every check passed.
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <vector>

static int g_fail = 0;
#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);        \
            g_fail++;                                                 \
        }                                                             \
    } while (0)

__global__ void k_ids(int *out)
{
    out[blockIdx.x * blockDim.x + threadIdx.x] = blockIdx.x * 1000 + threadIdx.x;
}

__global__ void k_wave(int *out, unsigned long long *bal)
{
    const int lane = threadIdx.x & 63;
    int v = lane + 1;
    // inclusive scan by DPP, the way spgemm/common.h does it
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);
    out[threadIdx.x] = s;
    int x = v;
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    out[256 + threadIdx.x] = x;
    bal[threadIdx.x] = __ballot(lane % 3 == 0);
    out[512 + threadIdx.x] = __shfl(v, 5) + __shfl_up(v, 1) * 100 + __shfl_down(v, 2) * 10000;
    out[768 + threadIdx.x] = __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f) + 1000 * __builtin_amdgcn_ds_bpermute((63 - lane) << 2, v);
    out[1024 + threadIdx.x] = __builtin_amdgcn_readlane(v, 63) + __builtin_amdgcn_readfirstlane(v);
    // divergent: only odd lanes vote
    if (lane & 1) out[1280 + threadIdx.x] = (int)__popcll(__ballot(1));
    else out[1280 + threadIdx.x] = -1;
    // sub-wave shuffles (width 4)
    out[1536 + threadIdx.x] = __shfl(v, 0, 4) + 100 * __shfl_xor(v, 1, 4);
}

__global__ void k_block(int *out)
{
    __shared__ int s[256];
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const int v = s[255 - threadIdx.x];
    atomicAdd(&total, v);
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = total + v;
    atomicAdd(out + 4096, 1);
}

__global__ void k_dyn(int *out, int n)
{

    int *d = reinterpret_cast<int *>(::emu::t_dyn_lds);
    for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = i * 2;
    __syncthreads();
    int acc = 0;
    for (int i = 0; i < n; i++) acc += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// a grid barrier: every workgroup waits for all others (needs them resident together)
__global__ void k_grid(int *cnt, int *out)
{
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = *cnt;
}

__global__ void k_dppmm(int *out)
{
    const int lane = threadIdx.x & 63;
    int x = (lane * 37) & 63;
    const int lo = emu_dpp_asm(false, 0x7fffdead, x, x, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    const int hi = emu_dpp_asm(true, 0x7fffdead, x, x, "row_mirror row_mask:0xf bank_mask:0xf");
    int y = x;
    y = emu_dpp_asm(false, y, 5, y, "quad_perm:[0,1,2,3] row_mask:0x5 bank_mask:0xf");  // rows 0 and 2: min(5, y)
    out[threadIdx.x] = lo;
    out[64 + threadIdx.x] = hi;
    out[128 + threadIdx.x] = y;
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-7, x, 0x138, 0xf, 0xf, false);  // wave_shr:1, lane 0 keeps old
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(-7, x, 0x111, 0xf, 0xf, true);   // row_shr:1 bound_ctrl: 0 at row starts
}

// counts how many workgroups of the launch are inside the kernel at the same time (what fused.h: k_census does)
__global__ void k_resident(int *now, int *peak)
{
    if (threadIdx.x == 0) {
        const int n = atomicAdd(now, 1) + 1;
        atomicMax(peak, n);
        for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(8);  // stay a while: the others must overlap if they can
        atomicAdd(now, -1);
    }
}

int main()
{
    int *d;
    unsigned long long *b;
    hipMalloc((void **)&d, sizeof(int) * 8192);
    hipMalloc((void **)&b, sizeof(unsigned long long) * 256);
    hipLaunchKernelGGL(k_ids, dim3(5), dim3(192), 0, nullptr, d);
    for (int i = 0; i < 5 * 192; i++) CHECK(d[i] == (i / 192) * 1000 + i % 192);

    hipLaunchKernelGGL(k_wave, dim3(1), dim3(256), 0, nullptr, d, b);
    for (int t = 0; t < 256; t++) {
        const int l = t & 63, v = l + 1;
        CHECK(d[t] == v * (v + 1) / 2);
        CHECK(d[256 + t] == 64 * 65 / 2);
        unsigned long long m = 0;
        for (int j = 0; j < 64; j++) m |= (unsigned long long)(j % 3 == 0) << j;
        CHECK(b[t] == m);
        const int up = l >= 1 ? v - 1 : v, down = l + 2 < 64 ? v + 2 : v;
        CHECK(d[512 + t] == 6 + up * 100 + down * 10000);
        CHECK(d[768 + t] == ((l ^ 4) + 1) + 1000 * (64 - l));
        CHECK(d[1024 + t] == 64 + 1);
        CHECK(d[1280 + t] == ((l & 1) ? 32 : -1));
        CHECK(d[1536 + t] == ((l & ~3) + 1) + 100 * ((l ^ 1) + 1));
    }
    memset(d, 0, sizeof(int) * 8192);
    hipLaunchKernelGGL(k_block, dim3(16), dim3(256), 0, nullptr, d);
    for (int i = 0; i < 4096; i++) CHECK(d[i] == 255 * 256 / 2 + (255 - (i & 255)));
    CHECK(d[4096] == 4096);

    hipLaunchKernelGGL(k_dyn, dim3(3), dim3(128), 4000, nullptr, d, 1000);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_dyn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k_dyn, dim3(1), dim3(128), 100 * 1024, nullptr, d, 1000);  // above 64 KiB: needs the attribute
    for (int i = 0; i < 3 * 128; i++) CHECK(d[i] == 999 * 1000);

    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    memset(d, 0, sizeof(int) * 8192);
    hipLaunchKernelGGL(k_grid, dim3(ncu), dim3(1024), 0, nullptr, d + 8000, d);
    for (int i = 0; i < ncu * 1024; i++) CHECK(d[i] == ncu);

    hipLaunchKernelGGL(k_dppmm, dim3(1), dim3(64), 0, nullptr, d);
    for (int l = 0; l < 64; l++) {
        auto X = [](int q) { return (q * 37) & 63; };
        CHECK(d[l] == std::min(X(l ^ 1), X(l)));
        CHECK(d[64 + l] == std::max(X((l & ~15) + 15 - (l & 15)), X(l)));
        CHECK(d[128 + l] == (((l >> 4) & 1) == 0 ? std::min(5, X(l)) : X(l)));
        CHECK(d[192 + l] == (l == 0 ? -7 : X(l - 1)));
        CHECK(d[256 + l] == ((l & 15) == 0 ? 0 : X(l - 1)));
    }
    // ---- LDS capacity and co-residency (round 6) --------------------------------------------------------------------
    // The pool stands for `ncu` compute units of 160 KiB: a kernel that takes more than half a CU's LDS runs one workgroup
    // per CU, one that takes a quarter runs up to four (never more than there are workers), and static + dynamic LDS above
    // 160 KiB is refused at launch.  Static LDS comes from the real compiler's table for the product's kernels; here it is
    // declared by hand.
    {
        auto peak_of = [&](size_t stat, size_t dyn) {
            emu_set_static_lds(reinterpret_cast<const void *>(k_resident), stat, 8);
            memset(d, 0, sizeof(int) * 2);
            hipLaunchKernelGGL(k_resident, dim3(64), dim3(64), dyn, nullptr, d, d + 1);
            CHECK(d[0] == 0);
            return d[1];
        };
        const int p_big = peak_of(90 * 1024, 6 * 1024);     // 96 KiB: one per CU
        const int p_quarter = peak_of(30 * 1024, 10 * 1024);  // 40 KiB: four per CU
        CHECK(p_big >= 1 && p_big <= ncu);
        CHECK(p_quarter <= 4 * ncu && p_quarter >= p_big);
        printf("selftest: %d CUs; resident workgroups of a 96 KiB kernel %d, of a 40 KiB kernel %d\n", ncu, p_big, p_quarter);
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) {  // 100 KiB static + 61 KiB dynamic = 161 KiB: must be refused (abort) before anything runs
            fclose(stderr);
            emu_set_static_lds(reinterpret_cast<const void *>(k_resident), 100 * 1024, 8);
            hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 61 * 1024, nullptr, d, d + 1);
            _exit(0);
        }
        int status = 0;
        waitpid(pid, &status, 0);
        CHECK(WIFSIGNALED(status) && WTERMSIG(status) == SIGABRT);
    }
    long long st[8];
    emu_get_stats(st);
    printf("selftest: %d failures; launches %lld, divergent instructions %lld, inactive reads %lld, workgroups %lld\n", g_fail,
           st[0], st[1], st[2], st[3]);
    return g_fail ? 1 : 0;
}
