// lds_atomic_bench.hip -- what does an LDS floating-point atomic add cost on gfx950?
//
// The numeric dense-window kernel (csrc/spgemm/window.h: k_num_dense) performs ONE no-return
// ds_add_f64 per intermediate product, so its floor is set by the rate of that instruction, not by
// HBM.  This micro-benchmark measures that rate as lane-operations per clock per CU for
//   * ds_add_f64 and ds_add_f32 (no return value),
//   * 64 / 48 / 32 / 24 / 16 / 8 active lanes per wave-instruction,
//   * address patterns: consecutive slots (conflict-free), a random slot in a 1536-slot window
//     (what a FEM row looks like), 2-, 4- and 8-way bank conflicts, all lanes one address.
// Every CU holds 32 waves (4 workgroups of 512 threads), each wave issues ITER * 16 atomics between
// two s_memtime reads; rate = active lanes * instructions / mean wave cycles * 32 waves per CU.
// Output: one JSON object on stdout (bench.py reads profiles/r02_lds_atomic.json made from it).
//
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic_bench.hip -o lds_atomic_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int BS = 512, ITER = 256, UNROLL = 16, SLOTS = 2048;

template <typename T>
__global__ __launch_bounds__(BS) void k_atomic(const int *__restrict__ addr, int active, unsigned long long *cycles,
                                               T *sink)
{
    __shared__ T acc[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += BS) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // 16 precomputed slots per lane, so that address arithmetic is not part of the loop
    int a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) a[u] = addr[(threadIdx.x & 63) * UNROLL + u];
    const T v = (T)(1 + lane);
    const bool on = lane < active;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (on) {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) unsafeAtomicAdd(acc + a[u], v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * (BS / 64) + (threadIdx.x >> 6)] = t1 - t0;
    __syncthreads();
    if (threadIdx.x < 8) sink[blockIdx.x * 8 + threadIdx.x] = acc[threadIdx.x * 7];
}

struct Pattern {
    const char *name;
    int kind;  // 0 consecutive, 1 random in window, 2.. k-way conflict, -1 same address
};

template <typename T>
static void run(const char *tname, int cus, bool first)
{
    const Pattern pats[] = {{"consecutive", 0}, {"random_1536", 1}, {"conflict_2way", 2}, {"conflict_4way", 4},
                            {"conflict_8way", 8}, {"same_address", -1}};
    const int lanes[] = {64, 48, 32, 24, 16, 8};
    const int grid = cus * 4;
    int *d_addr;
    unsigned long long *d_cyc;
    T *d_sink;
    CHECK(hipMalloc(&d_addr, sizeof(int) * 64 * UNROLL));
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * grid * (BS / 64)));
    CHECK(hipMalloc(&d_sink, sizeof(T) * grid * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    bool first_row = first;
    for (const Pattern &p : pats) {
        std::vector<int> h(64 * UNROLL);
        unsigned long long s = 0x9E3779B97F4A7C15ull;
        for (int l = 0; l < 64; l++)
            for (int u = 0; u < UNROLL; u++) {
                int v;
                if (p.kind == 0) v = (l + 64 * u) % SLOTS;
                else if (p.kind == 1) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (int)((s >> 33) % 1536); }
                else if (p.kind == -1) v = 5 * u;
                else {
                    // k-way: slots 256 bytes apart share their bank(s); inside every 32-lane half k
                    // lanes do, with different addresses
                    const int g = l & 31, grp = l >> 5;
                    const int period = 256 / (int)sizeof(T);
                    v = (g / p.kind) + (g % p.kind) * period + grp * (32 / p.kind);
                    v = (v + 8 * period * (u & 1)) % SLOTS;
                }
                h[l * UNROLL + u] = v;
            }
        CHECK(hipMemcpy(d_addr, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
        for (int act : lanes) {
            hipLaunchKernelGGL(k_atomic<T>, dim3(grid), dim3(BS), 0, 0, d_addr, act, d_cyc, d_sink);  // warm-up
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_atomic<T>, dim3(grid), dim3(BS), 0, 0, d_addr, act, d_cyc, d_sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> cyc((size_t)grid * (BS / 64));
            CHECK(hipMemcpy(cyc.data(), d_cyc, sizeof(unsigned long long) * cyc.size(), hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto c : cyc) mean += (double)c;
            mean /= (double)cyc.size();
            const double inst = (double)ITER * UNROLL;
            const double lanes_per_clk_cu = (double)act * inst * 32.0 / mean;     // 32 waves per CU
            const double cyc_per_inst_cu = mean / (inst * 32.0);                  // CU cycles per wave-instruction
            const double glaneops = (double)act * inst * (double)grid * (BS / 64) / (ms * 1e-3) / 1e9;
            printf("%s\n  {\"type\": \"%s\", \"pattern\": \"%s\", \"active_lanes\": %d, \"lanes_per_clk_per_cu\": %.3f, "
                   "\"cu_cycles_per_wave_instruction\": %.2f, \"kernel_ms\": %.4f, \"chip_glaneops_per_s\": %.1f}",
                   first_row ? "" : ",", tname, p.name, act, lanes_per_clk_cu, cyc_per_inst_cu, ms, glaneops);
            first_row = false;
        }
    }
    CHECK(hipFree(d_addr));
    CHECK(hipFree(d_cyc));
    CHECK(hipFree(d_sink));
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_cu\": 32, \"instructions_per_wave\": %d,\n"
           " \"note\": \"no-return LDS atomic add; lanes_per_clk_per_cu = active lanes * instructions * 32 waves / mean wave cycles (s_memtime)\",\n"
           " \"rows\": [",
           prop.name, cus, prop.clockRate / 1000, ITER * UNROLL);
    run<double>("ds_add_f64", cus, true);
    run<float>("ds_add_f32", cus, false);
    printf("\n]}\n");
    return 0;
}
