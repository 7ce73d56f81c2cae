// PMC calibration: known-size streams in the access widths the library uses, so that
// FETCH_SIZE / WRITE_SIZE can be turned into bytes (MI355X_MICROARCH.md, HBM section: the
// counters are only calibrated for 16 B/lane reads; calibrate your own pattern).
// Each kernel moves exactly BYTES bytes (> 256 MiB Infinity Cache) once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr size_t BYTES = 1ull << 30;
template <typename T> __global__ void read_stream(const T* __restrict__ p, size_t n, T* sink) {
    T acc{}; size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { T v = __builtin_nontemporal_load(p + i); acc = acc + v; }
    if (acc == T(123456789)) *sink = acc;
}
__global__ void read_stream16(const double2* __restrict__ p, size_t n, double* sink) {
    double acc = 0; size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 123456789.0) *sink = acc;
}
template <typename T> __global__ void write_stream(T* __restrict__ p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = T(1);
}
int main() {
    char *a, *b; hipMalloc(&a, BYTES); hipMalloc(&b, BYTES); hipMemset(a, 1, BYTES); hipMemset(b, 0, BYTES);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read_stream16, dim3(4096), dim3(256), 0, 0, (const double2*)a, BYTES / 16, (double*)b);
        hipLaunchKernelGGL(read_stream<double>, dim3(4096), dim3(256), 0, 0, (const double*)a, BYTES / 8, (double*)b);
        hipLaunchKernelGGL(read_stream<int>, dim3(4096), dim3(256), 0, 0, (const int*)a, BYTES / 4, (int*)b);
        hipLaunchKernelGGL(read_stream<unsigned short>, dim3(4096), dim3(256), 0, 0, (const unsigned short*)a, BYTES / 2, (unsigned short*)b);
        hipLaunchKernelGGL(write_stream<double>, dim3(4096), dim3(256), 0, 0, (double*)b, BYTES / 8);
        hipLaunchKernelGGL(write_stream<int>, dim3(4096), dim3(256), 0, 0, (int*)b, BYTES / 4);
    }
    hipDeviceSynchronize();
    printf("moved %zu bytes per kernel\n", BYTES);
    return 0;
}
