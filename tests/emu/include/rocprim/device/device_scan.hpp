// tests/emu: host stand-in for the one rocPRIM scan the library calls (same two-phase calling convention).
#pragma once
#include <hip/hip_runtime.h>
namespace rocprim {
template <typename T>
struct plus {
    T operator()(const T &a, const T &b) const { return a + b; }
};
template <typename In, typename Out, typename Init, typename Op>
inline hipError_t exclusive_scan(void *tmp, size_t &tmp_bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr,
                                 bool = false)
{
    if (tmp == nullptr) {
        tmp_bytes = 16;
        return hipSuccess;
    }
    auto acc = (typename std::remove_reference<decltype(out[0])>::type)init;
    for (size_t i = 0; i < n; i++) {
        const auto v = in[i];  // (in may alias out)
        out[i] = acc;
        acc = op(acc, v);
    }
    return hipSuccess;
}
}  // namespace rocprim
