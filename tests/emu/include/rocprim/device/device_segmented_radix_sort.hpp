// tests/emu: host stand-ins for the rocPRIM segmented sorts the library calls (stable, keys compared on bits
// [begin_bit, end_bit) only, like a radix sort).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>
namespace rocprim {
template <typename K>
inline unsigned long long emu_key_bits(K k, unsigned b0, unsigned b1)
{
    static_assert(std::is_integral<K>::value, "integer keys");
    using U = typename std::make_unsigned<K>::type;
    unsigned long long u = (unsigned long long)(U)k;
    if (std::is_signed<K>::value) u ^= 1ull << (8 * sizeof(K) - 1);  // radix order of signed keys
    const unsigned w = b1 - b0;
    return w >= 64 ? u >> b0 : (u >> b0) & ((1ull << w) - 1ull);
}
template <typename K, typename V, typename Off>
inline hipError_t segmented_radix_sort_pairs(void *tmp, size_t &tmp_bytes, const K *kin, K *kout, const V *vin, V *vout,
                                             unsigned size, unsigned segments, Off begin, Off end, unsigned b0 = 0,
                                             unsigned b1 = 8 * sizeof(K), hipStream_t = nullptr, bool = false)
{
    if (tmp == nullptr) {
        tmp_bytes = 16;
        return hipSuccess;
    }
    std::vector<unsigned> idx;
    for (unsigned s = 0; s < segments; s++) {
        const long long lo = begin[s], hi = end[s];
        idx.resize((size_t)(hi - lo));
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) {
            return emu_key_bits(kin[lo + a], b0, b1) < emu_key_bits(kin[lo + b], b0, b1);
        });
        for (size_t i = 0; i < idx.size(); i++) {
            kout[lo + i] = kin[lo + idx[i]];
            vout[lo + i] = vin[lo + idx[i]];
        }
    }
    return hipSuccess;
}
template <typename K, typename Off>
inline hipError_t segmented_radix_sort_keys_desc(void *tmp, size_t &tmp_bytes, const K *kin, K *kout, unsigned size,
                                                 unsigned segments, Off begin, Off end, unsigned b0 = 0,
                                                 unsigned b1 = 8 * sizeof(K), hipStream_t = nullptr, bool = false)
{
    if (tmp == nullptr) {
        tmp_bytes = 16;
        return hipSuccess;
    }
    std::vector<K> seg;
    for (unsigned s = 0; s < segments; s++) {
        const long long lo = begin[s], hi = end[s];
        seg.assign(kin + lo, kin + hi);
        std::stable_sort(seg.begin(), seg.end(), [&](K a, K b) { return emu_key_bits(a, b0, b1) > emu_key_bits(b, b0, b1); });
        std::copy(seg.begin(), seg.end(), kout + lo);
    }
    return hipSuccess;
}
}  // namespace rocprim
