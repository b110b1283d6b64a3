// TEST INFRASTRUCTURE ONLY (oracle): the two CUB entry points the reference calls (rasterizer_impl.cu:165,187,285,311),
// by their documented contract -- InclusiveSum = inclusive prefix sum; SortPairs = STABLE ascending sort of (key, value)
// pairs on key bits [begin_bit, end_bit) (CUB's LSD radix sort is stable).  d_temp_storage == nullptr => size query.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "../cuda_runtime.h"
namespace cub {
struct DeviceScan {
    template <typename In, typename Out>
    static cudaError_t InclusiveSum(void* tmp, size_t& tmp_bytes, In in, Out out, int n) {
        if (!tmp) { tmp_bytes = 256; return cudaSuccess; }
        std::partial_sum(in, in + n, out);
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    template <typename K, typename V>
    static cudaError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, int n,
                                 int begin_bit = 0, int end_bit = sizeof(K) * 8) {
        if (!tmp) { tmp_bytes = 256; return cudaSuccess; }
        const int bits = end_bit - begin_bit;
        const K m = bits >= (int)sizeof(K) * 8 ? ~K(0) : ((K(1) << bits) - 1);
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ((kin[a] >> begin_bit) & m) < ((kin[b] >> begin_bit) & m); });
        for (int i = 0; i < n; ++i) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
        return cudaSuccess;
    }
};
}  // namespace cub
