// Compile-time sorting networks for short register arrays (Batcher's odd-even merge sort, the
// iterative form that is valid for any length).  Every comparator has constant indices, so an
// array sorted with sort_desc() stays in registers; there is no data-dependent control flow.
// Plain C++17: the same code is checked on the host and used in the kernels.
#pragma once
#include <stdint.h>

#include <utility>

#if defined(__HIPCC__)
#define EPI_HD __host__ __device__ __forceinline__
#else
#define EPI_HD inline
#endif

namespace epi {

template <int N>
struct OddEvenNet {
    static constexpr int kCap = N * 8;  // upper bound on the comparator count for N <= 64
    int a[kCap], b[kCap];
    int count;
    constexpr OddEvenNet() : a{}, b{}, count(0) {
        for (int p = 1; p < N; p *= 2)
            for (int k = p; k >= 1; k /= 2)
                for (int j = k % p; j + k < N; j += 2 * k) {
                    const int lim = k < N - j - k ? k : N - j - k;
                    for (int i = 0; i < lim; ++i)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                            a[count] = i + j;
                            b[count] = i + j + k;
                            ++count;
                        }
                }
    }
};

template <int N, int I>
EPI_HD void sort_net_step(uint32_t (&s)[N]) {
    constexpr OddEvenNet<N> net{};
    if constexpr (I < net.count) {
        constexpr int ia = net.a[I], ib = net.b[I];
        const uint32_t hi = s[ia] > s[ib] ? s[ia] : s[ib];
        const uint32_t lo = s[ia] > s[ib] ? s[ib] : s[ia];
        s[ia] = hi;
        s[ib] = lo;
    }
}

template <int N, int... I>
EPI_HD void sort_net_run(uint32_t (&s)[N], std::integer_sequence<int, I...>) {
    (sort_net_step<N, I>(s), ...);
}

// Sorts s[0..N) in place, largest first.
template <int N>
EPI_HD void sort_desc(uint32_t (&s)[N]) {
    sort_net_run<N>(s, std::make_integer_sequence<int, OddEvenNet<N>::kCap>{});
}

}  // namespace epi
