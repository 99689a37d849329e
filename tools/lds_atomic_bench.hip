// Microbenchmark: throughput of no-return LDS atomics with random addresses (the scatter pattern of the
// sparse mobility kernel): two 32-bit adds into two arrays versus one 64-bit add into a packed array.
//   hipcc -O3 --offload-arch=gfx950 -o tools/lds_atomic_bench tools/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 20000, THREADS = 768, ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(THREADS) void k(unsigned* out) {
    extern __shared__ unsigned char raw[];
    unsigned* a32 = reinterpret_cast<unsigned*>(raw);
    unsigned long long* a64 = reinterpret_cast<unsigned long long*>(raw);
    for (int i = threadIdx.x; i < 2 * N; i += THREADS) a32[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned idx = (x >> 8) % N;
        if (MODE == 0) {
            atomicAdd(&a32[idx], 1u);
            atomicAdd(&a32[N + idx], 3u);
        } else if (MODE == 1) {
            atomicAdd(&a64[idx], (3ull << 32) | 1ull);
        } else if (MODE == 2) {
            atomicAdd(&a32[idx], 1u);
        } else {  // conflict-free reference: consecutive addresses
            atomicAdd(&a32[(threadIdx.x + it * 64) % N], 1u);
        }
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 2 * N; i += THREADS) s += a32[i];
    atomicAdd(out, s);
}

template <int MODE>
float run(unsigned* d) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, N * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(THREADS), N * 8, 0, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(THREADS), N * 8, 0, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    unsigned* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    const double waveops = (double)THREADS / 64 * ITERS;  // per CU
    const float t0 = run<0>(d), t1 = run<1>(d), t2 = run<2>(d), t3 = run<3>(d);
    printf("2 x u32 random: %.3f ms  (%.1f cycles per wave iteration at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / waveops);
    printf("1 x u64 random: %.3f ms  (%.1f)\n", t1, t1 * 1e-3 * 2.4e9 / waveops);
    printf("1 x u32 random: %.3f ms  (%.1f)\n", t2, t2 * 1e-3 * 2.4e9 / waveops);
    printf("1 x u32 linear: %.3f ms  (%.1f)\n", t3, t3 * 1e-3 * 2.4e9 / waveops);
    return 0;
}
