// Issue rate of the fp64 vector instructions on this GPU: v_fma_f64 against separate v_mul_f64 + v_add_f64
// (what code must use when every operation has to round on its own), with vector and with scalar operands.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/fp64_rate.hip -o /tmp/fp64_rate && /tmp/fp64_rate
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kAcc = 16;

// MODE 0: v_fma_f64, weights from a small table (scalar loads that hit the scalar cache)
// MODE 2: v_mul_f64 + v_add_f64, same weights
// MODE 3: as 2, the weights streamed from a large table, a different tile per wave (scalar cache misses)
// MODE 4: as 3, and the per-lane factor read from LDS every step
template <int MODE>
__global__ void __launch_bounds__(512) rate_kernel(double* out, const double* __restrict__ w, int iters) {
    __shared__ double sh[128 * 65];
    double acc[kAcc];
#pragma unroll
    for (int u = 0; u < kAcc; ++u) acc[u] = (double)(threadIdx.x + u);
    for (int i = threadIdx.x; i < 128 * 65; i += blockDim.x) sh[i] = out[i & 7] + 1.0;
    __syncthreads();
    double h = out[threadIdx.x & 7];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t tile = (MODE >= 3) ? ((size_t)(blockIdx.x % 192) * 16384 + wave * kAcc) : 0;
    const int ldw = (MODE >= 3) ? 128 : kAcc;
    for (int it = 0; it < iters; ++it) {
        const int j = MODE >= 3 ? (it & 127) : (it & 63);
        if (MODE == 4) h = sh[j * 65 + (threadIdx.x & 63)];
        double ws[kAcc];
#pragma unroll
        for (int u = 0; u < kAcc; ++u) ws[u] = w[tile + (size_t)j * ldw + u];  // uniform -> scalar loads
#pragma unroll
        for (int u = 0; u < kAcc; ++u) {
            if (MODE == 0) acc[u] = __builtin_fma(ws[u], h, acc[u]);
            else acc[u] = acc[u] + ws[u] * h;  // separate multiply and add (-ffp-contract=off)
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < kAcc; ++u) s += acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(double* out, const double* w, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, w, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, w, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 512 * iters * kAcc * 2.0 / (ms * 1e-3) / 1e12;  // flop: one multiply and one add per element
}

int main() {
    double *out, *w;
    hipMalloc(&out, sizeof(double) * 256 * 16 * 256);
    hipMalloc(&w, sizeof(double) * 192 * 16384 + 4096);
    hipMemset(out, 0, sizeof(double) * 256 * 16 * 256);
    hipMemset(w, 0, sizeof(double) * 192 * 16384 + 4096);
    const int iters = 4096;
    std::printf("fma, cached weights                      : %.1f TFLOP/s\n", run<0>(out, w, iters));
    std::printf("mul + add, cached weights                : %.1f TFLOP/s\n", run<2>(out, w, iters));
    std::printf("mul + add, streamed weights              : %.1f TFLOP/s\n", run<3>(out, w, iters));
    std::printf("mul + add, streamed weights, LDS factor  : %.1f TFLOP/s\n", run<4>(out, w, iters));
    return 0;
}
