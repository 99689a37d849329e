// Issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on one device: independent accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_rate tools/mfma_f64_rate.hip && tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
    f64x4 d[ACC];
    for (int k = 0; k < ACC; ++k) d[k] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < ACC; ++k) d[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d[k], 0, 0, 0);
    double s = 0;
    for (int k = 0; k < ACC; ++k) s += d[k][0] + d[k][1] + d[k][2] + d[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
    double d[ACC];
    for (int k = 0; k < ACC; ++k) d[k] = k;
    double a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < ACC; ++k) d[k] = __builtin_fma(a, b, d[k]);
    double s = 0;
    for (int k = 0; k < ACC; ++k) s += d[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name, double flop_per_thread_iter) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double flop = flop_per_thread_iter * iters * (double)blocks * threads;
        printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
    };
    // one MFMA: 16 x 16 x 4 multiply-adds per wave = 2048 flop per 64 lanes = 32 flop per lane
    time([&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, "mfma_f64_16x16x4, 8 acc", 8 * 32.0);
    time([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, "mfma_f64_16x16x4, 4 acc", 4 * 32.0);
    time([&] { hipLaunchKernelGGL(fma_loop<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, "v_fma_f64, 16 acc", 16 * 2.0);
    return 0;
}
