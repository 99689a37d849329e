// Calibration of the rocprofv3 FETCH_SIZE counter for the access pattern of the episode kernel:
// every lane reads 8 bytes, consecutive lanes consecutive addresses (512 B per wave instruction).
// Reads a buffer far larger than L2 + Infinity Cache once, so the true HBM read volume is its size.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch      (and a second run with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double* __restrict__ p, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += p[i];
    if (acc == 12345.678) out[0] = acc;  // keep the loads alive
}

__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}

// WRITE_SIZE for the kernels' store pattern: 4 bytes per lane, consecutive lanes consecutive addresses.
__global__ void write4(int* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (int)i;
}

__global__ void write8(double* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (double)i;
}

int main() {
    const size_t bytes = (size_t)4 << 30;  // 4 GiB
    double *buf = nullptr, *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read8, dim3(2048), dim3(256), 0, 0, buf, bytes / 8, out);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(write4, dim3(2048), dim3(256), 0, 0, (int*)buf, bytes / 4);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, 0, buf, bytes / 8);
    hipDeviceSynchronize();
    printf("read / wrote %zu bytes per kernel\n", bytes);
    return 0;
}
