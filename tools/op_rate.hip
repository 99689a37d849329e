// Issue rate of single vector instructions the apportionment's key computation uses (wave-instructions per
// cycle and SIMD, from a long dependent-free stream): v_mul_f64, v_fract_f64, v_cvt_i32_f64, v_add_u32, and the
// instructions of a double-precision division (EpidemicEnv.step divides seven times per country and day).
//   hipcc -O3 --offload-arch=gfx950 tools/op_rate.hip -o tools/op_rate && tools/op_rate
#include <hip/hip_runtime.h>

#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) op_kernel(double* out, int iters) {
    double v[8];
    int w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = out[threadIdx.x + u] + 1.5 + u; w[u] = threadIdx.x + u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 1) asm volatile("v_fract_f64 %0, %1" : "=v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 2) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(w[u]) : "v"(v[u]));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[u]) : "v"(w[(u + 1) & 7]));
            if (OP == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 6) asm volatile("v_rcp_f64 %0, %1" : "=v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 7) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2" : "=v"(v[u]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]) : "vcc");
            if (OP == 8) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(v[u]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]) : "vcc");
            if (OP == 9) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(v[u]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
            if (OP == 10) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[u]) : "v"(v[(u + 1) & 7]), "v"(v[(u + 2) & 7]));
            if (OP == 11) asm volatile("v_floor_f64 %0, %1" : "=v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 12) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[u]) : "v"(v[(u + 1) & 7]));
            if (OP == 13) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[u]) : "v"(w[u]));
            if (OP == 5) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(w[u]) : "v"(w[(u + 1) & 7]) : "vcc");
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u] + w[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, double* out, int per_iter) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 20000;
    hipLaunchKernelGGL(op_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(op_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 4 * iters * 8 * per_iter;
    const double cycles = ms * 1e-3 * 2.4e9 * 1024;  // SIMD-cycles at 2.4 GHz
    std::printf("%-28s %.2f cycles per wave instruction and SIMD\n", name, cycles / wave_instr);
}

int main() {
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * 8 * 256 + 64);
    hipMemset(out, 0, sizeof(double) * 256 * 8 * 256 + 64);
    run<3>("v_add_u32", out, 1);
    run<0>("v_mul_f64", out, 1);
    run<4>("v_add_f64", out, 1);
    run<1>("v_fract_f64", out, 1);
    run<2>("v_cvt_i32_f64", out, 1);
    run<5>("v_cmp + v_addc", out, 2);
    run<10>("v_fma_f64", out, 1);
    run<6>("v_rcp_f64", out, 1);
    run<7>("v_div_scale_f64", out, 1);
    run<8>("v_div_fmas_f64", out, 1);
    run<9>("v_div_fixup_f64", out, 1);
    run<11>("v_floor_f64", out, 1);
    run<12>("v_max_f64", out, 1);
    run<13>("v_cvt_f64_i32", out, 1);
    return 0;
}
