// Development aid: issue cost of the f64 instructions the pixel kernel is made of, on this GPU.
// Each wave runs N dependent-free streams of one instruction; cycles per wave-instruction per SIMD
// = elapsed * clock * SIMDs / (waves * instructions).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int n) {
    double x[8];
    float f[8];
    int iv[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i + threadIdx.x; f[i] = (float)x[i]; iv[i] = (int)threadIdx.x + i; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
            else if (OP == 1) x[i] = x[i] + b;
            else if (OP == 2) x[i] = x[i] * a;
            else if (OP == 3) f[i] = __builtin_fmaf(f[i], (float)a, (float)b);
            else if (OP == 4) { x[i] = (double)iv[i]; iv[i] += (int)x[i] & 1; }  // cvt_f64_i32 + cvt back + and + add
            else if (OP == 5) { iv[i] = iv[i] * 3 + 1; }                          // integer baseline
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + f[i] + iv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
static void run(const char *name, double *d, int waves_per_simd) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one wave per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double clk = p.clockRate * 1e3; // Hz
    double instr_per_simd = (double)waves_per_simd * ITERS * 8;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction at %.0f MHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk / instr_per_simd, clk / 1e6);
}
int main() {
    double *d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 16);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f64", d, w); run<1>("v_add_f64", d, w); run<2>("v_mul_f64", d, w);
        run<3>("v_fma_f32", d, w); run<4>("cvt_f64_i32 + 3 int ops", d, w); run<5>("int mad (baseline)", d, w);
    }
    hipFree(d);
    return 0;
}
