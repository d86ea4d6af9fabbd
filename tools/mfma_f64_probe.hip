// Development aid (round-2 lead): could the pixel kernel's two 8-point passes run on the matrix
// pipe beside the vector pipe?  v_mfma_f64_4x4x4_4b_f64 does four independent (4x4)*(4x4) products
// per instruction -- the shape of the even and odd halves of an 8-point transform on 4 columns.
//  1. lane maps of A, B, D (found with one-hot operands), 2. issue cost alone,
//  3. issue cost next to independent v_fma_f64 of the same wave and of other waves.
// hipcc --offload-arch=gfx950 -O3 mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_map(double *out) { // block = (la, lb): A one-hot at lane la, B one-hot at lane lb
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[blockIdx.x * 64 + l] = d;
}

template <int MODE> // 0: MFMA only, 1: v_fma_f64 only, 2: both interleaved 1 MFMA : 4 FMA, 3: 1 : 8
__global__ __launch_bounds__(256) void k_rate(double *out, double a, double b, int n) {
    double acc[4] = {a, a + 1, a + 2, a + 3};
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE != 1) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE == 1 || MODE == 2) { x[2 * i] = __builtin_fma(x[2 * i], a, b); x[2 * i + 1] = __builtin_fma(x[2 * i + 1], a, b);
                                          x[(2 * i + 2) & 7] = __builtin_fma(x[(2 * i + 2) & 7], a, b); x[(2 * i + 3) & 7] = __builtin_fma(x[(2 * i + 3) & 7], a, b); }
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], a, b);
            }
        }
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void rate(const char *name, double *d, int wps, int mfma_per_it, int fma_per_it) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wps, iters = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 64);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * p.clockRate * 1e3 / ((double)wps * iters); // cycles per loop iteration per wave-slot
    printf("%-34s waves/SIMD %d: %.3f ms, %.1f cycles per iteration (%d MFMA + %d v_fma_f64) per SIMD\n", name, wps, ms, cyc,
           mfma_per_it, fma_per_it);
}

int main() {
    double *d; (void)hipMalloc(&d, sizeof(double) * 64 * 4096 * 16);
    hipLaunchKernelGGL(k_map, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(4096 * 64);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // for each (la, lb): which D lanes are non-zero
    printf("pairs (A lane, B lane) -> D lane for A lane 0..3 and 16, 17 (all B lanes that hit):\n");
    for (int la : {0, 1, 2, 3, 4, 5, 16, 17, 20}) {
        printf(" A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" B%d->D%d", lb, l);
        printf("\n");
    }
    // summary statistics: how many (la, lb) pairs hit, and hits per pair
    int hits = 0, multi = 0;
    for (int q = 0; q < 4096; ++q) { int c = 0; for (int l = 0; l < 64; ++l) c += h[q * 64 + l] != 0.0; hits += c > 0; multi += c > 1; }
    printf("pairs with a product: %d of 4096 (expected 4 blocks * 4 i * 4 k * 4 j = 256), with more than one D lane: %d\n", hits, multi);
    for (int w : {1, 2, 4}) {
        rate<0>("MFMA f64 4x4x4_4b only", d, w, 4, 0);
        rate<1>("v_fma_f64 only", d, w, 0, 16);
        rate<2>("interleaved 4 MFMA + 16 FMA", d, w, 4, 16);
        rate<3>("interleaved 4 MFMA + 32 FMA", d, w, 4, 32);
    }
    (void)hipFree(d);
    return 0;
}
