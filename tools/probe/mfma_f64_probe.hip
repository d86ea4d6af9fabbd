// Development probe (not part of the library): operand layout and issue cost of v_mfma_f64_4x4x4_4b_f64 on
// gfx950, alone and mixed with v_fma_f64, to decide whether the IDCT's 4x4 even / odd products can
// run on the matrix pipe beside the vector work.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_layout(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

template <int MODE> // 0: mfma only, 1: fma only, 2: both (independent chains), 3: 16x16x4 mfma only
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, long long *cyc) {
    double acc[4] = {1.0, 2.0, 3.0, 4.0};
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001 + i;
    const double a = threadIdx.x * 0.5, b = 1.0 - threadIdx.x * 0.25;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 big = {0, 0, 0, 0};
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fma(f[j], 1.0000001, 0.5);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fma(f[j], 0.9999999, -0.5);
        }
        if (MODE == 3) big = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, big, 0, 0, 0);
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    double s = acc[0] + acc[1] + acc[2] + acc[3] + big[0] + big[1] + big[2] + big[3];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int MODE>
static void run_rate(const char *name, int waves_per_simd, int iters) {
    double *out; long long *cyc, h[2];
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
    hipMalloc(&out, sizeof(double) * 256 * blocks); hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %.3f ms, %.1f shader cycles per iteration (wave 0), per-SIMD ns/iter/wave-set %.2f\n", name, waves_per_simd, ms,
           (double)h[0] / iters, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    // layout: D = sum_k A[i][k] B[k][j] per block; probe with one-hot A and B
    double *a, *b, *d; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    std::vector<double> ha(64), hb(64), hd(64);
    printf("one-hot probe: for A lane la = 1 and all B = 1: which D lanes become 1\n");
    for (int la = 0; la < 64; la += 1) {
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d ->", la);
        for (int i = 0; i < 64; ++i) if (hd[i] != 0) printf(" %d", i);
        printf("\n");
    }
    printf("one-hot probe: for B lane lb = 1 and all A = 1\n");
    for (int lb = 0; lb < 64; lb += 1) {
        for (int i = 0; i < 64; ++i) { hb[i] = (i == lb); ha[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("B lane %2d ->", lb);
        for (int i = 0; i < 64; ++i) if (hd[i] != 0) printf(" %d", i);
        printf("\n");
    }
    // pairing: A lane la and B lane lb both one-hot: nonzero D tells which k they share
    printf("pairs (A lane, B lane) with a nonzero product, block 0 lanes only:\n");
    for (int la = 0; la < 16; ++la) {
        printf("A %2d:", la);
        for (int lb = 0; lb < 16; ++lb) {
            for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = (i == lb); }
            hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
            int hit = -1; for (int i = 0; i < 64; ++i) if (hd[i] != 0) hit = i;
            printf(" %2d", hit);
        }
        printf("\n");
    }
    const int iters = 20000;
    for (int w : {1, 2, 4}) {
        run_rate<0>("4 x mfma_f64_4x4x4 / iter", w, iters);
        run_rate<1>("16 x v_fma_f64 / iter", w, iters);
        run_rate<2>("4 mfma + 16 fma / iter", w, iters);
        run_rate<3>("1 x mfma_f64_16x16x4 / iter", w, iters);
    }
    return 0;
}
