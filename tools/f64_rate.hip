// Microbenchmark: wave64 f64 VALU issue rate on gfx950 (cycles per wave-instruction per SIMD).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/f64_rate.hip -o tools/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 0.5;
    uint64_t t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
        if (OP == 1) { a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m; }
        if (OP == 2) { a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                       a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c); }
        if (OP == 3) {  // sub, mul, add chain like dist2 (dependent chain of 3 per accumulator)
            double d0 = a0 - c, d1 = a1 - c, d2 = a2 - c, d3 = a3 - c;
            d0 *= d0; d1 *= d1; d2 *= d2; d3 *= d3;
            a0 = d0 + d1; a1 = d2 + d3; a2 = d0 + d3; a3 = d1 + d2; }
        if (OP == 4) {  // cmp + 2 cndmask
            bool l0 = a0 < a1, l1 = a2 < a3, l2 = a4 < a5, l3 = a6 < a7;
            a1 = l0 ? a0 : a1; a3 = l1 ? a2 : a3; a5 = l2 ? a4 : a5; a7 = l3 ? a6 : a7;
            a0 += c; a2 += c; a4 += c; a6 += c; }
        if (OP == 5) {  // f32 fma for reference
            float f0 = (float)a0, f1 = (float)a1;
            for (int j = 0; j < 4; ++j) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); }
            a0 = f0; a1 = f1; }
    }
    uint64_t t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((uint64_t*)out)[1 << 20] = t1 - t0;
}

template <int OP>
void run(const char* name, int ops_per_iter, int threads) {
    double* d;
    hipMalloc(&d, ((1 << 20) + 8) * sizeof(double));
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, threads>>>(d, 100, 1.0);
    hipEventRecord(e0);
    k<OP><<<256, threads>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t cyc; hipMemcpy(&cyc, (uint64_t*)d + (1 << 20), 8, hipMemcpyDeviceToHost);
    int waves_per_simd = threads / 64 / 4;
    double per = (double)cyc / ((double)iters * ops_per_iter * (waves_per_simd ? waves_per_simd : 1));
    printf("%-28s threads=%4d  %.3f ms  %llu cyc  -> %.2f cyc per wave-instr per SIMD (clock %.2f GHz)\n", name, threads, ms,
           (unsigned long long)cyc, per, cyc / (ms * 1e6));
    hipFree(d);
}

int main() {
    for (int threads : {256, 1024}) {
        run<0>("v_add_f64 x8", 8, threads);
        run<1>("v_mul_f64 x8", 8, threads);
        run<2>("v_fma_f64 x8", 8, threads);
        run<3>("dist2-like 12 f64 ops", 12, threads);
        run<4>("4x(cmp+2cndmask)+4 add", 8 + 8, threads);
    }
    return 0;
}
