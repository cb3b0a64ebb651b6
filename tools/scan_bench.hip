// Microbenchmark: the register-resident nearest-neighbour scan alone (no resolve, no barriers),
// 16 waves per CU, 10 slots x 3 f64 per thread -- what bounds the resident kernel's scan phase.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I oxmpl_amd/csrc tools/scan_bench.hip -o tools/scan_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "rrt_device.hpp"
using namespace oxhip;

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int VAR, int NT = 1024, int S = 10>
__global__ __launch_bounds__(NT) void scan_kernel(const double* pts, double* out, int iters) {
    double tr[3][S];
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < 3; ++k) tr[k][s] = pts[(size_t)((threadIdx.x + NT * s) % 10240) * 3 + k];
    double q[3] = {5.0 + 1e-3 * blockIdx.x, 5.0, 5.0};
    double accd = 0.0; uint32_t acci = 0;
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // a new uniform query every iteration (cheap LCG on the scalar unit)
        uint32_t r = (uint32_t)it * 2654435761u + blockIdx.x;
        q[0] = 1.0 + (double)(r & 1023) * (8.0 / 1024.0);
        q[1] = 1.0 + (double)((r >> 10) & 1023) * (8.0 / 1024.0);
        q[2] = 1.0 + (double)((r >> 20) & 1023) * (8.0 / 1024.0);
        if (VAR == 0) {          // current: b1 + i1 + b2 via fmin/fmax
            Best b = best_init();
#pragma unroll
            for (int s = 0; s < S; ++s) { double c[3] = {tr[0][s], tr[1][s], tr[2][s]}; best_push(b, dist2<3>(c, q, 3), threadIdx.x + 1024 * s); }
            accd += b.b1 + b.b2; acci += b.i1;
        } else if (VAR == 1) {   // b1 + slot + high-dword second-min via med3
            double b1 = __builtin_inf(); uint32_t sl = 0, h2 = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double c[3] = {tr[0][s], tr[1][s], tr[2][s]};
                double d = dist2<3>(c, q, 3);
                uint32_t hd = (uint32_t)__double2hiint(d), hb = (uint32_t)__double2hiint(b1);
                h2 = umed3(hd, hb, h2);
                bool lt = d < b1;
                b1 = lt ? d : b1; sl = lt ? (uint32_t)s : sl;
            }
            accd += b1; acci += sl + h2;
        } else if (VAR == 2) {   // b1 + slot only (no tie detection): lower bound with index
            double b1 = __builtin_inf(); uint32_t sl = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double c[3] = {tr[0][s], tr[1][s], tr[2][s]};
                double d = dist2<3>(c, q, 3);
                bool lt = d < b1;
                b1 = lt ? d : b1; sl = lt ? (uint32_t)s : sl;
            }
            accd += b1; acci += sl;
        } else if (VAR == 3) {   // pure min of d2 (8 arithmetic + 1 min)
            double b1 = __builtin_inf();
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double c[3] = {tr[0][s], tr[1][s], tr[2][s]};
                double d = dist2<3>(c, q, 3);
                asm("v_min_f64 %0, %1, %2" : "=v"(b1) : "v"(b1), "v"(d));
            }
            accd += b1;
        } else if (VAR == 4) {   // integer-compare variant: d2 >= 0 so the bit pattern orders like the value
            uint64_t b1 = ~0ull; uint32_t sl = 0, h2 = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double c[3] = {tr[0][s], tr[1][s], tr[2][s]};
                uint64_t d = (uint64_t)__double_as_longlong(dist2<3>(c, q, 3));
                h2 = umed3((uint32_t)(d >> 32), (uint32_t)(b1 >> 32), h2);
                bool lt = d < b1;
                b1 = lt ? d : b1; sl = lt ? (uint32_t)s : sl;
            }
            accd += __longlong_as_double((long long)b1); acci += sl + h2;
        }
    }
    uint64_t t1 = clock64();
    out[blockIdx.x * NT + threadIdx.x] = accd + acci;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((uint64_t*)out)[1 << 20] = t1 - t0;
}

template <int VAR, int NT = 1024, int S = 10>
int run(const char* name, const double* pts, double* out) {
    int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    scan_kernel<VAR, NT, S><<<256, NT>>>(pts, out, 200);
    CK(hipEventRecord(e0));
    scan_kernel<VAR, NT, S><<<256, NT>>>(pts, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cyc; CK(hipMemcpy(&cyc, (uint64_t*)out + (1 << 20), 8, hipMemcpyDeviceToHost));
    printf("%-44s %.3f ms  %.0f cyc/iter (wave0)  %.3f us/iter/CU  clock %.2f GHz  -> %.1f M scans/s chip\n", name, ms,
           (double)cyc / iters, ms * 1e3 / iters, cyc / (ms * 1e6), 256.0 * iters / (ms * 1e3));
    return 0;
}

int main() {
    double *pts, *out;
    CK(hipMalloc(&pts, 10240 * 3 * sizeof(double)));
    CK(hipMalloc(&out, ((1 << 20) + 8) * sizeof(double)));
    double* h = new double[10240 * 3];
    uint64_t s = 12345;
    for (int i = 0; i < 10240 * 3; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = 10.0 * (double)(s >> 11) / 9007199254740992.0; }
    CK(hipMemcpy(pts, h, 10240 * 3 * sizeof(double), hipMemcpyHostToDevice));
    run<0>("0 b1+i1+b2 (fmin/fmax, current)", pts, out);
    run<1>("1 b1+slot+med3 high-dword second-min", pts, out);
    run<2>("2 b1+slot only", pts, out);
    run<3>("3 pure v_min_f64", pts, out);
    run<4>("4 u64 compare + med3", pts, out);
    run<1, 768, 14>("1 med3, 768 threads x 14 slots", pts, out);
    run<1, 512, 20>("1 med3, 512 threads x 20 slots", pts, out);
    run<1, 256, 40>("1 med3, 256 threads x 40 slots", pts, out);
    run<3, 512, 20>("3 pure min, 512 threads x 20 slots", pts, out);
    return 0;
}
