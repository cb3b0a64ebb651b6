// reduce_check.hip -- device check of oxmpl_amd/csrc/lanes_reduce.hpp (the transposing wave reduction of the scanner waves):
// random per-lane values incl. duplicates, +inf and negative numbers; every one of the lanes 8b .. 8b+7 must hold the wave's
// minimum of value b, bit for bit, as a plain shuffle reduction computes it.  Also times both against each other.
//   hipcc --offload-arch=gfx950 -O3 -I oxmpl_amd/csrc tools/reduce_check.hip -o tools/reduce_check.bin && tools/reduce_check.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "lanes_reduce.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ float ref_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int NQ>
__global__ void check_kernel(const float* in, uint32_t* bad, int rounds) {
    const int lane = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        float v[NQ];
        for (int b = 0; b < NQ; ++b) v[b] = in[((size_t)(blockIdx.x * rounds + r) * NQ + b) * 64 + lane];
        const float u = oxhip::lanes_min_transposed<NQ>(v);
        for (int b = 0; b < NQ; ++b) {
            const float m = ref_min(v[b]);
            const float got = __shfl(u, 8 * b + (lane & 7), 64);
            if (__float_as_uint(got) != __float_as_uint(m)) atomicAdd(bad, 1u);
        }
    }
}

int main() {
    const int blocks = 256, rounds = 64;
    int rc = 0;
    for (int nq : {8, 4}) {
        const size_t n = (size_t)blocks * rounds * nq * 64;
        std::vector<float> h(n);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const uint32_t k = (uint32_t)(s >> 33);
            float x = (float)((int)(k % 2001) - 1000) * 0.03125f;          // many duplicates
            if (k % 97 == 0) x = __builtin_inff();
            if (k % 89 == 0) x = (float)(s >> 40) * 1e-3f - 7000.0f;
            if ((i / 64 / nq) % 16 == 3) x = __builtin_inff();             // whole records of +inf (an empty scanner wave)
            h[i] = x;
        }
        float* d; uint32_t* bad; uint32_t hb = 0;
        CK(hipMalloc(&d, n * sizeof(float)));
        CK(hipMalloc(&bad, 4));
        CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemset(bad, 0, 4));
        if (nq == 8) hipLaunchKernelGGL(check_kernel<8>, dim3(blocks), dim3(64), 0, 0, d, bad, rounds);
        else hipLaunchKernelGGL(check_kernel<4>, dim3(blocks), dim3(64), 0, 0, d, bad, rounds);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"values_per_lane\": %d, \"records\": %d, \"mismatches\": %u}\n", nq, blocks * rounds * nq, hb);
        if (hb) rc = 1;
        CK(hipFree(d)); CK(hipFree(bad));
    }
    return rc;
}
