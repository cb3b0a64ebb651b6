// valu_mix_bench.hip -- what one gfx950 SIMD really issues per cycle, for the instruction mix of the binary32 screen
// (rrt_resident32.hip), measured rather than assumed.  Two families of kernels:
//
//   op<K>      : an unrolled stream of one instruction kind with independent destinations (v_pk_fma_f32, v_pk_add_f32,
//                v_pk_mul_f32, v_and_or_b32, v_med3_u32, v_min_u32, v_fma_f32, v_add_f64, v_fma_f64) -> issue cycles per
//                wave64 instruction per SIMD at 1..4 waves per SIMD.  Settles "2 or 4 cycles" per kind.
//   screen<S>  : the screen loop itself, same source shape as the scanner waves (S register rows x 8 queries, queries in
//                scalar registers, 3 packed + and_or + med3 + min per (row, query)) with no reduce / publish / ring ->
//                (row, query) pairs per second per CU, i.e. the ceiling of any kernel built on this screen.
//
// One workgroup per CU (LDS padding forces it), 256*W threads = W waves per SIMD.  Times are in-kernel s_memtime cycles
// (shader clock) and HIP-event wall time; JSON on stdout.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_mix_bench.hip -o tools/valu_mix_bench.bin && tools/valu_mix_bench.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kLdsPad = 84 * 1024;   // > half of 160 KB: one workgroup per CU

enum Op { PK_FMA, PK_ADD, PK_MUL, AND_OR, MED3, MIN_U32, FMA_F32, ADD_F64, FMA_F64, MIN_F32, MED3_F32, MIN3_F32, N_OPS };
static const char* kOpName[N_OPS] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_and_or_b32", "v_med3_u32", "v_min_u32",
                                     "v_fma_f32", "v_add_f64", "v_fma_f64", "v_min_f32", "v_med3_f32", "v_min3_f32"};

// 16 independent destination registers (pairs), 4 rounds unrolled in the asm block = 64 instructions per block
#define R16(INS)                                                                                                     \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

template <int OP>
__global__ void op_kernel(uint64_t* cyc, uint32_t* sink, int iters) {
    extern __shared__ char pad[];
    uint64_t a[16];
    uint64_t x = 0x3F8000013F800001ull + threadIdx.x, y = 0x3F0000003F000000ull;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + (uint64_t)i * 0x0000010000000100ull;
    if (threadIdx.x == 0xFFFF) pad[0] = 1;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
                   "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(x), "v"(y)
        if (OP == PK_FMA) {
#define I(n) "v_pk_fma_f32 %" #n ", %" #n ", %17, %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == PK_ADD) {
#define I(n) "v_pk_add_f32 %" #n ", %" #n ", %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == PK_MUL) {
#define I(n) "v_pk_mul_f32 %" #n ", %" #n ", %17\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == FMA_F64) {
#define I(n) "v_fma_f64 %" #n ", %" #n ", %17, %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == ADD_F64) {
#define I(n) "v_add_f64 %" #n ", %" #n ", %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        }
#undef OPERANDS
    }
    const uint64_t t1 = clock64();
    uint64_t f = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) f ^= a[i];
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (f == 0x1234567) sink[0] = (uint32_t)f;
}

template <int OP>
__global__ void op32_kernel(uint64_t* cyc, uint32_t* sink, int iters) {
    extern __shared__ char pad[];
    uint32_t a[16];
    uint32_t x = 0x3F800001u + threadIdx.x, y = 0x3F000000u, z = 0x1Fu;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + (uint32_t)i * 0x100u;
    if (threadIdx.x == 0xFFFF) pad[0] = 1;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
                   "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(x), "v"(y), "v"(z)
        if (OP == AND_OR) {
#define I(n) "v_and_or_b32 %" #n ", %" #n ", %17, %18\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == MED3) {
#define I(n) "v_med3_u32 %" #n ", %" #n ", %16, %17\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == MIN_U32) {
#define I(n) "v_min_u32 %" #n ", %" #n ", %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == FMA_F32) {
#define I(n) "v_fma_f32 %" #n ", %" #n ", %17, %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == MIN_F32) {
#define I(n) "v_min_f32 %" #n ", %" #n ", %16\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == MED3_F32) {
#define I(n) "v_med3_f32 %" #n ", %" #n ", %16, %17\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        } else if (OP == MIN3_F32) {
#define I(n) "v_min3_f32 %" #n ", %" #n ", %16, %17\n"
            asm volatile(R16(I) R16(I) R16(I) R16(I) OPERANDS);
#undef I
        }
#undef OPERANDS
    }
    const uint64_t t1 = clock64();
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) f ^= a[i];
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (f == 0x1234567) sink[0] = f;
}

// ---- the screen loop, as in rrt_resident32_kernel's scanner waves
__device__ __forceinline__ uint32_t f32_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bits_f32(uint32_t v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct Screen { uint32_t b1, h2; };
// MODE 0: the kernel's bookkeeping (slot folded into the key: v_and_or_b32 + v_med3_u32 + v_min_u32)
// MODE 1: smallest value only, as a float (v_min_f32)
// MODE 2: smallest and second smallest value as floats (v_med3_f32 + v_min_f32), no slot
// MODE 3: slot folded in (v_and_or_b32), then float minimum / median
template <int MODE>
__device__ __forceinline__ void screen_push(Screen& v, float s, uint32_t slot) {
    if (MODE == 0) {
        const uint32_t key = (f32_bits(s) & ~31u) | slot;
        v.h2 = umed3(key, v.b1, v.h2);
        v.b1 = key < v.b1 ? key : v.b1;
    } else {
        float key = MODE == 3 ? bits_f32((f32_bits(s) & ~31u) | slot) : s;
        float b1 = bits_f32(v.b1), h2 = bits_f32(v.h2), r;
        if (MODE >= 2) { asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(key), "v"(b1), "v"(h2)); v.h2 = f32_bits(r); }
        asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(key), "v"(b1));
        v.b1 = f32_bits(r);
    }
}

template <int S, int D, int NQ, int NT, int MODE>
__global__ __launch_bounds__(NT) void screen_kernel(const float* pts, uint64_t* cyc, uint32_t* sink, int iters) {
    static_assert(S % 2 == 0, "rows are held two per register pair");
    extern __shared__ char pad[];
    float (*qring)[4] = reinterpret_cast<float (*)[4]>(pad);   // 64 queries
    // two rows per 64-bit register pair: a packed instruction's src0 is a pair anyway, and op_sel picks the half that is
    // broadcast to both lanes of the operation -- the tree costs S*D VGPRs instead of 2*S*D
    f32x2 tr[D][S / 2];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < D; ++k) tr[k][s / 2][s % 2] = pts[((size_t)(threadIdx.x + NT * s) % 16384) * 8 + k];
    if (threadIdx.x < 64)
        for (int k = 0; k < 4; ++k) qring[threadIdx.x][k] = 1.0f + 0.125f * (float)((threadIdx.x * 7 + k * 13 + blockIdx.x) % 64);
    __syncthreads();
    Screen sc[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) sc[b] = Screen{0x7F7FFFFFu, 0x7F7FFFFFu};
    uint32_t fold = 0;
    const uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // the scanners' rows change between passes (absorbed nodes): keep the broadcasts inside the loop so that they fold
        // into the packed instruction's op_sel, as in the kernel, instead of being hoisted into registers of their own
#pragma unroll
        for (int s = 0; s < S / 2; ++s)
#pragma unroll
            for (int k = 0; k < D; ++k) asm volatile("" : "+v"(tr[k][s]));
        f32x2 q[NQ / 2][D];
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int k = 0; k < D; ++k) q[b / 2][k][b % 2] = bits_f32(uni(f32_bits(qring[(it * NQ + b) & 63][k & 3])));
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int bp = 0; bp < NQ / 2; ++bp) {
                f32x2 e = ((s & 1) ? __builtin_shufflevector(tr[0][s / 2], tr[0][s / 2], 1, 1) : __builtin_shufflevector(tr[0][s / 2], tr[0][s / 2], 0, 0)) - q[bp][0];
                f32x2 acc = e * e;
#pragma unroll
                for (int k = 1; k < D; ++k) {
                    e = ((s & 1) ? __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 1, 1) : __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 0, 0)) - q[bp][k];
                    acc = __builtin_elementwise_fma(e, e, acc);
                }
                screen_push<MODE>(sc[2 * bp], acc[0], (uint32_t)s);
                screen_push<MODE>(sc[2 * bp + 1], acc[1], (uint32_t)s);
            }
        }
        // a pass ends like the kernel's: per-query state consumed and reset (two VALU per query, amortised over S rows)
#pragma unroll
        for (int b = 0; b < NQ; ++b) { fold ^= sc[b].b1 + sc[b].h2; sc[b] = Screen{0x7F7FFFFFu, 0x7F7FFFFFu}; }
    }
    const uint64_t t1 = clock64();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (NT / 64) + threadIdx.x / 64] = t1 - t0;
    if (fold == 0x1234567) sink[0] = fold;
}

// ---- the dot-product screen of rrt_lanes.hip: a node is (a, cc = |a|^2), a query Q = -2 b, s' = cc + a . Q: D packed fused
// multiply-adds per (row, query pair) and one v_min_f32 per (row, query) -- 2.5 instructions per (row, query) in R^3
// MIN3 = 1: the screen of rrt_lanes.hip since round 2's second half -- a packed fused multiply-add covers the two ROWS of a
// register pair for one query (the query's coordinate is the broadcast half, op_sel on the second operand) and one
// v_min3_f32 folds both rows into the query's running minimum: D/2 + 1/2 = 2.0 instructions per (row, query) in R^3
template <int S, int D, int NQ, int NT, int MIN3 = 0>
__global__ __launch_bounds__(NT) void dot_kernel(const float* pts, uint64_t* cyc, uint32_t* sink, int iters) {
    static_assert(S % 2 == 0, "rows are held two per register pair");
    extern __shared__ char pad[];
    float (*qring)[8] = reinterpret_cast<float (*)[8]>(pad);   // 64 queries
    f32x2 tr[D][S / 2], tcc[S / 2];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) { const float f = pts[((size_t)(threadIdx.x + NT * s) % 16384) * 8 + k] - 5.0f; tr[k][s / 2][s % 2] = f; sq += f * f; }
        tcc[s / 2][s % 2] = sq;
    }
    if (threadIdx.x < 64)
        for (int k = 0; k < 8; ++k) qring[threadIdx.x][k] = -2.0f * (-4.0f + 0.125f * (float)((threadIdx.x * 7 + k * 13 + blockIdx.x) % 64));
    __syncthreads();
    float b1[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) b1[b] = __builtin_inff();
    uint32_t fold = 0;
    const uint64_t t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < S / 2; ++s) {
#pragma unroll
            for (int k = 0; k < D; ++k) asm volatile("" : "+v"(tr[k][s]));
            asm volatile("" : "+v"(tcc[s]));
        }
        f32x2 q[NQ / 2][D];
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int k = 0; k < D; ++k) q[b / 2][k][b % 2] = bits_f32(uni(f32_bits(qring[(it * NQ + b) & 63][k])));
        if (MIN3) {
#pragma unroll
            for (int sp = 0; sp < S / 2; ++sp) {
#pragma unroll
                for (int b0 = 0; b0 < NQ; b0 += 4) {   // four queries' chains side by side
                    f32x2 acc[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = tcc[sp];
#pragma unroll
                    for (int k = 0; k < D; ++k) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int b = b0 + t;
                            const f32x2 qq = (b & 1) ? __builtin_shufflevector(q[b / 2][k], q[b / 2][k], 1, 1) : __builtin_shufflevector(q[b / 2][k], q[b / 2][k], 0, 0);
                            acc[t] = __builtin_elementwise_fma(tr[k][sp], qq, acc[t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) asm("v_min3_f32 %0, %0, %1, %2" : "+v"(b1[b0 + t]) : "v"(acc[t][0]), "v"(acc[t][1]));
                }
            }
        } else
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x2 acc[NQ / 2];
#pragma unroll
            for (int bp = 0; bp < NQ / 2; ++bp)
                acc[bp] = (s & 1) ? __builtin_shufflevector(tcc[s / 2], tcc[s / 2], 1, 1) : __builtin_shufflevector(tcc[s / 2], tcc[s / 2], 0, 0);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const f32x2 a = (s & 1) ? __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 1, 1) : __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 0, 0);
#pragma unroll
                for (int bp = 0; bp < NQ / 2; ++bp) acc[bp] = __builtin_elementwise_fma(a, q[bp][k], acc[bp]);
            }
#pragma unroll
            for (int bp = 0; bp < NQ / 2; ++bp) {
                asm("v_min_f32 %0, %0, %1" : "+v"(b1[2 * bp]) : "v"(acc[bp][0]));
                asm("v_min_f32 %0, %0, %1" : "+v"(b1[2 * bp + 1]) : "v"(acc[bp][1]));
            }
        }
#pragma unroll
        for (int b = 0; b < NQ; ++b) { fold ^= f32_bits(b1[b]); b1[b] = __builtin_inff(); }
    }
    const uint64_t t1 = clock64();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (NT / 64) + threadIdx.x / 64] = t1 - t0;
    if (fold == 0x1234567) sink[0] = fold;
}

struct Result { double ns_per_inst_simd, cyc_median_per_inst_simd, wall_ms, clk_ghz_1w; };

static int n_cu = 256;
static double g_clk_ghz = 0.0;   // shader clock from the one-wave-per-SIMD runs (every wave runs the whole time: ticks / wall)

// Wall time (HIP events) is the ground truth: with three or more waves per SIMD the issue arbiter is not fair (oldest
// first), so one wave's s_memtime span says little about the SIMD.  ns per wave-instruction per SIMD = wall / (inst x W).
template <class F>
static int run(F launch, int waves_per_simd, int iters, double inst_per_wave_iter, Result& r, uint64_t* d_cyc, std::vector<uint64_t>& h_cyc) {
    const int n_waves = n_cu * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(iters / 8 + 1);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        launch(iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipMemcpy(h_cyc.data(), d_cyc, n_waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> v(h_cyc.begin(), h_cyc.begin() + n_waves);
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2];
    r.cyc_median_per_inst_simd = med / (inst_per_wave_iter * iters * waves_per_simd);
    r.wall_ms = best;
    r.ns_per_inst_simd = best * 1e6 / (inst_per_wave_iter * iters * waves_per_simd);
    r.clk_ghz_1w = med / (best * 1e6);
    if (waves_per_simd == 1) g_clk_ghz = r.clk_ghz_1w;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    n_cu = prop.multiProcessorCount;
    uint64_t* d_cyc; uint32_t* d_sink; float* d_pts;
    CK(hipMalloc(&d_cyc, (size_t)n_cu * 16 * sizeof(uint64_t)));
    CK(hipMalloc(&d_sink, 64));
    std::vector<float> pts(16384 * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& p : pts) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; p = (float)((s >> 40) * (10.0 / 16777216.0)); }
    CK(hipMalloc(&d_pts, pts.size() * sizeof(float)));
    CK(hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<uint64_t> h_cyc((size_t)n_cu * 16);

    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n"
           " \"note\": \"issue cost of one wave64 instruction on one SIMD, from wall time: ns and cycles at the clock the 1-wave run of the same kernel measured\",\n"
           " \"op_issue\": {\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    const int iters = 4000;
#define RUN_OP(KERN, OP, LAST)                                                                                    \
    do {                                                                                                          \
        printf("  \"%s\": {", kOpName[OP]);                                                                     \
        for (int w = 1; w <= 4; ++w) {                                                                            \
            Result r;                                                                                             \
            auto L = [&](int it) { hipLaunchKernelGGL((KERN<OP>), dim3(n_cu), dim3(256 * w), kLdsPad, 0, d_cyc, d_sink, it); }; \
            if (run(L, w, iters, 64.0, r, d_cyc, h_cyc)) return 1;                                                \
            printf("\"%dw\": {\"ns\": %.4f, \"cycles\": %.3f}%s", w, r.ns_per_inst_simd, r.ns_per_inst_simd * g_clk_ghz, w < 4 ? ", " : ""); \
        }                                                                                                         \
        printf(", \"clock_ghz\": %.3f}%s\n", g_clk_ghz, LAST ? "" : ",");                                       \
    } while (0)
    RUN_OP(op_kernel, PK_FMA, 0);
    RUN_OP(op_kernel, PK_ADD, 0);
    RUN_OP(op_kernel, PK_MUL, 0);
    RUN_OP(op32_kernel, AND_OR, 0);
    RUN_OP(op32_kernel, MED3, 0);
    RUN_OP(op32_kernel, MIN_U32, 0);
    RUN_OP(op32_kernel, FMA_F32, 0);
    RUN_OP(op32_kernel, MIN_F32, 0);
    RUN_OP(op32_kernel, MED3_F32, 0);
    RUN_OP(op32_kernel, MIN3_F32, 0);
    RUN_OP(op_kernel, ADD_F64, 0);
    RUN_OP(op_kernel, FMA_F64, 1);
    printf(" },\n \"screen\": [\n");
    // the screen loop: S rows x 8 queries per pass; rq = (row, query) pairs; mode: see screen_push
    bool first = true;
#define RUN_SCREEN(S_, D_, W_, MODE_)                                                                              \
    do {                                                                                                          \
        Result r;                                                                                                 \
        const int it = 3000;                                                                                      \
        auto L = [&](int n) { hipLaunchKernelGGL((screen_kernel<S_, D_, 8, 256 * W_, MODE_>), dim3(n_cu), dim3(256 * W_), kLdsPad, 0, d_pts, d_cyc, d_sink, n); }; \
        if (run(L, W_, it, (double)(S_) * 8.0, r, d_cyc, h_cyc)) return 1;                                        \
        const double rq_per_s_chip = (double)n_cu * 4 * W_ * (double)(S_) * 8.0 * it / (r.wall_ms * 1e-3);       \
        printf("%s  {\"mode\": %d, \"rows\": %d, \"dim\": %d, \"waves_per_simd\": %d, \"ns_per_row_query_per_simd\": %.4f, \"wall_ms\": %.3f, " \
               "\"row_queries_per_s_chip\": %.4e}", first ? "" : ",\n", MODE_, S_, D_, W_, r.ns_per_inst_simd, r.wall_ms, rq_per_s_chip); \
        first = false;                                                                                            \
    } while (0)
    RUN_SCREEN(22, 3, 1, 0); RUN_SCREEN(22, 3, 2, 0); RUN_SCREEN(22, 3, 3, 0);
    RUN_SCREEN(14, 3, 2, 0); RUN_SCREEN(14, 3, 3, 0); RUN_SCREEN(14, 3, 4, 0);
    RUN_SCREEN(20, 2, 2, 0); RUN_SCREEN(20, 2, 3, 0);
    RUN_SCREEN(20, 4, 2, 0); RUN_SCREEN(14, 6, 2, 0); RUN_SCREEN(10, 6, 3, 0);
    // alternative bookkeeping (design study for the next kernel): float min only / float min + med3 / fold + float ops
    RUN_SCREEN(22, 3, 2, 1); RUN_SCREEN(22, 3, 3, 1); RUN_SCREEN(22, 3, 2, 2); RUN_SCREEN(22, 3, 3, 2); RUN_SCREEN(22, 3, 2, 3);
    RUN_SCREEN(40, 3, 2, 1); RUN_SCREEN(40, 3, 3, 1); RUN_SCREEN(40, 3, 2, 2); RUN_SCREEN(40, 3, 3, 2);
    RUN_SCREEN(20, 6, 2, 1); RUN_SCREEN(20, 4, 2, 1); RUN_SCREEN(20, 4, 3, 1);
    printf("\n ],\n \"dot_screen\": [\n");
    first = true;
#define RUN_DOT(S_, D_, W_)                                                                                        \
    do {                                                                                                          \
        Result r;                                                                                                 \
        const int it = 3000;                                                                                      \
        auto L = [&](int n) { hipLaunchKernelGGL((dot_kernel<S_, D_, 8, 256 * W_>), dim3(n_cu), dim3(256 * W_), kLdsPad, 0, d_pts, d_cyc, d_sink, n); }; \
        if (run(L, W_, it, (double)(S_) * 8.0, r, d_cyc, h_cyc)) return 1;                                        \
        const double rq_per_s_chip = (double)n_cu * 4 * W_ * (double)(S_) * 8.0 * it / (r.wall_ms * 1e-3);       \
        printf("%s  {\"rows\": %d, \"dim\": %d, \"waves_per_simd\": %d, \"ns_per_row_query_per_simd\": %.4f, \"wall_ms\": %.3f, " \
               "\"row_queries_per_s_chip\": %.4e}", first ? "" : ",\n", S_, D_, W_, r.ns_per_inst_simd, r.wall_ms, rq_per_s_chip); \
        first = false;                                                                                            \
    } while (0)
    RUN_DOT(24, 3, 1); RUN_DOT(24, 3, 2); RUN_DOT(24, 3, 3); RUN_DOT(16, 3, 2);
    RUN_DOT(24, 2, 2); RUN_DOT(20, 4, 2); RUN_DOT(20, 5, 2); RUN_DOT(16, 6, 2);
    printf("\n ],\n \"dot_screen_min3\": [\n");
    first = true;
#define RUN_DOT3(S_, D_, W_, NQ_)                                                                                  \
    do {                                                                                                          \
        Result r;                                                                                                 \
        const int it = 3000;                                                                                      \
        auto L = [&](int n) { hipLaunchKernelGGL((dot_kernel<S_, D_, NQ_, 256 * W_, 1>), dim3(n_cu), dim3(256 * W_), kLdsPad, 0, d_pts, d_cyc, d_sink, n); }; \
        if (run(L, W_, it, (double)(S_) * (double)(NQ_), r, d_cyc, h_cyc)) return 1;                              \
        const double rq_per_s_chip = (double)n_cu * 4 * W_ * (double)(S_) * (double)(NQ_) * it / (r.wall_ms * 1e-3); \
        printf("%s  {\"rows\": %d, \"dim\": %d, \"waves_per_simd\": %d, \"queries_per_pass\": %d, \"ns_per_row_query_per_simd\": %.4f, \"wall_ms\": %.3f, " \
               "\"row_queries_per_s_chip\": %.4e}", first ? "" : ",\n", S_, D_, W_, NQ_, r.ns_per_inst_simd, r.wall_ms, rq_per_s_chip); \
        first = false;                                                                                            \
    } while (0)
    RUN_DOT3(24, 3, 1, 8); RUN_DOT3(24, 3, 2, 8); RUN_DOT3(24, 3, 3, 8); RUN_DOT3(16, 3, 2, 8); RUN_DOT3(20, 3, 2, 8);
    RUN_DOT3(24, 2, 2, 8); RUN_DOT3(20, 4, 2, 8); RUN_DOT3(20, 5, 2, 8); RUN_DOT3(16, 6, 2, 8); RUN_DOT3(16, 6, 2, 4);
    printf("\n ]}\n");
    return 0;
}
