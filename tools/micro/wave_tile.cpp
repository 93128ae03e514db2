// Microbenchmark (round 5): what the WAVE TILE of the tail kernel's products is worth before the kernel is
// rewritten around it.  Eight waves per workgroup (two per SIMD), a 128-token x 256-deep activation tile stationary in LDS, the
// weights streamed from L2 into a two-set register ping-pong -- tail16_kernel's structure -- in two decompositions of the
// 128 x 256 output block:
//   layout 0 (the kernel's):  8 waves x (32 features x 128 tokens): every wave reads the WHOLE activation tile (and, in the
//                             compensated products, gathers its e5m2 bytes) for its 32 columns;
//   layout 1 (2 x 4):         wave = 64 features x 64 tokens: an activation fragment (and its gather, and its lo bytes) feeds
//                             TWO column tiles -- half the LDS read bytes and gather instructions per MFMA, the weight bytes from
//                             L2 twice (the two token halves both stream the block's weights);
//   layout 2 (4 waves):       4 waves x (64 features x 128 tokens), ONE wave per SIMD with the 512-register budget: the same halving
//                             of LDS reads and gathers with no weight byte streamed twice -- and nothing to issue MFMAs from while
//                             the one wave of a SIMD waits (the form VERDICT r04 item 1b names).
// Products: PLAIN = fp16 x fp16 (the MLP: two thirds of the tile's FLOPs), COMP = hi + lo8 + lo2 (in_proj / out_proj / score:
// per 64-deep set and row tile four fp16 MFMAs + two block-scaled fp8 MFMAs, compute_tm's LO2 form).
// FILL VALU instructions per set stand for the LayerNorm / GELU / epilogue work next to the products (7.7 VALU per MFMA in the
// kernel: profiles/r04_pmc_summary.txt).  Wall time, in-kernel clock, TFLOP/s (hi products only), medians of interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_tile tools/micro/wave_tile.cpp && /tmp/wave_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int RS16 = 264, RSL = 272, BM = 128;

__device__ __forceinline__ void gather_e5m2(f16x8 af, int& w0, int& w1) {
    const u32x4 d = __builtin_bit_cast(u32x4, af);
    const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
    w0 = (int)__builtin_amdgcn_perm(d1, d0, 0x07050301u);
    w1 = (int)__builtin_amdgcn_perm(d3, d2, 0x07050301u);
}

// one lane's weight set: NC column tiles x (4 hi fragments + 32 lo bytes)
template <int NC>
struct WSet {
    f16x8 hi[NC][4];
    i32x8 lo[NC];
};
template <int NC, bool COMP>
__device__ __forceinline__ void load_wset(WSet<NC>& s, const f16x8* wp, int it, int wave, int lane) {
    // (a set of the packed weights: consecutive 16-byte fragments 64 lanes apart, as load_set reads them)
    const f16x8* p = wp + ((size_t)((it & 15) * 8 + wave) * (NC * 6)) * 64 + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s.hi[c][k] = p[(c * 6 + k) * 64];
        if constexpr (COMP) {
            const i32x4 a = __builtin_bit_cast(i32x4, p[(c * 6 + 4) * 64]), b = __builtin_bit_cast(i32x4, p[(c * 6 + 5) * 64]);
            s.lo[c] = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
    }
}

template <int LAYOUT, bool COMP, int FILL>
__global__ __launch_bounds__(LAYOUT == 2 ? 256 : 512) void probe(unsigned long long* out, int iters, float* sink, const _Float16* src, const f16x8* wp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    unsigned char* Al = smem + BM * RS16 * 2;
    for (int i = threadIdx.x; i < BM * RS16; i += blockDim.x) As[i] = src[(i * 7 + blockIdx.x * 131) & 65535];
    for (int i = threadIdx.x; i < BM * RSL; i += blockDim.x) Al[i] = (unsigned char)(__builtin_bit_cast(unsigned short, src[(i * 13 + 5) & 65535]) >> 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NC = LAYOUT == 0 ? 1 : 2, NMT = LAYOUT == 1 ? 2 : 4;
    const int tok0 = LAYOUT == 1 ? 64 * (wave >> 2) : 0;
    __syncthreads();
    float fill = (float)lane;
    f32x16 acc[NMT][NC] = {};
    WSet<NC> ws[2];
    load_wset<NC, COMP>(ws[0], wp, 0, wave, lane);
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto one_set = [&](const WSet<NC>& w, int it) {
        const int kb = (it & 3) * 64;
        const _Float16* a0 = As + (tok0 + (lane & 31)) * RS16 + kb + (lane >> 5) * 8;
        const unsigned char* al0 = Al + (tok0 + (lane & 31)) * RSL + kb + 32 * (lane >> 5);
        i32x8 w8h[NC];
        if constexpr (COMP) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    int x0, x1;
                    gather_e5m2(w.hi[c][ks], x0, x1);
                    w8h[c][2 * ks] = x0, w8h[c][2 * ks + 1] = x1;
                }
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            i32x8 a8 = {0, 0, 0, 0, 0, 0, 0, 0}, alo = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (COMP) {
                const i32x4 p0 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL), p1 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL + 16);
                alo = i32x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * RS16 + ks * 16);
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi[c][ks], a, acc[mt][c], 0, 0, 0);
                if constexpr (COMP) {
                    int x0, x1;
                    gather_e5m2(a, x0, x1);
                    a8[2 * ks] = x0, a8[2 * ks + 1] = x1;
                }
            }
            if constexpr (COMP) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    acc[mt][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.lo[c], a8, acc[mt][c], 0, 1, 0, 117, 0, 127);
                    acc[mt][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8h[c], alo, acc[mt][c], 1, 1, 0, 127, 0, 117);
                }
            }
        }
        if constexpr (FILL > 0) {
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fill) : "v"(1.0001f));
        }
    };
    // layout 0: `iters` sets of 16 hi MFMAs per wave; layout 1: the same 16 hi MFMAs per wave and set (2 row tiles x 2 column tiles x 4)
    for (int it = 0; it < iters; it += 2) {
        load_wset<NC, COMP>(ws[1], wp, it + 1, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        one_set(ws[0], it);
        __builtin_amdgcn_sched_barrier(0);
        load_wset<NC, COMP>(ws[0], wp, it + 2, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        one_set(ws[1], it + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = fill;
    for (int k = 0; k < NMT; ++k) for (int c = 0; c < NC; ++c) for (int r = 0; r < 16; ++r) s += acc[k][c][r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) { atomicMax(&out[2 * blockIdx.x], m1 - m0); atomicMax(&out[2 * blockIdx.x + 1], r1 - r0); }
}

struct Arm { const char* name; void (*kern)(unsigned long long*, int, float*, const _Float16*, const f16x8*); };
static int arm_threads(const char* n) { return (n[7] == '4') ? 256 : 512; }

int main() {
    unsigned long long* out; float* sink; _Float16* src; f16x8* wp;
    const int grid = 256, iters = 20000;
    const size_t wbytes = (size_t)16 * 8 * 12 * 64 * 16 + 4096;
    (void)hipMalloc(&out, grid * 16); (void)hipMalloc(&sink, 4); (void)hipMalloc(&src, 65536 * 2); (void)hipMalloc(&wp, wbytes);
    std::vector<_Float16> h(65536);
    srand(1);
    for (auto& x : h) { float u = 0, v; for (int i = 0; i < 4; ++i) u += rand() / (float)RAND_MAX; v = (u - 2.f) * 1.7f; x = (_Float16)v; }   // ~N(0,1)
    (void)hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    {
        std::vector<_Float16> w(wbytes / 2);
        for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(0.06f * (float)h[(i * 31 + 7) & 65535]);
        (void)hipMemcpy(wp, w.data(), wbytes, hipMemcpyHostToDevice);
    }
    const size_t lds = BM * RS16 * 2 + BM * RSL;
    const Arm arms[] = {
        {"plain  8 x (32f x 128t)          ", probe<0, false, 0>},  {"plain  2 x 4 x (64f x 64t)       ", probe<1, false, 0>},
        {"plain  8 x (32f x 128t) + 64 VALU", probe<0, false, 64>}, {"plain  2 x 4 x (64f x 64t) + 64 V", probe<1, false, 64>},
        {"plain  4 x (64f x 128t)          ", probe<2, false, 0>},  {"plain  4 x (64f x 128t) + 64 VALU", probe<2, false, 128>},
        {"comp   8 x (32f x 128t)          ", probe<0, true, 0>},   {"comp   2 x 4 x (64f x 64t)       ", probe<1, true, 0>},
        {"comp   8 x (32f x 128t) + 64 VALU", probe<0, true, 64>},  {"comp   2 x 4 x (64f x 64t) + 64 V", probe<1, true, 64>},
        {"comp   4 x (64f x 128t)          ", probe<2, true, 0>},   {"comp   4 x (64f x 128t) + 64 VALU", probe<2, true, 128>},
    };
    const int NA = sizeof(arms) / sizeof(arms[0]);
    for (int a = 0; a < NA; ++a) (void)hipFuncSetAttribute((const void*)arms[a].kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(arms[i & 1].kern, dim3(grid), dim3(512), lds, 0, out, iters, sink, src, wp);   // warm the card
    (void)hipDeviceSynchronize();
    const int ROUNDS = 5;
    std::vector<double> ms[NA], mhz[NA], cyc[NA];
    for (int r = 0; r < ROUNDS; ++r)
        for (int a = 0; a < NA; ++a) {
            hipLaunchKernelGGL(arms[a].kern, dim3(grid), dim3(arm_threads(arms[a].name)), lds, 0, out, iters, sink, src, wp);
            (void)hipMemsetAsync(out, 0, grid * 16, 0);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(arms[a].kern, dim3(grid), dim3(arm_threads(arms[a].name)), lds, 0, out, iters, sink, src, wp);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            std::vector<unsigned long long> o(2 * grid);
            (void)hipMemcpy(o.data(), out, 16 * grid, hipMemcpyDeviceToHost);
            double m = 0, rr = 0; for (int i = 0; i < grid; ++i) m += o[2 * i], rr += o[2 * i + 1];
            ms[a].push_back(t); mhz[a].push_back(m / rr * 100); cyc[a].push_back(m / grid / iters);
        }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double flop = (double)grid * 8 * iters * 16 * 32768.0;   // 16 hi MFMAs per wave and set in either layout
    for (int a = 0; a < NA; ++a)
        std::printf("%s: %.2f ms (min %.2f), %.0f MHz in kernel, %.0f cycles per set per wave, %.0f TFLOP/s (hi products)\n", arms[a].name,
                    med(ms[a]), *std::min_element(ms[a].begin(), ms[a].end()), med(mhz[a]), med(cyc[a]), flop / (med(ms[a]) * 1e-3) / 1e12);
    return 0;
}
