// Microbenchmark (round 5; cdna_hip_programming.md rule 28, MI355X_MICROARCH.md 'DVFS give-back' item 7):
// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 at the SAME output tile per wave (32 features x 128 tokens, the tail
// kernel's), the activation operand re-read from an LDS tile by ds_read_b128 exactly as compute_tm does (row stride 528 B), the
// weight fragments in registers, 8 waves per CU on all CUs, random data.  Wall time, in-kernel shader clock (s_memtime against the
// 100-MHz s_memrealtime) and TFLOP/s, back to back and at the MFMA duty the tail kernel runs at (FILL VALU instructions + a sleep
// after every 64-deep set, identical in both arms).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/micro/mfma_shape.cpp && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RS16 = 264, BM = 128;

template <int SHAPE, int FILL, int SLEEP>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, float* sink, const _Float16* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    for (int i = threadIdx.x; i < BM * RS16; i += 512) As[i] = src[(i * 7 + blockIdx.x * 131) & 65535];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 w[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) w[s][i] = src[(threadIdx.x * 32 + s * 8 + i + wave * 977) & 65535];
    __syncthreads();
    float fill = (float)lane;
    f32x16 acc32[4] = {};
    f32x4 acc16[8][2] = {};
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const int kb = (it & 3) * 64;                           // the 64-deep group of the 256-wide tile
        if constexpr (SHAPE == 0) {
            const _Float16* a0 = As + (lane & 31) * RS16 + kb + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mt = i >> 2, ks = i & 3;
                const f16x8 a = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * RS16 + ks * 16);
                acc32[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], a, acc32[mt], 0, 0, 0);
            }
        } else {
            // lane l: token l & 15, k chunk l >> 4 (8 halfs) of the 32-deep step.  The 16-byte chunks of a row are stored swizzled
            // (chunk ^ 1 for rows 4..11 of every 16) so that each of ds_read_b128's four 16-lane groups covers 16 bank slots.
            const int r = lane & 15, c = (lane >> 4) ^ (((r >> 2) ^ (r >> 3)) & 1);
            const _Float16* a0 = As + r * RS16 + kb + c * 8;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tt = i >> 1, k2 = i & 1;
                const f16x8 a = *reinterpret_cast<const f16x8*>(a0 + tt * 16 * RS16 + k2 * 32);
                acc16[tt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[k2], a, acc16[tt][0], 0, 0, 0);
                acc16[tt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 + k2], a, acc16[tt][1], 0, 0, 0);
            }
        }
        if constexpr (FILL > 0) {
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fill) : "v"(1.0001f));
        }
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = fill;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc32[k][r];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) s += acc16[k][0][r] + acc16[k][1][r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) { atomicMax(&out[2 * blockIdx.x], m1 - m0); atomicMax(&out[2 * blockIdx.x + 1], r1 - r0); }
}

struct Arm { const char* name; void (*kern)(unsigned long long*, int, float*, const _Float16*); };

int main() {
    unsigned long long* out; float* sink; _Float16* src;
    const int grid = 256, iters = 40000;
    (void)hipMalloc(&out, grid * 16); (void)hipMalloc(&sink, 4); (void)hipMalloc(&src, 65536 * 2);
    std::vector<_Float16> h(65536);
    srand(1);
    for (auto& x : h) { float u = 0, v; for (int i = 0; i < 4; ++i) u += rand() / (float)RAND_MAX; v = (u - 2.f) * 1.7f; x = (_Float16)v; }   // ~N(0,1)
    (void)hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    const size_t lds = BM * RS16 * 2;
    const Arm arms[] = {
        {"32x32x16 back-to-back       ", probe<0, 0, 0>},  {"16x16x32 back-to-back       ", probe<1, 0, 0>},
        {"32x32x16 + 64 VALU / set    ", probe<0, 64, 0>}, {"16x16x32 + 64 VALU / set    ", probe<1, 64, 0>},
        {"32x32x16 + 64 VALU + sleep 8", probe<0, 64, 8>}, {"16x16x32 + 64 VALU + sleep 8", probe<1, 64, 8>},
    };
    const int NA = sizeof(arms) / sizeof(arms[0]);
    for (int a = 0; a < NA; ++a) (void)hipFuncSetAttribute((const void*)arms[a].kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // warm the card: ~2 s of launches before anything is recorded
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(arms[i & 1].kern, dim3(grid), dim3(512), lds, 0, out, iters, sink, src);
    (void)hipDeviceSynchronize();
    const int ROUNDS = 5;
    std::vector<double> ms[NA], mhz[NA], cyc[NA];
    for (int r = 0; r < ROUNDS; ++r)
        for (int a = 0; a < NA; ++a) {
            (void)hipMemset(out, 0, grid * 16);
            // two launches back to back, the second one timed
            hipLaunchKernelGGL(arms[a].kern, dim3(grid), dim3(512), lds, 0, out, iters, sink, src);
            (void)hipMemsetAsync(out, 0, grid * 16, 0);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(arms[a].kern, dim3(grid), dim3(512), lds, 0, out, iters, sink, src);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            std::vector<unsigned long long> o(2 * grid);
            (void)hipMemcpy(o.data(), out, 16 * grid, hipMemcpyDeviceToHost);
            double m = 0, rr = 0; for (int i = 0; i < grid; ++i) m += o[2 * i], rr += o[2 * i + 1];
            ms[a].push_back(t); mhz[a].push_back(m / rr * 100); cyc[a].push_back(m / grid / iters);
        }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double flop = (double)grid * 8 * iters * 16 * 32768.0;   // 16 x (32x32x16) MFMAs = 32 x (16x16x32) per set, same FLOP
    for (int a = 0; a < NA; ++a)
        std::printf("%s: %.2f ms (min %.2f), %.0f MHz in kernel, %.0f cycles per 64-deep set per wave, %.0f TFLOP/s\n", arms[a].name,
                    med(ms[a]), *std::min_element(ms[a].begin(), ms[a].end()), med(mhz[a]), med(cyc[a]), flop / (med(ms[a]) * 1e-3) / 1e12);
    return 0;
}
