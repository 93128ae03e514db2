// Probe for the next lever on the compensated mode's lo half (HISTORY.md section 8, round 3): the lo product as an FP6 MFMA.
//   v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) / bf6 (e3m2) operands: layout, cycles against the fp8 form, and the error of
//   a . hi (fp16 MFMA) + fp6(a) . fp6(lo) against the exact product; v_cvt_scalef32_pk32_{fp6,bf6}_f16: element order, rounding,
//   saturation, cycles.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_fp6_lo tools/micro/mfma_fp6_lo.cpp && /tmp/mfma_fp6_lo
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

// e2m3 (bias 1: values 0, 0.125 .. 0.875 subnormal; 1 .. 7.5 normal), e3m2 (bias 3: subnormal step 0.0625; max 28); RNE, saturating
static uint8_t to_f6(float x, bool bf6) {
    const int mb = bf6 ? 2 : 3, eb = bf6 ? 3 : 2, bias = bf6 ? 3 : 1;
    const uint8_t s = std::signbit(x) ? 0x20 : 0;
    float a = std::fabs(x);
    const float maxv = bf6 ? 28.f : 7.5f;
    if (a >= maxv) return s | 0x1f;
    int e;
    std::frexp(a, &e);
    int E = e - 1;
    const int Emin = 1 - bias;
    if (a == 0.f || E < Emin) E = Emin;
    int mant = (int)std::nearbyint(std::ldexp(a, mb - E));    // units of 2^(E - mb)
    int be = E + bias;
    if (E == Emin && mant < (1 << mb)) return s | (uint8_t)mant;
    if (mant == (2 << mb)) { mant = 1 << mb; ++be; }
    if (be >= (1 << eb)) return s | 0x1f;
    return s | (uint8_t)(be << mb) | (uint8_t)(mant - (1 << mb));
}
static float from_f6(uint8_t b, bool bf6) {
    const int mb = bf6 ? 2 : 3, bias = bf6 ? 3 : 1;
    const float sg = (b & 0x20) ? -1.f : 1.f;
    const int be = (b & 0x1f) >> mb, m = b & ((1 << mb) - 1);
    if (be == 0) return sg * std::ldexp((float)m, 1 - bias - mb);
    return sg * std::ldexp(1.f + m / (float)(1 << mb), be - bias);
}
// 32 six-bit values -> 6 dwords, value j at bits [6 j, 6 j + 6)
static void pack6(const uint8_t* v, uint32_t* out) {
    std::memset(out, 0, 24);
    for (int j = 0; j < 32; ++j)
        for (int b = 0; b < 6; ++b)
            if (v[j] >> b & 1) out[(6 * j + b) >> 5] |= 1u << ((6 * j + b) & 31);
}

// D = A(32 x 64) . B(64 x 32), both operands six-bit: a6 / b6 [row or col][2 halves][6 dwords]
__global__ void f6_tile(const uint32_t* a6, const uint32_t* b6, float* d, int fmt_a, int fmt_b, int scale_a, int scale_b) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a = {}, b = {};
    for (int j = 0; j < 6; ++j) a[j] = (int)a6[(r * 2 + h) * 6 + j], b[j] = (int)b6[(r * 2 + h) * 6 + j];
    f32x16 acc = {};
    if (fmt_a == 2 && fmt_b == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, scale_a, 0, scale_b);
    else if (fmt_a == 3 && fmt_b == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 3, 2, 0, scale_a, 0, scale_b);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 3, 3, 0, scale_a, 0, scale_b);
    for (int reg = 0; reg < 16; ++reg) d[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
}
// the conversion instruction on 32 halfs per lane
__global__ void cvt_probe(const _Float16* in, uint32_t* out_fp6, uint32_t* out_bf6, float scale) {
    h32 v;
    for (int j = 0; j < 32; ++j) v[j] = in[threadIdx.x * 32 + j];
    const i32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale), r2 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, scale);
    for (int j = 0; j < 6; ++j) out_fp6[threadIdx.x * 6 + j] = (uint32_t)r[j], out_bf6[threadIdx.x * 6 + j] = (uint32_t)r2[j];
}
// accuracy: hi by four fp16 MFMAs + lo by one six-bit MFMA, activations converted in registers (fmt_a: 2 fp6, 3 bf6; weights fp6)
template <int FA>
__global__ void acc_probe(const _Float16* a16, const _Float16* whi, const uint32_t* lo6 /*[col][2][6]*/, const int* scale_lo /*[col][2]*/,
                          float a_scale, int a_scale_e8, float* d_hi, float* d_f6) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 hi = {};
    h32 av;
    for (int s = 0; s < 4; ++s) {
        f16x8 a, bh;
        for (int j = 0; j < 8; ++j) {
            a[j] = a16[r * 64 + 16 * s + 8 * h + j];
            bh[j] = whi[r * 64 + 16 * s + 8 * h + j];
            av[8 * s + j] = a[j];
        }
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, hi, 0, 0, 0);
    }
    i32x6 c6;
    if (FA == 2) c6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(av, a_scale);
    else c6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(av, a_scale);
    i32x8 a = {c6[0], c6[1], c6[2], c6[3], c6[4], c6[5], 0, 0}, b = {};
    for (int j = 0; j < 6; ++j) b[j] = (int)lo6[(r * 2 + h) * 6 + j];
    const int sb = scale_lo[r * 2 + h];
    f32x16 f6 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, hi, FA, 2, 0, a_scale_e8, 0, sb);
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r;
        d_hi[o] = hi[reg], d_f6[o] = f6[reg];
    }
}
template <int KIND>   // 0: fp16 32x32x16, 1: fp8 x fp8, 2: fp6 x fp6, 3: bf6 x fp6, 4: fp6 (A) x fp8 (B), 5: pk32 conversion, 6: fp4 (A) x fp6 (B)
__global__ void rate_probe(unsigned long long* out, int iters, float* sink) {
    f32x16 acc[4] = {};
    f16x8 ah = {}, bh = {};
    i32x8 a8 = {}, b8 = {};
    h32 hv;
    for (int j = 0; j < 32; ++j) hv[j] = (_Float16)(0.01f * (threadIdx.x + j));
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(0.01f * (threadIdx.x + j)); bh[j] = (_Float16)(0.02f * j); a8[j] = 0x08080808 + j; b8[j] = 0x10101010 + threadIdx.x; }
    i32x6 cv = {};
    __syncthreads();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[k & 3], 0, 0, 0);
            else if (KIND == 1) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 0, 0, 0, 127, 0, 120);
            else if (KIND == 2) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 2, 2, 0, 127, 0, 120);
            else if (KIND == 3) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 3, 2, 0, 127, 0, 120);
            else if (KIND == 4) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 2, 0, 0, 127, 0, 120);
            else if (KIND == 6) acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[k & 3], 4, 2, 0, 127, 0, 120);
            else {
                hv[k] = (_Float16)((float)hv[k] + 1.0f);          // a dependency so that the conversions are not hoisted
                const i32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hv, 1.0f);
                for (int j = 0; j < 6; ++j) cv[j] ^= c[j];
            }
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int j = 0; j < 6; ++j) s += (float)cv[j];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&out[blockIdx.x], m1 - m0);
}
// e2m1: 0, 0.5, 1, 1.5, 2, 3, 4, 6; RNE, saturating
static uint8_t to_f4(float x) {
    const uint8_t s = std::signbit(x) ? 8 : 0;
    const float a = std::fabs(x), grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    int best = 0;
    for (int i = 1; i < 8; ++i) {
        const float d0 = std::fabs(a - grid[best]), d1 = std::fabs(a - grid[i]);
        if (d1 < d0 || (d1 == d0 && (i & 1) == 0)) best = i;           // ties to the even mantissa
    }
    return s | (uint8_t)best;
}
static float from_f4(uint8_t b) {
    const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return ((b & 8) ? -1.f : 1.f) * grid[b & 7];
}
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
// the lane's 32 halfs (k order of the four fp16 fragments) -> 32 e2m1 values in 4 dwords, in registers
__device__ inline void frags_to_fp4(const _Float16* p /*32 halfs*/, float scale, int* out4) {
    for (int s = 0; s < 4; ++s) {
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{p[8 * s + 0], p[8 * s + 1]}, scale, 0);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{p[8 * s + 2], p[8 * s + 3]}, scale, 1);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{p[8 * s + 4], p[8 * s + 5]}, scale, 2);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{p[8 * s + 6], p[8 * s + 7]}, scale, 3);
        out4[s] = (int)w;
    }
}
// fp4 activations (converted in registers) x fp6 weights: exact-data layout check and the accuracy probe
__global__ void f4_tile(const _Float16* a16 /*[32][64] in MFMA k order: [row][half][32]*/, const uint32_t* b6, float* d, unsigned* raw) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    int a4[4];
    frags_to_fp4(a16 + (r * 2 + h) * 32, 1.0f, a4);
    for (int j = 0; j < 4; ++j) raw[l * 4 + j] = (unsigned)a4[j];
    i32x8 a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0}, b = {};
    for (int j = 0; j < 6; ++j) b[j] = (int)b6[(r * 2 + h) * 6 + j];
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 2, 0, 127, 0, 124);
    for (int reg = 0; reg < 16; ++reg) d[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
}
__global__ void acc_probe4(const _Float16* a16, const _Float16* whi, const uint32_t* lo6, const int* scale_lo, float a_scale, int a_scale_e8,
                           float* d_hi, float* d_f4) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 hi = {};
    _Float16 av[32];
    for (int s = 0; s < 4; ++s) {
        f16x8 a, bh;
        for (int j = 0; j < 8; ++j) {
            a[j] = a16[r * 64 + 16 * s + 8 * h + j];
            bh[j] = whi[r * 64 + 16 * s + 8 * h + j];
            av[8 * s + j] = a[j];
        }
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, hi, 0, 0, 0);
    }
    int a4[4];
    frags_to_fp4(av, a_scale, a4);
    i32x8 a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0}, b = {};
    for (int j = 0; j < 6; ++j) b[j] = (int)lo6[(r * 2 + h) * 6 + j];
    f32x16 f4 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, hi, 4, 2, 0, a_scale_e8, 0, scale_lo[r * 2 + h]);
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r;
        d_hi[o] = hi[reg], d_f4[o] = f4[reg];
    }
}
template <typename T>
static T* dev(const std::vector<T>& v) {
    T* p;
    (void)hipMalloc(&p, v.size() * sizeof(T));
    (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}
int main() {
    // ---- 1. MFMA layout with exact data (integers 0..7 are exact in e2m3, 0..7 in e3m2 too)
    for (int fmt = 0; fmt < 3; ++fmt) {
        const int fa = fmt == 0 ? 2 : 3, fb = fmt == 2 ? 3 : 2;
        std::vector<uint32_t> a6(32 * 2 * 6), b6(32 * 2 * 6);
        std::vector<float> ref(32 * 32, 0.f);
        for (int i = 0; i < 32; ++i)
            for (int h = 0; h < 2; ++h) {
                uint8_t va[32], vb[32];
                for (int j = 0; j < 32; ++j) {
                    const int k = 32 * h + j;
                    va[j] = to_f6((float)((i + k) % 5), fa == 3), vb[j] = to_f6((float)((2 * i + k) % 7 - 3), fb == 3);
                }
                pack6(va, &a6[(i * 2 + h) * 6]), pack6(vb, &b6[(i * 2 + h) * 6]);
            }
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < 64; ++k) ref[i * 32 + j] += (float)((i + k) % 5) * (float)((2 * j + k) % 7 - 3) * 0.125f;
        float* d;
        (void)hipMalloc(&d, 32 * 32 * 4);
        hipLaunchKernelGGL(f6_tile, dim3(1), dim3(64), 0, 0, dev(a6), dev(b6), d, fa, fb, 127, 124);
        std::vector<float> got(32 * 32);
        (void)hipMemcpy(got.data(), d, got.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32 * 32; ++i) bad += got[i] != ref[i];
        std::printf("[layout] A %s x B %s, k = 32 (lane >> 5) + j, value j at bits [6j, 6j+6) of v[0:5], scale_b 2^-3: %d of 1024 outputs differ (got %g want %g)\n",
                    fa == 2 ? "fp6" : "bf6", fb == 2 ? "fp6" : "bf6", bad, got[5], ref[5]);
    }
    // ---- 2. conversion: order, rounding, saturation
    {
        std::vector<_Float16> in(64 * 32);
        const float tests[32] = {0.f, 0.0625f, 0.125f, 0.1875f, 0.3f, 0.5f, 0.9f, 0.95f, 1.f, 1.0625f, 1.1875f, 1.5f, 2.f, 3.3f, 5.f, 7.5f,
                                 7.7f, 8.f, 20.f, 28.f, 30.f, 100.f, 60000.f, -0.f, -0.3f, -1.0625f, -7.7f, -30.f, 0.03f, 0.0312f, 0.0938f, 6.9f};
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) in[l * 32 + j] = (_Float16)tests[(j + l) % 32];
        uint32_t *o1, *o2;
        (void)hipMalloc(&o1, 64 * 6 * 4), (void)hipMalloc(&o2, 64 * 6 * 4);
        for (float scale : {1.0f, 4.0f}) {
            hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dev(in), o1, o2, scale);
            std::vector<uint32_t> h1(64 * 6), h2(64 * 6);
            (void)hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
            int bad1 = 0, bad2 = 0;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 32; ++j) {
                    const float x = (float)in[l * 32 + j] / scale;
                    auto get = [&](const std::vector<uint32_t>& hh) {
                        uint32_t v = 0;
                        for (int b = 0; b < 6; ++b) v |= ((hh[l * 6 + ((6 * j + b) >> 5)] >> ((6 * j + b) & 31)) & 1u) << b;
                        return (uint8_t)v;
                    };
                    const uint8_t g1 = get(h1), g2 = get(h2), w1 = to_f6(x, false), w2 = to_f6(x, true);
                    if (from_f6(g1, false) != from_f6(w1, false)) { if (bad1 < 6) std::printf("   fp6 scale %g: x = %g -> %g, host RNE+sat gives %g\n", scale, x * scale, from_f6(g1, false), from_f6(w1, false)); ++bad1; }
                    if (from_f6(g2, true) != from_f6(w2, true)) { if (bad2 < 6) std::printf("   bf6 scale %g: x = %g -> %g, host RNE+sat gives %g\n", scale, x * scale, from_f6(g2, true), from_f6(w2, true)); ++bad2; }
                }
            std::printf("[cvt] v_cvt_scalef32_pk32_{fp6,bf6}_f16, scale %g, element j -> bits [6j, 6j+6): %d / %d of 2048 differ from host RNE + saturation of x / scale\n", scale, bad1, bad2);
        }
    }
    // ---- 3. rates
    {
        unsigned long long* out;
        float* sink;
        const int nb = 256, iters = 1000;
        (void)hipMalloc(&out, nb * 8);
        (void)hipMalloc(&sink, 4);
        const char* names[7] = {"v_mfma_f32_32x32x16_f16", "scaled 32x32x64 fp8 x fp8", "scaled 32x32x64 fp6 x fp6", "scaled 32x32x64 bf6 x fp6", "scaled 32x32x64 fp6 x fp8", "v_cvt_scalef32_pk32_bf6_f16", "scaled 32x32x64 fp4 x fp6"};
        for (int kind = 0; kind < 7; ++kind) {
            (void)hipMemset(out, 0, nb * 8);
            switch (kind) {
                case 0: hipLaunchKernelGGL(rate_probe<0>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                case 1: hipLaunchKernelGGL(rate_probe<1>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                case 2: hipLaunchKernelGGL(rate_probe<2>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                case 3: hipLaunchKernelGGL(rate_probe<3>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                case 4: hipLaunchKernelGGL(rate_probe<4>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                case 6: hipLaunchKernelGGL(rate_probe<6>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
                default: hipLaunchKernelGGL(rate_probe<5>, dim3(nb), dim3(256), 0, 0, out, iters, sink); break;
            }
            std::vector<unsigned long long> h(nb);
            (void)hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto t : h) mean += (double)t;
            std::printf("[rate] %-28s: %.1f s_memtime ticks per instruction (one wave per SIMD, 256 CUs busy)\n", names[kind], mean / nb / (16.0 * iters));
        }
    }
    // ---- 4. accuracy: activations N(0,1) (LayerNorm outputs) and a GELU-like set, weights N(0, 0.05^2)
    {
        std::mt19937 rng(7);
        std::normal_distribution<float> na(0.f, 1.f), nw(0.f, 0.05f);
        const int trials = 64;
        float *d_hi, *d_f6;
        (void)hipMalloc(&d_hi, 4096), (void)hipMalloc(&d_f6, 4096);
        for (int cfg = 0; cfg < 6; ++cfg) {
            const int fa = cfg < 3 ? 2 : 3;
            const float a_scale = (cfg % 3 == 0) ? 1.f : (cfg % 3 == 1) ? 2.f : 4.f;
            double e_hi = 0, e_f6 = 0, ref_rms = 0;
            std::mt19937 rg(7);
            for (int t = 0; t < trials; ++t) {
                std::vector<_Float16> a16(32 * 64), whi(32 * 64);
                std::vector<float> w(32 * 64), lo(32 * 64);
                std::vector<uint32_t> lo6(32 * 2 * 6);
                std::vector<int> sc(32 * 2);
                for (int i = 0; i < 32 * 64; ++i) {
                    a16[i] = (_Float16)na(rg);
                    w[i] = nw(rg);
                    whi[i] = (_Float16)w[i];
                    lo[i] = w[i] - (float)whi[i];
                }
                for (int r = 0; r < 32; ++r)
                    for (int h = 0; h < 2; ++h) {
                        float mx = 0.f;
                        uint8_t v[32];
                        for (int s = 0; s < 4; ++s)
                            for (int j = 0; j < 8; ++j) mx = std::fmax(mx, std::fabs(lo[r * 64 + 16 * s + 8 * h + j]));
                        int e;
                        std::frexp(mx, &e);                       // mx = m 2^e, m in [0.5, 1): scaled max in [4, 8) -> below 7.5 after rounding mostly
                        const int S = 3 - e;                      // lo * 2^S
                        for (int s = 0; s < 4; ++s)
                            for (int j = 0; j < 8; ++j) v[8 * s + j] = to_f6(std::ldexp(lo[r * 64 + 16 * s + 8 * h + j], S), false);
                        pack6(v, &lo6[(r * 2 + h) * 6]);
                        sc[r * 2 + h] = 127 - S;
                    }
                const int a_e8 = 127 + (int)std::log2(a_scale);
                if (fa == 2) hipLaunchKernelGGL(acc_probe<2>, dim3(1), dim3(64), 0, 0, dev(a16), dev(whi), dev(lo6), dev(sc), a_scale, a_e8, d_hi, d_f6);
                else hipLaunchKernelGGL(acc_probe<3>, dim3(1), dim3(64), 0, 0, dev(a16), dev(whi), dev(lo6), dev(sc), a_scale, a_e8, d_hi, d_f6);
                std::vector<float> hi(1024), f6(1024);
                (void)hipMemcpy(hi.data(), d_hi, 4096, hipMemcpyDeviceToHost);
                (void)hipMemcpy(f6.data(), d_f6, 4096, hipMemcpyDeviceToHost);
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double ref = 0;
                        for (int k = 0; k < 64; ++k) ref += (double)(float)a16[i * 64 + k] * (double)w[j * 64 + k];
                        const int o = i * 32 + j;
                        e_hi += (hi[o] - ref) * (hi[o] - ref), e_f6 += (f6[o] - ref) * (f6[o] - ref), ref_rms += ref * ref;
                    }
            }
            const double n = trials * 1024.0;
            std::printf("[error] activations %s / scale %g, weights' lo fp6 with a block scale per (column, 32 k): rms exact %.3g; rms error hi only %.3g, hi + six-bit lo %.3g\n",
                        fa == 2 ? "fp6 (e2m3)" : "bf6 (e3m2)", a_scale, std::sqrt(ref_rms / n), std::sqrt(e_hi / n), std::sqrt(e_f6 / n));
        }
    }
    // ---- 5. fp4 activations converted in registers (v_cvt_scalef32_pk_fp4_f16, four per fragment) x fp6 weights
    {
        std::vector<_Float16> a16(32 * 64);
        std::vector<uint32_t> b6(32 * 2 * 6);
        std::vector<float> ref(32 * 32, 0.f);
        const float vals[6] = {0.f, 0.5f, 1.f, 1.5f, 3.f, -2.f};
        for (int i = 0; i < 32; ++i)
            for (int h = 0; h < 2; ++h) {
                uint8_t vb[32];
                for (int j = 0; j < 32; ++j) {
                    const int k = 32 * h + j;
                    a16[(i * 2 + h) * 32 + j] = (_Float16)vals[(i + k) % 6];
                    vb[j] = to_f6((float)((2 * i + k) % 7 - 3), false);
                }
                pack6(vb, &b6[(i * 2 + h) * 6]);
            }
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < 64; ++k) ref[i * 32 + j] += vals[(i + k) % 6] * (float)((2 * j + k) % 7 - 3) * 0.125f;
        float* d;
        unsigned* raw;
        (void)hipMalloc(&d, 4096), (void)hipMalloc(&raw, 64 * 16);
        hipLaunchKernelGGL(f4_tile, dim3(1), dim3(64), 0, 0, dev(a16), dev(b6), d, raw);
        std::vector<float> got(1024);
        std::vector<unsigned> rw(256);
        (void)hipMemcpy(got.data(), d, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(rw.data(), raw, 1024, hipMemcpyDeviceToHost);
        int bad = 0, badc = 0;
        for (int i = 0; i < 1024; ++i) bad += got[i] != ref[i];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const uint8_t g = (rw[l * 4 + (j >> 3)] >> (4 * (j & 7))) & 15;
                badc += from_f4(g) != (float)a16[((l & 31) * 2 + (l >> 5)) * 32 + j];
            }
        std::printf("[fp4] conversion of exact values, element j -> nibble j of v[0:3]: %d of 2048 differ; fp4 (A) x fp6 (B) product: %d of 1024 outputs differ (got %g want %g)\n",
                    badc, bad, got[7], ref[7]);
        const int trials = 64;
        float *d_hi, *d_f4;
        (void)hipMalloc(&d_hi, 4096), (void)hipMalloc(&d_f4, 4096);
        for (float a_scale : {0.5f, 1.f, 2.f}) {
            double e_hi = 0, e_f4 = 0, ref_rms = 0;
            std::mt19937 rg(7);
            std::normal_distribution<float> na(0.f, 1.f), nw(0.f, 0.05f);
            for (int t = 0; t < trials; ++t) {
                std::vector<_Float16> x16(32 * 64), whi(32 * 64);
                std::vector<float> w(32 * 64), lo(32 * 64);
                std::vector<uint32_t> lo6(32 * 2 * 6);
                std::vector<int> sc(32 * 2);
                for (int i = 0; i < 32 * 64; ++i) {
                    x16[i] = (_Float16)na(rg);
                    w[i] = nw(rg);
                    whi[i] = (_Float16)w[i];
                    lo[i] = w[i] - (float)whi[i];
                }
                for (int r = 0; r < 32; ++r)
                    for (int h = 0; h < 2; ++h) {
                        float mx = 0.f;
                        uint8_t v[32];
                        for (int s2 = 0; s2 < 4; ++s2)
                            for (int j = 0; j < 8; ++j) mx = std::fmax(mx, std::fabs(lo[r * 64 + 16 * s2 + 8 * h + j]));
                        int e;
                        std::frexp(mx, &e);
                        const int S = 3 - e;
                        for (int s2 = 0; s2 < 4; ++s2)
                            for (int j = 0; j < 8; ++j) v[8 * s2 + j] = to_f6(std::ldexp(lo[r * 64 + 16 * s2 + 8 * h + j], S), false);
                        pack6(v, &lo6[(r * 2 + h) * 6]);
                        sc[r * 2 + h] = 127 - S;
                    }
                hipLaunchKernelGGL(acc_probe4, dim3(1), dim3(64), 0, 0, dev(x16), dev(whi), dev(lo6), dev(sc), a_scale, 127 + (int)std::log2(a_scale), d_hi, d_f4);
                std::vector<float> hi(1024), f4(1024);
                (void)hipMemcpy(hi.data(), d_hi, 4096, hipMemcpyDeviceToHost);
                (void)hipMemcpy(f4.data(), d_f4, 4096, hipMemcpyDeviceToHost);
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double rf = 0;
                        for (int k = 0; k < 64; ++k) rf += (double)(float)x16[i * 64 + k] * (double)w[j * 64 + k];
                        const int o = i * 32 + j;
                        e_hi += (hi[o] - rf) * (hi[o] - rf), e_f4 += (f4[o] - rf) * (f4[o] - rf), ref_rms += rf * rf;
                    }
            }
            const double n = trials * 1024.0;
            std::printf("[error] activations fp4 (e2m1) / scale %g x weights' lo fp6 (block scale): rms exact %.3g; rms error hi only %.3g, hi + lo %.3g\n",
                        a_scale, std::sqrt(ref_rms / n), std::sqrt(e_hi / n), std::sqrt(e_f4 / n));
        }
    }
    return 0;
}
