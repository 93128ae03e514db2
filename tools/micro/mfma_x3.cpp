// Round 4: a 64-deep product of the fp16x3 mode in isolation (csrc/tail32.hip AR_X3), with the arithmetic the kernels use:
// fp32 N(0,1)-ish activations x fp32 N(0, 0.05^2) weights, each split as hi = fp16(x), lo = fp16(x - hi) (weights pre-scaled by
// 2^10), as
//   (1) fp16(a) . fp16(w)                                      -- one fp16 MFMA per 16-deep step (the plain fp16 mode)
//   (2) a_hi . w_hi + a_hi . w_lo + a_lo . w_hi                 -- fp16x3: three fp16 MFMAs per step, one fp32 accumulator
//   (3) v_mfma_f32_32x32x2_f32 on the fp32 operands             -- the exact mode
// against the exact (double) product of the fp32 inputs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_x3 tools/micro/mfma_x3.cpp && /tmp/mfma_x3
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
constexpr float WS = 1024.f, WSI = 1.0f / 1024.f;

// a32 [32 tokens][64], w32 [32 features][64]; outputs [feature][token]
__global__ void probe(const float* a32, const float* w32, float* d1, float* d2, float* d3) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 c1 = {}, c2 = {}, c3 = {};
    for (int s = 0; s < 4; ++s) {
        f16x8 ah, al, wh, wl, w1;
        for (int j = 0; j < 8; ++j) {
            const float a = a32[r * 64 + 16 * s + 8 * h + j], w = w32[r * 64 + 16 * s + 8 * h + j];
            ah[j] = (_Float16)a;
            al[j] = (_Float16)(a - (float)ah[j]);
            const float ws = w * WS;
            wh[j] = (_Float16)ws;
            wl[j] = (_Float16)(ws - (float)wh[j]);
            w1[j] = (_Float16)w;
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, ah, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, c2, 0, 0, 0);
    }
    for (int k = 0; k < 64; k += 2)
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w32[r * 64 + k + h], a32[r * 64 + k + h], c3, 0, 0, 0);
    for (int q = 0; q < 16; ++q) {
        const int feat = (q & 3) + 8 * (q >> 2) + 4 * h;     // accumulator row; column = token r
        d1[feat * 32 + r] = c1[q];
        d2[feat * 32 + r] = c2[q] * WSI;
        d3[feat * 32 + r] = c3[q];
    }
}

int main() {
    std::mt19937 g(7);
    std::normal_distribution<float> na(0.f, 1.f), nw(0.f, 0.05f);
    double e1 = 0, e2 = 0, e3 = 0, ref2 = 0;
    const int trials = 64;
    float *da, *dw, *d1, *d2, *d3;
    (void)hipMalloc(&da, 32 * 64 * 4); (void)hipMalloc(&dw, 32 * 64 * 4);
    (void)hipMalloc(&d1, 4096); (void)hipMalloc(&d2, 4096); (void)hipMalloc(&d3, 4096);
    for (int t = 0; t < trials; ++t) {
        std::vector<float> a(32 * 64), w(32 * 64), o1(1024), o2(1024), o3(1024);
        for (auto& x : a) x = na(g) * (t % 4 == 3 ? 30.f : 1.f);          // every fourth trial: large activations (the y tile's range)
        for (auto& x : w) x = nw(g);
        (void)hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dw, d1, d2, d3);
        (void)hipMemcpy(o1.data(), d1, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o2.data(), d2, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o3.data(), d3, 4096, hipMemcpyDeviceToHost);
        for (int f = 0; f < 32; ++f)
            for (int tk = 0; tk < 32; ++tk) {
                double ex = 0;
                for (int k = 0; k < 64; ++k) ex += (double)w[f * 64 + k] * (double)a[tk * 64 + k];
                e1 += (o1[f * 32 + tk] - ex) * (o1[f * 32 + tk] - ex);
                e2 += (o2[f * 32 + tk] - ex) * (o2[f * 32 + tk] - ex);
                e3 += (o3[f * 32 + tk] - ex) * (o3[f * 32 + tk] - ex);
                ref2 += ex * ex;
            }
    }
    const double n = (double)trials * 1024;
    std::printf("[x3] 64-deep products of fp32 activations x fp32 weights, rms of the exact value %.3g; rms error: fp16 x fp16 %.3g, "
                "fp16x3 %.3g, fp32 MFMA %.3g\n", std::sqrt(ref2 / n), std::sqrt(e1 / n), std::sqrt(e2 / n), std::sqrt(e3 / n));
    return 0;
}
