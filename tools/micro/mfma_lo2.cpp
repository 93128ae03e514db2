// Round 4: the lo half of an ACTIVATION operand of the compensated mode (gemm_common.h lo8_pack4 / mfma_lo2), checked in isolation
// with the kernel's own helpers: 64-deep dot products of fp32 N(0,1)-ish activations with fp32 N(0, 0.05^2) weights, as
//   (1) fp16(a) . fp16(w)                                    -- the plain fp16 mode
//   (2) (1) + e5m2t(fp16(a)) . e4m3((w - hi) 2^17 / 0.9155)   -- fp16c through round 3: weights compensated
//   (3) (2) + e5m2((a - fp16(a)) 2^10 / 0.9155) . e5m2t(w_hi)  -- round 4: activations compensated too
// against the exact (double) product of the fp32 inputs.  Also: lo8_unpack4(lo8_pack4(x)) + fp16(x) against x.
//   hipcc --offload-arch=gfx950 -O3 -I chimeralm_amd/csrc -I include -o /tmp/mfma_lo2 tools/micro/mfma_lo2.cpp && /tmp/mfma_lo2
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "gemm_common.h"
using namespace clm;

// OCP e4m3fn, round to nearest even, saturating (as pack_weight_split_kernel produces the weights' lo bytes)
static unsigned char to_e4m3(float x) {
    const unsigned char s = std::signbit(x) ? 0x80 : 0;
    float a = std::fabs(x);
    if (!(a == a)) return 0x7f;
    if (a >= 448.f) return s | 0x7e;
    if (a < std::ldexp(1.f, -10)) return s;
    int e;
    std::frexp(a, &e);
    int E = e - 1;
    if (E < -6) E = -6;
    int mant = (int)std::nearbyint(std::ldexp(a, 3 - E));
    int be = E + 7;
    if (E == -6 && mant < 8) return s | (unsigned char)mant;
    if (mant == 16) { mant = 8; ++be; }
    if (be > 15 || (be == 15 && mant - 8 > 6)) return s | 0x7e;
    return s | (unsigned char)(be << 3) | (unsigned char)(mant - 8);
}

// a32 [32 rows][64] fp32, whi [32 cols][64] halfs, wlo8 [32][64] bytes in the MFMA's k order; outputs [32][32]
__global__ void probe(const float* a32, const _Float16* whi, const unsigned char* wlo8, float* d1, float* d2, float* d3, float* rt) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {};
    i32x8 a8 = {}, w8h = {}, alo = {}, w8;
    __builtin_memcpy(&w8, wlo8 + r * 64 + 32 * h, 32);
    float worst = 0.f;
    for (int s = 0; s < 4; ++s) {
        u16x8 af, wf;
        unsigned lo[2];
        for (int jj = 0; jj < 2; ++jj) {
            const float* p = a32 + r * 64 + 16 * s + 8 * h + 4 * jj;
            u16x4 hb;
            lo[jj] = lo8_pack4(p[0], p[1], p[2], p[3], hb);
            for (int e = 0; e < 4; ++e) af[4 * jj + e] = hb[e];
            float d[4];
            lo8_unpack4(lo[jj], d);
            for (int e = 0; e < 4; ++e) {
                f16_t hh; hh.bits = hb[e];
                worst = fmaxf(worst, fabsf(to_float(hh) + d[e] - p[e]) / fmaxf(fabsf(p[e]), 1e-3f));
            }
        }
        for (int j = 0; j < 8; ++j) wf[j] = __builtin_bit_cast(unsigned short, whi[r * 64 + 16 * s + 8 * h + j]);
        alo[2 * s] = (int)lo[0], alo[2 * s + 1] = (int)lo[1];
        acc = mfma<PREC_F16C>(af, wf, acc);
        int x0, x1;
        frag_to_e5m2t(af, x0, x1);
        a8[2 * s] = x0, a8[2 * s + 1] = x1;
        frag_to_e5m2t(wf, x0, x1);
        w8h[2 * s] = x0, w8h[2 * s + 1] = x1;
    }
    const f32x16 c1 = acc;
    const f32x16 c2 = mfma_lo8<false>(w8, a8, c1);
    const f32x16 c3 = mfma_lo2<false>(w8h, alo, c2);
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r;
        d1[o] = c1[reg], d2[o] = c2[reg], d3[o] = c3[reg];
    }
    atomicMax(reinterpret_cast<int*>(rt), __float_as_int(worst));
}

template <typename T>
static T* dev(const std::vector<T>& v) {
    T* p;
    (void)hipMalloc(&p, v.size() * sizeof(T));
    (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main() {
    std::mt19937 rng(11);
    std::normal_distribution<float> na(0.f, 1.f), nw(0.f, 0.05f);
    const int trials = 64;
    double e1 = 0, e2 = 0, e3 = 0, ref_rms = 0;
    float *d1, *d2, *d3, *rt;
    (void)hipMalloc(&d1, 4096), (void)hipMalloc(&d2, 4096), (void)hipMalloc(&d3, 4096), (void)hipMalloc(&rt, 4);
    (void)hipMemset(rt, 0, 4);
    for (int t = 0; t < trials; ++t) {
        std::vector<float> a(32 * 64), w(32 * 64);
        std::vector<_Float16> whi(32 * 64);
        std::vector<unsigned char> lo8(32 * 64);
        for (int i = 0; i < 32 * 64; ++i) {
            a[i] = na(rng) * (t % 4 == 3 ? 30.f : 1.f);          // every fourth trial: large activations
            w[i] = nw(rng);
            whi[i] = (_Float16)w[i];
        }
        for (int r = 0; r < 32; ++r)
            for (int h = 0; h < 2; ++h)
                for (int s = 0; s < 4; ++s)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * s + 8 * h + j;
                        lo8[r * 64 + 32 * h + 8 * s + j] = to_e4m3((w[r * 64 + k] - (float)whi[r * 64 + k]) * LO8_SCALE * LO8_TRUNC_GAIN);
                    }
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dev(a), dev(whi), dev(lo8), d1, d2, d3, rt);
        std::vector<float> g1(1024), g2(1024), g3(1024);
        (void)hipMemcpy(g1.data(), d1, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(g2.data(), d2, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(g3.data(), d3, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)a[i * 64 + k] * (double)w[j * 64 + k];
                const int o = i * 32 + j;
                e1 += (g1[o] - ref) * (g1[o] - ref), e2 += (g2[o] - ref) * (g2[o] - ref), e3 += (g3[o] - ref) * (g3[o] - ref);
                ref_rms += ref * ref;
            }
    }
    float worst;
    (void)hipMemcpy(&worst, rt, 4, hipMemcpyDeviceToHost);
    const double n = trials * 1024.0;
    std::printf("[lo2] 64-deep products of fp32 activations x fp32 weights, rms of the exact value %.3g; rms error: fp16 x fp16 %.3g, + weights' lo %.3g, "
                "+ activations' lo %.3g\n", std::sqrt(ref_rms / n), std::sqrt(e1 / n), std::sqrt(e2 / n), std::sqrt(e3 / n));
    std::printf("[lo2] fp16(x) + unpack(pack(x)) against x: worst relative error %.3g (2^-12 = %.3g for fp16 alone)\n", worst, std::ldexp(1.0, -12));
    return 0;
}
