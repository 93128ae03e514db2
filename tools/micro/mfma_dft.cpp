// Round 4, VERDICT r03 item 2: PRICE an MFMA formulation of the long convolution's transform before building (or burying) it.
//
// The 16384-point complex transform of one convolution unit (two reads packed as re / im) as Stockham stages whose butterflies are
// DFT MATRICES on the MFMA: radix 32, 32, 16 (8 N R real FLOP per radix-R stage: 10.5 MFLOP forward, 21 with the inverse).  fp16
// MFMA operands carry 11 bits and y must be good to ~1e-5, so both the DFT matrix and the data are split hi + lo
// and every product is THREE `v_mfma_f32_32x32x16_f16` (F_hi x_hi + F_lo x_hi + F_hi x_lo): 63 MFLOP of fp16 MFMA per unit, 15.4k
// cycles of the CU's four MFMA pipes at 100 % -- against ~49k cycles per unit for the VALU / LDS kernel in the product.
//
// This probe runs ONE such stage, complete, the way a kernel would: data in LDS as fp32 SoA (re | im, 128 KiB, as the product's
// buffer), a radix-32 stage over the 512 columns j of x[r][j] = x[512 r + j]:
//     y[k][j] = W_N^(k j) / 8 * sum_r W_32^(k r) x[r][j]                       (scaled by 1/8: rms changes by sqrt(32) / 8)
// as the real product  [Yr; Yi] (64 x 512) = [[Fr, -Fi], [Fi, Fr]] (64 x 64) . [Xr; Xi] (64 x 512):
//   * a wave owns two 32-column tiles x both row tiles (re / im of an element in the same lane and register: the twiddle is local);
//   * B fragments: 8 LDS words per lane (stride 512: lanes = consecutive columns), split hi / lo in registers;
//   * A fragments (the DFT matrix, hi + lo): 64 registers, built once;
//   * epilogue: complex twiddle from a table, results written back IN PLACE (a wave reads and writes its own columns only).
// Reported: s_memtime ticks per stage per workgroup (512 threads, one per CU on all CUs, the stage repeated), the stage's MFMA pipe
// time, and the error of one stage against a double-precision DFT.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dft tools/micro/mfma_dft.cpp && /tmp/mfma_dft
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N = 16384, R = 32, J = N / R, NT = 512;

__device__ __forceinline__ void split8(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)(v[e] - (float)h);
    }
}

// MODE 0: the whole stage; 1: no MFMAs (operand delivery + epilogue only); 2: MFMAs only (operands built once)
template <int MODE>
__global__ __launch_bounds__(NT) void dft_stage(const float* __restrict__ xin, float* __restrict__ xout, const float2* __restrict__ tw,
                                                int iters, float scale, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xr = lds;
    float* xi = lds + N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = lane & 31, h = lane >> 5;
    for (int i = tid; i < N; i += NT) xr[i] = xin[i], xi[i] = xin[N + i];
    // A fragments: A[m][kk], m = 32 mt + l (mt 0: real rows, 1: imaginary rows), kk = 16 ks + 8 h + e; kk < 32: real inputs
    f16x8 ah[2][4], al[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * ks + 8 * h + e, c = kk >> 5, r = kk & 31;
                float s, co;
                sincospif(-2.0f * (float)((l * r) & 31) / 32.0f, &s, &co);      // W_32^(k r) = co + i s
                // [Yr; Yi] = [[Fr, -Fi], [Fi, Fr]] [Xr; Xi]
                v[e] = scale * (mt == 0 ? (c == 0 ? co : -s) : (c == 0 ? s : co));
            }
            split8(v, ah[mt][ks], al[mt][ks]);
        }
    __syncthreads();
    f16x8 bh[4], bl[4];
    if (MODE == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xr[(8 * h + e) * J + l + ks];
            split8(v, bh[ks], bl[ks]);
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 32 * (2 * wave + q) + l;
            f32x16 acc[2] = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (MODE != 2) {
                    const float* src = (ks >> 1) ? xi : xr;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = src[(16 * (ks & 1) + 8 * h + e) * J + j];
                    split8(v, bh[ks], bl[ks]);
                }
                if (MODE != 1) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt][ks], bh[ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt][ks], bh[ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt][ks], bl[ks], acc[mt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[ks & 1][e] += (float)bh[ks][e] + (float)bl[ks][e];
                }
            }
            if (MODE != 2) {
                // epilogue: row k = (reg & 3) + 8 (reg >> 2) + 4 h, column j: twiddle W_N^(k j), written back in place
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int k = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const float2 w = tw[k * J + j];
                    const float re = acc[0][reg], im = acc[1][reg];
                    xr[k * J + j] = re * w.x - im * w.y;
                    xi[k * J + j] = re * w.y + im * w.x;
                }
            } else {
                if (acc[0][0] == 123.456f) xr[j] = acc[1][3];
            }
        }
        __syncthreads();                    // (the next stage reads other waves' columns)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    for (int i = tid; i < N; i += NT) xout[(size_t)blockIdx.x * 2 * N + i] = xr[i], xout[(size_t)blockIdx.x * 2 * N + N + i] = xi[i];
}

int main() {
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(2 * N);
    for (auto& v : x) v = nd(rng);
    std::vector<float2> tw((size_t)R * J);
    for (int k = 0; k < R; ++k)
        for (int j = 0; j < J; ++j) {
            const double a = -2.0 * M_PI * (double)((k * j) % N) / N;
            tw[(size_t)k * J + j] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    float *dx, *dy;
    float2* dtw;
    unsigned long long* dt;
    const int nb = 256;
    (void)hipMalloc(&dx, x.size() * 4), (void)hipMalloc(&dy, (size_t)nb * 2 * N * 4), (void)hipMalloc(&dtw, tw.size() * 8), (void)hipMalloc(&dt, nb * 8);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)2 * N * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft_stage<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft_stage<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft_stage<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // ---- 1. one stage against double precision (scale 1/8 exact)
    {
        hipLaunchKernelGGL(dft_stage<0>, dim3(1), dim3(NT), lds, 0, dx, dy, dtw, 1, 0.125f, dt);
        std::vector<float> y(2 * N);
        (void)hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0, se = 0, sr = 0;
        for (int k = 0; k < R; ++k)
            for (int j = 0; j < J; ++j) {
                std::complex<double> s = 0;
                for (int r = 0; r < R; ++r)
                    s += std::polar(1.0, -2.0 * M_PI * ((k * r) % R) / R) * std::complex<double>(x[r * J + j], x[N + r * J + j]);
                s *= 0.125 * std::polar(1.0, -2.0 * M_PI * (double)((k * j) % N) / N);
                const double er = y[k * J + j] - s.real(), ei = y[N + k * J + j] - s.imag();
                maxerr = std::fmax(maxerr, std::fmax(std::fabs(er), std::fabs(ei)));
                maxref = std::fmax(maxref, std::abs(s));
                se += er * er + ei * ei, sr += std::norm(s);
            }
        std::printf("[error] one radix-32 stage (3 fp16 MFMAs per product, fp32 twiddle): max |err| %.3g of max |y| %.3g = %.3g relative; rms relative %.3g"
                    "  (budget for y: ~1e-5 after six stages)\n", maxerr, maxref, maxerr / maxref, std::sqrt(se / sr));
    }
    // ---- 2. time: 256 workgroups (one per CU), the stage repeated
    const int iters = 200;
    const float keep = 1.0f / std::sqrt(32.0f);                 // keeps the rms constant over the repetitions (timing only)
    const char* names[3] = {"whole stage", "no MFMAs (LDS reads + hi/lo split + twiddle + LDS writes)", "MFMAs only (48 per wave)"};
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(dft_stage<0>, dim3(nb), dim3(NT), lds, 0, dx, dy, dtw, iters, keep, dt);
        else if (mode == 1) hipLaunchKernelGGL(dft_stage<1>, dim3(nb), dim3(NT), lds, 0, dx, dy, dtw, iters, keep, dt);
        else hipLaunchKernelGGL(dft_stage<2>, dim3(nb), dim3(NT), lds, 0, dx, dy, dtw, iters, keep, dt);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(nb);
        (void)hipMemcpy(t.data(), dt, nb * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : t) mean += (double)v;
        mean /= nb * (double)iters;
        std::printf("[time] %-62s %8.0f s_memtime ticks per radix-32 stage and workgroup (100 MHz x ?: see clock), %.2f us wall per stage\n",
                    names[mode], mean, ms * 1e3 / iters);
    }
    std::printf("[price] a 16384-point convolution = 2 x (radix 32, 32, 16) stages = ~5 radix-32 stage equivalents + the spectrum product and the\n"
                "        gating phases the product kernel has today (~13k of its ~49k cycles per unit); MFMA pipe time alone: 48 x 32 cycles per\n"
                "        wave and stage = 3,072 cycles per SIMD with two waves.\n");
    return 0;
}
