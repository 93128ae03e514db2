// fft_core.h -- register-level building blocks of the LDS-resident Stockham FFT used by the Hyena long
// convolution (hyena_conv.hip).  Everything here is plain C++ that also compiles for the host, so the
// index algebra (pass order, twiddles, output permutation, padding) is unit-tested on the CPU
// (tests/test_fft_core.py builds csrc/fft_core_test.cpp with g++).
//
// Reference arithmetic being replaced: HyenaDNA `fftconv` (rfft/irfft of size 2L).
// A linear causal convolution does not care which transform size realises it, so the engine uses a
// power-of-two complex FFT of N >= 2L-2 points with two reads packed as real/imaginary parts.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CLM_HD __host__ __device__ __forceinline__
#define CLM_HDC __host__ __device__ constexpr
#else
#include <cmath>
#define CLM_HD inline
#define CLM_HDC constexpr
struct float2 {
    float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
#endif

namespace clmfft {

CLM_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
CLM_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
CLM_HD float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
CLM_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// multiply by -i (forward transform) or +i (inverse transform)
template <bool INV>
CLM_HD float2 mul_mi(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// LDS index padding: one float2 of padding per 32 elements.  Unit-stride ds_read_b64 (32-lane groups, 64 banks)
// stays conflict-free -- padding per 16 made every such read 2-way -- while the stride-R stores of the first
// (Ns = 1) pass drop from 16-way to 2-way ("long convolution").
CLM_HDC int pad_index(int i) { return i + (i >> 5); }
CLM_HDC int padded_size(int n) { return n + (n >> 5); }

// ---- small DFTs, in place, natural order in and out ------------------------------------------------------
template <bool INV>
CLM_HD void dft2(float2& a0, float2& a1) {
    float2 t = a0;
    a0 = cadd(t, a1);
    a1 = csub(t, a1);
}

template <bool INV>
CLM_HD void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

// exp(-+ 2*pi*i * m / 16), m = 0..9
template <bool INV>
CLM_HD float2 w16(int m) {
    const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, h = 0.70710678118654752440f;
    float cx, sy;
    switch (m) {
        case 0: cx = 1.f; sy = 0.f; break;
        case 1: cx = c1; sy = s1; break;
        case 2: cx = h; sy = h; break;
        case 3: cx = s1; sy = c1; break;
        case 4: cx = 0.f; sy = 1.f; break;
        case 5: cx = -s1; sy = c1; break;
        case 6: cx = -h; sy = h; break;
        case 7: cx = -c1; sy = s1; break;
        case 8: cx = -1.f; sy = 0.f; break;
        default: cx = -c1; sy = -s1; break;  // m = 9
    }
    return make_float2(cx, INV ? sy : -sy);
}

// Generic radix-R DFT on v[0..R-1] (R in {2,4,8,16}); all indices are compile-time after unrolling.
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    static CLM_HD void run(float2* v) { dft2<INV>(v[0], v[1]); }
};
template <bool INV>
struct Dft<4, INV> {
    static CLM_HD void run(float2* v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
};
// R = R1*R2 with n = n1 + R1*n2, k = R2*k1 + k2:
//   Y[n1][k2] = DFT_R2 over n2;  Y *= W_R^(n1*k2);  X[R2*k1+k2] = DFT_R1 over n1.
template <bool INV>
struct Dft<8, INV> {  // R1 = 2, R2 = 4
    static CLM_HD void run(float2* v) {
        dft4<INV>(v[0], v[2], v[4], v[6]);  // n1 = 0: position 0 + 2*k2
        dft4<INV>(v[1], v[3], v[5], v[7]);  // n1 = 1: position 1 + 2*k2
        v[3] = cmul(v[3], w16<INV>(2));     // W8^(1*1)
        v[5] = mul_mi<INV>(v[5]);           // W8^(1*2) = -+i
        v[7] = cmul(v[7], w16<INV>(6));     // W8^(1*3)
        dft2<INV>(v[0], v[1]);              // k2 = 0 -> X[0], X[4]
        dft2<INV>(v[2], v[3]);              // k2 = 1 -> X[1], X[5]
        dft2<INV>(v[4], v[5]);              // k2 = 2 -> X[2], X[6]
        dft2<INV>(v[6], v[7]);              // k2 = 3 -> X[3], X[7]
        // position p = k1 + 2*k2 holds X[4*k1 + k2]  ->  natural order
        float2 x1 = v[2], x2 = v[4], x3 = v[6], x4 = v[1], x5 = v[3], x6 = v[5];
        v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
    }
};
template <bool INV>
struct Dft<16, INV> {  // R1 = R2 = 4
    static CLM_HD void run(float2* v) {
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) dft4<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);  // pos n1 + 4*k2
#pragma unroll
        for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = cmul(v[n1 + 4 * k2], w16<INV>(n1 * k2));
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) dft4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
        // position p = k1 + 4*k2 holds X[4*k1 + k2]: transpose the 4x4 register grid
        float2 t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) t[p] = v[p];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v[4 * k1 + k2] = t[k1 + 4 * k2];
    }
};

// v[r] *= w^r for r = 1..R-1 with a multiplication tree of depth <= 4 (error ~ 4 ulp instead of R ulp)
template <int R>
CLM_HD void apply_twiddle_powers(float2* v, float2 w1) {
    if (R >= 2) v[1] = cmul(v[1], w1);
    if (R >= 4) {
        float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        v[2] = cmul(v[2], w2);
        v[3] = cmul(v[3], w3);
        if (R >= 8) {
            float2 w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            v[4] = cmul(v[4], w4);
            v[5] = cmul(v[5], w5);
            v[6] = cmul(v[6], w6);
            v[7] = cmul(v[7], w7);
            if (R >= 16) {
                float2 w8 = cmul(w4, w4);
                v[8] = cmul(v[8], w8);
                v[9] = cmul(v[9], cmul(w8, w1));
                v[10] = cmul(v[10], cmul(w8, w2));
                v[11] = cmul(v[11], cmul(w8, w3));
                v[12] = cmul(v[12], cmul(w8, w4));
                v[13] = cmul(v[13], cmul(w8, w5));
                v[14] = cmul(v[14], cmul(w8, w6));
                v[15] = cmul(v[15], cmul(w8, w7));
            }
        }
    }
}

// ---- transform plan ------------------------------------------------------------------------------------
// Forward radices for N = 2^LOGN (8 <= LOGN <= 14); the inverse runs them in reverse order so that the last
// forward pass and the first inverse pass use the same radix and can be fused in registers.
template <int LOGN>
struct Plan {
    static constexpr int N = 1 << LOGN;
    static constexpr int NPASS = (LOGN + 3) / 4;
    static constexpr int LAST = (LOGN % 4 == 0) ? 16 : (1 << (LOGN % 4));  // radix of the last forward pass
    static constexpr int NT = (N / 32 < 64) ? 64 : N / 32;                 // threads per transform (32 points each)
    static constexpr int radix(int pass) { return pass < NPASS - 1 ? 16 : LAST; }
};

// One Stockham butterfly, register part: twiddle by exp(-+2*pi*i*k*r/(Ns*R)) then DFT_R.
// `tw` is the table exp(-2*pi*i*m/N), m < N/2 (forward sign); Ns = product of the radices already applied.
template <int LOGN, int R, bool INV>
CLM_HD void butterfly(float2* v, int jb, int Ns, const float2* tw) {
    constexpr int N = 1 << LOGN;
    if (Ns > 1) {
        int k = jb & (Ns - 1);
        float2 w1 = tw[k * (N / (Ns * R))];
        if (INV) w1 = cconj(w1);
        apply_twiddle_powers<R>(v, w1);
    }
    Dft<R, INV>::run(v);
}

// same with the twiddle already in a register (prefetched at kernel start)
template <int R, bool INV>
CLM_HD void butterfly_w(float2* v, bool has_tw, float2 w1) {
    if (has_tw) apply_twiddle_powers<R>(v, w1);
    Dft<R, INV>::run(v);
}
template <int LOGN, int R, bool INV>
CLM_HD float2 twiddle_for(int jb, int Ns, const float2* tw) {
    constexpr int N = 1 << LOGN;
    if (Ns <= 1) return make_float2(1.f, 0.f);
    float2 w1 = tw[(jb & (Ns - 1)) * (N / (Ns * R))];
    return INV ? cconj(w1) : w1;
}

// element index read by butterfly jb for input r, and written for output q
template <int LOGN, int R>
CLM_HD int stockham_in(int jb, int r) {
    return jb + r * ((1 << LOGN) / R);
}
template <int R>
CLM_HD int stockham_out(int jb, int q, int Ns) {
    int k = jb & (Ns - 1);
    return (jb - k) * R + k + q * Ns;
}

}  // namespace clmfft
