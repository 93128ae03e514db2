// fft_core.h -- register-level building blocks of the LDS-resident Stockham FFT used by the Hyena long
// convolution (hyena_conv.hip).  Everything here is plain C++ that also compiles for the host, so the
// index algebra (pass order, twiddles, output permutation, padding) is unit-tested on the CPU
// (tests/test_fft_core.py builds csrc/fft_core_test.cpp with g++).
//
// Reference arithmetic being replaced: HyenaDNA `fftconv` (rfft/irfft of size 2L).
// A linear causal convolution does not care which transform size realises it, so the engine uses a
// power-of-two complex FFT of N >= 2L-2 points with two reads packed as real/imaginary parts.
//
// Butterflies are written over an abstract complex type C.  `Cx2` carries TWO butterflies per thread in
// structure-of-arrays form -- re = (re_a, re_b), im = (im_a, im_b) as 2-wide float vectors -- so that every
// complex add is 2 and every complex multiply 4 packed instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
// for two butterflies, with no lane or register swizzles at all (the array-of-structs form let hipcc emit packed
// math too, but 27 % of its instruction stream were v_mov shuffles).  `Cx1` is the scalar form (host tests, odd counts).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CLM_HD __host__ __device__ __forceinline__
#define CLM_HDC __host__ __device__ constexpr
namespace clmfft {
using V2 = float __attribute__((ext_vector_type(2)));
CLM_HD V2 make_v2(float a, float b) { return V2{a, b}; }
CLM_HD V2 splat2(float a) { return V2{a, a}; }
}  // namespace clmfft
#else
#include <cmath>
#define CLM_HD inline
#define CLM_HDC constexpr
struct float2 {
    float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
namespace clmfft {
struct V2 {
    float x, y;
};
inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
inline V2 operator-(V2 a) { return V2{-a.x, -a.y}; }
inline V2 make_v2(float a, float b) { return V2{a, b}; }
inline V2 splat2(float a) { return V2{a, a}; }
}  // namespace clmfft
#endif

namespace clmfft {

CLM_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
CLM_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
CLM_HD float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
CLM_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// ---- the two complex carriers ----------------------------------------------------------------------------
struct Cx1 {  // one complex number
    float re, im;
    static CLM_HD Cx1 add(Cx1 a, Cx1 b) { return Cx1{a.re + b.re, a.im + b.im}; }
    static CLM_HD Cx1 sub(Cx1 a, Cx1 b) { return Cx1{a.re - b.re, a.im - b.im}; }
    static CLM_HD Cx1 mul(Cx1 a, Cx1 b) { return Cx1{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
    static CLM_HD Cx1 mulc(Cx1 a, float cr, float ci) { return Cx1{a.re * cr - a.im * ci, a.re * ci + a.im * cr}; }
    static CLM_HD Cx1 conj(Cx1 a) { return Cx1{a.re, -a.im}; }
    static CLM_HD Cx1 neg(Cx1 a) { return Cx1{-a.re, -a.im}; }
    template <bool INV>
    static CLM_HD Cx1 mul_mi(Cx1 a) { return INV ? Cx1{-a.im, a.re} : Cx1{a.im, -a.re}; }   // * (+i | -i)
};
struct Cx2 {  // two complex numbers, SoA: (a, b) in the two lanes of each component
    V2 re, im;
    static CLM_HD Cx2 add(Cx2 a, Cx2 b) { return Cx2{a.re + b.re, a.im + b.im}; }
    static CLM_HD Cx2 sub(Cx2 a, Cx2 b) { return Cx2{a.re - b.re, a.im - b.im}; }
    static CLM_HD Cx2 mul(Cx2 a, Cx2 b) { return Cx2{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
    static CLM_HD Cx2 mulc(Cx2 a, float cr, float ci) {
        const V2 r = splat2(cr), i = splat2(ci);
        return Cx2{a.re * r - a.im * i, a.re * i + a.im * r};
    }
    static CLM_HD Cx2 conj(Cx2 a) { return Cx2{a.re, -a.im}; }
    static CLM_HD Cx2 neg(Cx2 a) { return Cx2{-a.re, -a.im}; }
    template <bool INV>
    static CLM_HD Cx2 mul_mi(Cx2 a) { return INV ? Cx2{-a.im, a.re} : Cx2{a.im, -a.re}; }
};
CLM_HD Cx2 pack2(float2 a, float2 b) { return Cx2{make_v2(a.x, b.x), make_v2(a.y, b.y)}; }
CLM_HD float2 lane_a(Cx2 c) { return make_float2(c.re.x, c.im.x); }
CLM_HD float2 lane_b(Cx2 c) { return make_float2(c.re.y, c.im.y); }

// LDS layout: structure of arrays -- re[] and im[] as separate float arrays (the two packed reads ARE the real and
// imaginary parts), index padded by 4 dwords per 32 elements:
//   * unit-stride ds_read_b32/read2 over 32 lanes: conflict-free; 8-float rows of phase A/C: conflict-free b128;
//   * stride-R stores of the first (Ns = 1) pass: 2-way instead of R-way;
//   * the two butterflies of a pair sit NT elements apart = NT*9/8 dwords, a multiple of 64 for NT >= 512, so one
//     ds_read2st64_b32 / ds_write2st64_b32 moves both lanes of a Cx2 component with no register shuffling.
CLM_HDC int pad_index(int i) { return i + 4 * (i >> 5); }
CLM_HDC int padded_size(int n) { return n + 4 * (n >> 5); }
// pad_index(a + c) == pad_index(a) + pad_offset(c) whenever (a mod 32) + (c mod 32) < 32.  Every access of the
// Stockham passes has that form with c a compile-time constant (r*N/R, q*Ns, NT), so LDS addresses are one
// per-thread base plus an immediate instead of a shift/add chain per access (35 % of the VALU stream before).
CLM_HDC int pad_offset(int c) { return c + 4 * (c >> 5); }

// ---- small DFTs, in place, natural order in and out ------------------------------------------------------
template <bool INV, class C>
CLM_HD void dft2(C& a0, C& a1) {
    C t = a0;
    a0 = C::add(t, a1);
    a1 = C::sub(t, a1);
}

template <bool INV, class C>
CLM_HD void dft4(C& a0, C& a1, C& a2, C& a3) {
    C t0 = C::add(a0, a2), t1 = C::sub(a0, a2), t2 = C::add(a1, a3), t3 = C::template mul_mi<INV>(C::sub(a1, a3));
    a0 = C::add(t0, t2);
    a2 = C::sub(t0, t2);
    a1 = C::add(t1, t3);
    a3 = C::sub(t1, t3);
}

// dft4 whose inputs a2 and a3 are zero (their incoming values are ignored): 4 complex additions instead of 8
template <bool INV, class C>
CLM_HD void dft4_lo(C& a0, C& a1, C& a2, C& a3) {
    C t3 = C::template mul_mi<INV>(a1), s = C::add(a0, a1), d = C::sub(a0, a1);
    a3 = C::sub(a0, t3);
    a1 = C::add(a0, t3);
    a0 = s;
    a2 = d;
}

// v * exp(-+ 2*pi*i * m / 16), m = 0..9 (m is a compile-time constant after unrolling)
template <bool INV, class C>
CLM_HD C mul_w16(C v, int m) {
    const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, h = 0.70710678118654752440f;
    float cx, sy;
    switch (m) {
        case 0: return v;
        case 1: cx = c1; sy = s1; break;
        case 2: cx = h; sy = h; break;
        case 3: cx = s1; sy = c1; break;
        case 4: return C::template mul_mi<INV>(v);
        case 5: cx = -s1; sy = c1; break;
        case 6: cx = -h; sy = h; break;
        case 7: cx = -c1; sy = s1; break;
        case 8: return C::neg(v);
        default: cx = -c1; sy = -s1; break;  // m = 9
    }
    return C::mulc(v, cx, INV ? sy : -sy);
}

// Generic radix-R DFT on v[0..R-1] (R in {2,4,8,16}); all indices are compile-time after unrolling.
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    template <class C>
    static CLM_HD void run(C* v) { dft2<INV>(v[0], v[1]); }
};
template <bool INV>
struct Dft<4, INV> {
    template <class C>
    static CLM_HD void run(C* v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
};
// R = R1*R2 with n = n1 + R1*n2, k = R2*k1 + k2:
//   Y[n1][k2] = DFT_R2 over n2;  Y *= W_R^(n1*k2);  X[R2*k1+k2] = DFT_R1 over n1.
template <bool INV>
struct Dft<8, INV> {  // R1 = 2, R2 = 4
    template <class C>
    static CLM_HD void run(C* v) {
        dft4<INV>(v[0], v[2], v[4], v[6]);  // n1 = 0: position 0 + 2*k2
        dft4<INV>(v[1], v[3], v[5], v[7]);  // n1 = 1: position 1 + 2*k2
        v[3] = mul_w16<INV>(v[3], 2);       // W8^(1*1)
        v[5] = mul_w16<INV>(v[5], 4);       // W8^(1*2) = -+i
        v[7] = mul_w16<INV>(v[7], 6);       // W8^(1*3)
        dft2<INV>(v[0], v[1]);              // k2 = 0 -> X[0], X[4]
        dft2<INV>(v[2], v[3]);              // k2 = 1 -> X[1], X[5]
        dft2<INV>(v[4], v[5]);              // k2 = 2 -> X[2], X[6]
        dft2<INV>(v[6], v[7]);              // k2 = 3 -> X[3], X[7]
        // position p = k1 + 2*k2 holds X[4*k1 + k2]  ->  natural order
        C x1 = v[2], x2 = v[4], x3 = v[6], x4 = v[1], x5 = v[3], x6 = v[5];
        v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
    }
};
template <bool INV>
struct Dft<16, INV> {  // R1 = R2 = 4
    template <class C>
    static CLM_HD void run(C* v) {
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) dft4<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);  // pos n1 + 4*k2
#pragma unroll
        for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = mul_w16<INV>(v[n1 + 4 * k2], n1 * k2);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) dft4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
        // position p = k1 + 4*k2 holds X[4*k1 + k2]: transpose the 4x4 register grid (pure renaming)
        C t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) t[p] = v[p];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v[4 * k1 + k2] = t[k1 + 4 * k2];
    }
};

// DFT16 of a vector whose upper half v[8..15] is zero (not read): the first pass over a zero-padded convolution input
template <bool INV>
struct Dft16Lo {
    template <class C>
    static CLM_HD void run(C* v) {
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) dft4_lo<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
#pragma unroll
        for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = mul_w16<INV>(v[n1 + 4 * k2], n1 * k2);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) dft4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
        C t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) t[p] = v[p];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v[4 * k1 + k2] = t[k1 + 4 * k2];
    }
};

// DFT16 of which only the outputs X[0..8] are wanted (the last inverse pass of a convolution whose upper output half is
// discarded; X[8] is the element at N/2 of 8193-token reads): the second stage computes 2 (3 for k2 = 0) of its 4 outputs.
// v[9..15] are left undefined.
template <bool INV>
struct Dft16HalfOut {
    template <class C>
    static CLM_HD void run(C* v) {
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) dft4<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
#pragma unroll
        for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = mul_w16<INV>(v[n1 + 4 * k2], n1 * k2);
        C x[9];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {   // X[4*k1 + k2], k1 = 0, 1 (and k1 = 2 for k2 = 0)
            const C a0 = v[4 * k2], a1 = v[4 * k2 + 1], a2 = v[4 * k2 + 2], a3 = v[4 * k2 + 3];
            const C t0 = C::add(a0, a2), t1 = C::sub(a0, a2), t2 = C::add(a1, a3), t3 = C::template mul_mi<INV>(C::sub(a1, a3));
            x[k2] = C::add(t0, t2);
            x[4 + k2] = C::add(t1, t3);
            if (k2 == 0) x[8] = C::sub(t0, t2);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] = x[q];
    }
};

// v[r] *= w^r for r = 1..R-1 with a multiplication tree of depth <= 4 (error ~ 4 ulp instead of R ulp)
template <int R, class C>
CLM_HD void apply_twiddle_powers(C* v, C w1) {
    if (R >= 2) v[1] = C::mul(v[1], w1);
    if (R >= 4) {
        C w2 = C::mul(w1, w1), w3 = C::mul(w2, w1);
        v[2] = C::mul(v[2], w2);
        v[3] = C::mul(v[3], w3);
        if (R >= 8) {
            C w4 = C::mul(w2, w2), w5 = C::mul(w4, w1), w6 = C::mul(w4, w2), w7 = C::mul(w4, w3);
            v[4] = C::mul(v[4], w4);
            v[5] = C::mul(v[5], w5);
            v[6] = C::mul(v[6], w6);
            v[7] = C::mul(v[7], w7);
            if (R >= 16) {
                C w8 = C::mul(w4, w4);
                v[8] = C::mul(v[8], w8);
                v[9] = C::mul(v[9], C::mul(w8, w1));
                v[10] = C::mul(v[10], C::mul(w8, w2));
                v[11] = C::mul(v[11], C::mul(w8, w3));
                v[12] = C::mul(v[12], C::mul(w8, w4));
                v[13] = C::mul(v[13], C::mul(w8, w5));
                v[14] = C::mul(v[14], C::mul(w8, w6));
                v[15] = C::mul(v[15], C::mul(w8, w7));
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

// twiddle of butterfly jb in a pass with sub-transform size Ns and radix R: exp(-+2*pi*i*(jb mod Ns)/(Ns*R));
// `tw` is the table exp(-2*pi*i*m/N), m < N/2.
template <int LOGN, int R, bool INV>
CLM_HD float2 twiddle_for(int jb, int Ns, const float2* tw) {
    constexpr int N = 1 << LOGN;
    if (Ns <= 1) return make_float2(1.f, 0.f);
    float2 w1 = tw[(jb & (Ns - 1)) * (N / (Ns * R))];
    return INV ? cconj(w1) : w1;
}

// one butterfly pair: optional twiddle by w1^r, then DFT_R
template <int R, bool INV, class C>
CLM_HD void butterfly_w(C* v, bool has_tw, C w1) {
    if (has_tw) apply_twiddle_powers<R>(v, w1);
    Dft<R, INV>::run(v);
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
