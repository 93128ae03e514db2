// fft_passes.h -- per-thread phases of one Stockham pass over an N-point complex array held in LDS.
// Shared by the device kernel (hyena_conv.hip) and the host emulation (fft_core_test.cpp): the host test
// runs the phases for every tid in turn with the barriers replaced by loop boundaries.
//
// A thread owns 32 complex values per pass: IT = (N/R)/NT butterflies of radix R (NT = N/32 threads, at least one
// wave), processed as NP = ceil(IT/2) PAIRS in the two lanes of a Cx2 (fft_core.h).  Pair p holds butterflies
// jb_a = tid + 2p*NT and jb_b = jb_a + NT; a missing partner (odd IT, tiny transforms) computes on zeros.
// Pass structure (in place, single buffer):
//     load  : v[p*R + r].re = (re[pad(jb_a + r*N/R)], re[pad(jb_b + r*N/R)]), same for im   (unit stride across lanes)
//     compute: twiddle + DFT_R in registers (no barrier needed between load and compute)
//     barrier (all loads of the pass done)
//     store : re/im[pad((jb-k)*R + k + q*Ns)] = v[p*R + q] lane a / b
//     barrier
#pragma once
#include "fft_core.h"

namespace clmfft {

template <int LOGN, int R>
struct PassGeom {
    static constexpr int N = 1 << LOGN;
    static constexpr int NB = N / R;  // butterflies in the pass
    static constexpr int NT = Plan<LOGN>::NT;
    static constexpr int IT = (NB + NT - 1) / NT;
    static constexpr int NP = (IT + 1) / 2;
    static_assert(NP * R <= 16, "a thread holds at most 16 complex pairs");
    static CLM_HD int jba(int tid, int p) { return tid + (2 * p) * NT; }
    static CLM_HD int jbb(int tid, int p) { return tid + (2 * p + 1) * NT; }
    static constexpr bool FULL = (NB % NT == 0);   // every thread owns exactly IT butterflies: no bounds checks at all
    static CLM_HD bool has_a(int tid, int p) { return FULL || jba(tid, p) < NB; }
    static CLM_HD bool has_b(int tid, int p) { return 2 * p + 1 < IT && (FULL || jbb(tid, p) < NB); }
    // padded LDS index of lane b's butterfly, given lane a's: jbb = jba + NT and NT is a multiple of the padding period, so the
    // distance is a CONSTANT.  Written as `pa + constant` (not pad_index(jbb), a second address register) the two lanes of a
    // component are one ds_read2st64_b32 straight into the register pair the packed arithmetic wants; with two base registers
    // hipcc paired the loads across r instead (same lane, inputs r and r + 1) and rebuilt every pair with v_mov -- 470 moves
    // per unit, 8 % of the convolution kernel's issue cycles.
    static_assert(NT % 32 == 0, "lane distance is a whole number of padding periods");
    static CLM_HDC int pb_of(int pa) { return pa + pad_offset(NT); }
};

template <int LOGN, int R>
CLM_HD void pass_load(const float* re, const float* im, Cx2* v, int tid) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        const bool ha = G::has_a(tid, p), hb = G::has_b(tid, p);
        // one base per component (lane a's butterfly); input r and lane b are immediates: r*NB and NT elements further on
        const float* ra = re + pad_index(G::jba(tid, p));
        const float* ia = im + pad_index(G::jba(tid, p));
        constexpr int dB = G::pb_of(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int off = pad_offset(r * G::NB);
            v[p * R + r].re = make_v2(ha ? ra[off] : 0.f, hb ? ra[off + dB] : 0.f);
            v[p * R + r].im = make_v2(ha ? ia[off] : 0.f, hb ? ia[off + dB] : 0.f);
        }
    }
}

// First forward pass of a zero-padded convolution input (elements N/2 .. N-1 are zero and are neither stored nor read):
// only inputs r < 8 of each radix-16 butterfly are loaded, and the butterfly skips the additions with zero.  A single
// non-zero element at index N/2 (reads of exactly N/2 + 1 tokens) is added back in the frequency domain, where it is
// (-1)^m times its value (spectrum_multiply_and_first_inverse_v).
template <int LOGN>
CLM_HD void pass_first_lower(const float* re, const float* im, Cx2* v, int tid) {
    using G = PassGeom<LOGN, 16>;
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        const bool ha = G::has_a(tid, p), hb = G::has_b(tid, p);
        const float* ra = re + pad_index(G::jba(tid, p));
        const float* ia = im + pad_index(G::jba(tid, p));
        constexpr int dB = G::pb_of(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int off = pad_offset(r * G::NB);
            v[p * 16 + r].re = make_v2(ha ? ra[off] : 0.f, hb ? ra[off + dB] : 0.f);
            v[p * 16 + r].im = make_v2(ha ? ia[off] : 0.f, hb ? ia[off + dB] : 0.f);
        }
        Dft16Lo<false>::run(v + p * 16);
    }
}

template <int LOGN, int R>
CLM_HD void pass_store(float* re, float* im, const Cx2* v, int tid, int Ns) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        const bool ha = G::has_a(tid, p), hb = G::has_b(tid, p);
        const int pa = pad_index(stockham_out<R>(G::jba(tid, p), 0, Ns));            // bases; q*Ns is the immediate
        // partner butterfly jb + NT: a constant distance away in the two common cases, so both lanes of a component
        // can leave in one ds_write2st64_b32
        const int pb = (Ns <= G::NT && G::NT % Ns == 0) ? pa + pad_offset(G::NT * R)
                       : (Ns >= G::NB)                 ? pa + pad_offset(G::NT)
                                                       : pad_index(stockham_out<R>(G::jbb(tid, p), 0, Ns));
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int off = pad_offset(q * Ns);
            if (ha) re[pa + off] = v[p * R + q].re.x, im[pa + off] = v[p * R + q].im.x;
            if (hb) re[pb + off] = v[p * R + q].re.y, im[pb + off] = v[p * R + q].im.y;
        }
    }
}

// Last inverse pass of a convolution whose outputs N/2+1 .. N-1 are discarded: twiddle, the half-output butterfly, and a store
// of outputs q < 8 only (Ns = N/16: output index jb + q*N/16); `keep_mid`: also output N/2 (butterfly 0, q = 8).
template <int LOGN>
CLM_HD void pass_compute_last_inverse_lower(Cx2* v, int tid, const Cx2* w) {
    using G = PassGeom<LOGN, 16>;
    static_assert(G::FULL, "every thread owns whole butterfly pairs");
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        apply_twiddle_powers<16>(v + p * 16, w[p]);
        Dft16HalfOut<true>::run(v + p * 16);
    }
}
template <int LOGN>
CLM_HD void pass_store_lower(float* re, float* im, const Cx2* v, int tid, bool keep_mid) {
    using G = PassGeom<LOGN, 16>;
    constexpr int Ns = G::NB;                                   // last pass: sub-transform size N/16 = number of butterflies
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        const int pa = pad_index(G::jba(tid, p)), pb = G::pb_of(pa);                   // k = jb, output jb + q*Ns
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int off = pad_offset(q * Ns);
            re[pa + off] = v[p * 16 + q].re.x, im[pa + off] = v[p * 16 + q].im.x;
            re[pb + off] = v[p * 16 + q].re.y, im[pb + off] = v[p * 16 + q].im.y;
        }
        if (keep_mid && G::jba(tid, p) == 0) re[pad_index(8 * Ns)] = v[p * 16 + 8].re.x, im[pad_index(8 * Ns)] = v[p * 16 + 8].im.x;
    }
}

// Twiddle prefetch: w[p] for the butterfly pairs of one pass (issued at kernel start, consumed passes later).
template <int LOGN, int R, bool INV>
CLM_HD void pass_twiddles(Cx2* w, int tid, int Ns, const float2* tw) {
    using G = PassGeom<LOGN, R>;
    const float2 one = make_float2(1.f, 0.f);
#pragma unroll
    for (int p = 0; p < G::NP; ++p)
        w[p] = pack2(G::has_a(tid, p) ? twiddle_for<LOGN, R, INV>(G::jba(tid, p), Ns, tw) : one,
                     G::has_b(tid, p) ? twiddle_for<LOGN, R, INV>(G::jbb(tid, p), Ns, tw) : one);
}

template <int LOGN, int R, bool INV>
CLM_HD void pass_compute_w(Cx2* v, int tid, bool has_tw, const Cx2* w) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int p = 0; p < G::NP; ++p) butterfly_w<R, INV>(v + p * R, has_tw, w[p]);
}

// filter spectrum bins of the last forward pass, fetched ahead of the pass: kv[p*R + q] = kf[jb + q*N/R]
template <int LOGN, int R>
CLM_HD void spectrum_fetch(Cx2* kv, int tid, const float2* kf) {
    using G = PassGeom<LOGN, R>;
    const float2 zero = make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
        const bool ha = G::has_a(tid, p), hb = G::has_b(tid, p);
#pragma unroll
        for (int q = 0; q < R; ++q)
            kv[p * R + q] = pack2(ha ? kf[stockham_in<LOGN, R>(G::jba(tid, p), q)] : zero,
                                  hb ? kf[stockham_in<LOGN, R>(G::jbb(tid, p), q)] : zero);
    }
}

// Fused middle of the convolution: after the LAST forward pass a thread holds, for each of its butterflies jb, the
// spectrum bins m = jb + q*N/R (q = 0..R-1) -- exactly the inputs of the FIRST inverse pass (Ns = 1) of the same
// radix.  Multiply by the filter spectrum and run that inverse butterfly (no twiddles at Ns = 1).
// `tail_re`, `tail_im`: the element at index N/2 of the packed input (zero unless L == N/2 + 1), left out of the pruned first
// pass; its transform is (-1)^m, and every bin m = jb + q*N/R of a thread has the parity of tid (N/R and NT are even).
template <int LOGN, int R>
CLM_HD void spectrum_multiply_and_first_inverse_v(Cx2* v, int tid, const Cx2* kv, float tail_re = 0.f, float tail_im = 0.f) {
    using G = PassGeom<LOGN, R>;
    static_assert(((1 << LOGN) / R) % 2 == 0 && G::NT % 2 == 0, "bin parity == thread parity");
    const float sg = (tid & 1) ? -1.f : 1.f;
    const Cx2 tl{splat2(sg * tail_re), splat2(sg * tail_im)};
#pragma unroll
    for (int p = 0; p < G::NP; ++p) {
#pragma unroll
        for (int q = 0; q < R; ++q) v[p * R + q] = Cx2::mul(Cx2::add(v[p * R + q], tl), kv[p * R + q]);
        Dft<R, true>::run(v + p * R);
    }
}

// Twiddle register layout of a whole convolution (forward passes 1..NPASS-1, inverse passes 1..NPASS-1), in pairs:
template <int LOGN>
struct TwLayout {
    using P = Plan<LOGN>;
    static constexpr int NP16 = PassGeom<LOGN, 16>::NP;
    static constexpr int NPL = PassGeom<LOGN, P::LAST>::NP;
    static constexpr int fwd(int p) { return (p - 1) * NP16; }              // R = 16 passes p = 1 .. NPASS-2
    static constexpr int fwd_last() { return (P::NPASS - 2) * NP16; }
    static constexpr int inv(int p) { return (P::NPASS - 2) * NP16 + NPL + (p - 1) * NP16; }   // p = 1 .. NPASS-1
    static constexpr int TOTAL = (2 * P::NPASS - 3) * NP16 + NPL;
};

}  // namespace clmfft
