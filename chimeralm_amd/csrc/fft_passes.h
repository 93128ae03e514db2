// fft_passes.h -- per-thread phases of one Stockham pass over an N-point complex array held in LDS.
// Shared by the device kernel (hyena_conv.hip) and the host emulation (fft_core_test.cpp): the host test
// runs the phases for every tid in turn with the barriers replaced by loop boundaries.
//
// A thread owns 32 complex values per pass: IT = (N/R)/NT butterflies of radix R (NT = N/32 threads, at
// least one wave).  Pass structure (in place, single buffer):
//     load  : v[it*R + r] = buf[pad(jb + r*N/R)],  jb = tid + it*NT          (unit stride across lanes)
//     -- no barrier needed between load and compute --
//     compute: twiddle + DFT_R in registers
//     barrier (all loads of the pass done)
//     store : buf[pad((jb-k)*R + k + q*Ns)] = v[it*R + q]
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
    static_assert(IT * R <= 32, "a thread holds at most 32 complex values");
};

template <int LOGN, int R>
CLM_HD void pass_load(const float2* buf, float2* v, int tid) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[it * R + r] = buf[pad_index(stockham_in<LOGN, R>(jb, r))];
        }
    }
}

template <int LOGN, int R, bool INV>
CLM_HD void pass_compute(float2* v, int tid, int Ns, const float2* tw) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) butterfly<LOGN, R, INV>(v + it * R, jb, Ns, tw);
    }
}

template <int LOGN, int R>
CLM_HD void pass_store(float2* buf, const float2* v, int tid, int Ns) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[pad_index(stockham_out<R>(jb, q, Ns))] = v[it * R + q];
        }
    }
}

// Twiddle prefetch: w[it] for the butterflies of one pass (issued at kernel start, consumed passes later).
template <int LOGN, int R, bool INV>
CLM_HD void pass_twiddles(float2* w, int tid, int Ns, const float2* tw) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        w[it] = (jb < G::NB) ? twiddle_for<LOGN, R, INV>(jb, Ns, tw) : make_float2(1.f, 0.f);
    }
}
template <int LOGN, int R, bool INV>
CLM_HD void pass_compute_w(float2* v, int tid, bool has_tw, const float2* w) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) butterfly_w<R, INV>(v + it * R, has_tw, w[it]);
    }
}
// filter spectrum bins of the last forward pass, fetched ahead of the pass: kv[it*R + q] = kf[jb + q*N/R]
template <int LOGN, int R>
CLM_HD void spectrum_fetch(float2* kv, int tid, const float2* kf) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) {
#pragma unroll
            for (int q = 0; q < R; ++q) kv[it * R + q] = kf[stockham_in<LOGN, R>(jb, q)];
        }
    }
}
template <int LOGN, int R>
CLM_HD void spectrum_multiply_and_first_inverse_v(float2* v, int tid, const float2* kv) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) {
#pragma unroll
            for (int q = 0; q < R; ++q) v[it * R + q] = cmul(v[it * R + q], kv[it * R + q]);
            Dft<R, true>::run(v + it * R);
        }
    }
}

// Twiddle register layout of a whole convolution (forward passes 1..NPASS-1, inverse passes 1..NPASS-1):
template <int LOGN>
struct TwLayout {
    using P = Plan<LOGN>;
    static constexpr int IT16 = PassGeom<LOGN, 16>::IT;
    static constexpr int ITL = PassGeom<LOGN, P::LAST>::IT;
    static constexpr int fwd(int p) { return (p - 1) * IT16; }              // R = 16 passes p = 1 .. NPASS-2
    static constexpr int fwd_last() { return (P::NPASS - 2) * IT16; }
    static constexpr int inv(int p) { return (P::NPASS - 2) * IT16 + ITL + (p - 1) * IT16; }   // p = 1 .. NPASS-1
    static constexpr int TOTAL = (2 * P::NPASS - 3) * IT16 + ITL;
};

// Fused middle of the convolution: after the LAST forward pass thread `tid` holds, for each of its
// butterflies jb, the spectrum bins m = jb + q*N/R (q = 0..R-1) -- exactly the inputs of the FIRST inverse
// pass (Ns = 1) of the same radix.  Multiply by the filter spectrum and run that inverse butterfly.
template <int LOGN, int R>
CLM_HD void spectrum_multiply_and_first_inverse(float2* v, int tid, const float2* kf) {
    using G = PassGeom<LOGN, R>;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
        int jb = tid + it * G::NT;
        if (jb < G::NB) {
#pragma unroll
            for (int q = 0; q < R; ++q) v[it * R + q] = cmul(v[it * R + q], kf[stockham_in<LOGN, R>(jb, q)]);
            butterfly<LOGN, R, true>(v + it * R, jb, 1, nullptr);
        }
    }
}

}  // namespace clmfft
