// fft_passes.h -- per-thread phases of one Stockham pass over an N-point complex array held in LDS.
// Shared by the device kernel (hyena_conv.hip) and the host emulation (fft_core_test.cpp): the host test
// runs the phases for every tid in turn with the barriers replaced by loop boundaries.
//
// A thread owns 16 complex values per pass: IT = (N/R)/NT butterflies of radix R (NT = N/16 threads, at
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
    static_assert(IT * R <= 16, "a thread holds at most 16 complex values");
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
