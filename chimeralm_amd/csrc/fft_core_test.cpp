// Host emulation of the LDS Stockham convolution used by hyena_conv.hip (built with g++ by
// tests/test_fft_core.py).  Every "thread" of the kernel is run in turn, barriers become loop boundaries,
// and the result is checked against a direct double-precision causal convolution, including the
// single-alias correction used when L == N/2 + 1 (HISTORY.md section 4.6, "long convolution").
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fft_passes.h"

using namespace clmfft;
using cd = std::complex<double>;

static void fft_ref(std::vector<cd>& a, bool inv) {  // plain recursive radix-2, double
    size_t n = a.size();
    if (n == 1) return;
    std::vector<cd> e(n / 2), o(n / 2);
    for (size_t i = 0; i < n / 2; ++i) e[i] = a[2 * i], o[i] = a[2 * i + 1];
    fft_ref(e, inv);
    fft_ref(o, inv);
    for (size_t k = 0; k < n / 2; ++k) {
        cd w = std::polar(1.0, (inv ? 2.0 : -2.0) * M_PI * double(k) / double(n)) * o[k];
        a[k] = e[k] + w;
        a[k + n / 2] = e[k] - w;
    }
}

template <int LOGN>
static double test_conv(int L, unsigned seed) {
    using P = Plan<LOGN>;
    constexpr int N = P::N, NT = P::NT, LAST = P::LAST;
    if (!(2 * L - 2 <= N && L >= 1)) {
        std::printf("bad L\n");
        std::exit(2);
    }
    std::mt19937 rng(seed);
    std::normal_distribution<double> nd;
    std::vector<double> g0(L), g1(L), k(L);
    for (int t = 0; t < L; ++t) g0[t] = nd(rng), g1[t] = nd(rng), k[t] = nd(rng) * std::exp(-3.0 * t / L);
    // filter spectrum / N  (what the engine precomputes in double, stored as float2)
    std::vector<cd> kf(N);
    for (int t = 0; t < L; ++t) kf[t] = k[t];
    fft_ref(kf, false);
    std::vector<float2> kff(N), tw(N / 2);
    for (int m = 0; m < N; ++m) kff[m] = make_float2(float(kf[m].real() / N), float(kf[m].imag() / N));
    for (int m = 0; m < N / 2; ++m) tw[m] = make_float2(float(std::cos(2 * M_PI * m / N)), float(-std::sin(2 * M_PI * m / N)));
    // signal in padded "LDS"
    std::vector<float> bre(padded_size(N), 0.f), bim(padded_size(N), 0.f);
    // like the kernel: tokens < N/2 in the lower half; the upper half is never written (the pruned first pass does not read
    // it); the single element at N/2 (only when L == N/2 + 1) is carried to the spectrum product
    for (int i = 0; i < padded_size(N); ++i) bre[i] = bim[i] = 1e30f;     // poison: anything read by mistake shows
    for (int t = 0; t < N / 2; ++t) bre[pad_index(t)] = t < L ? float(g0[t]) : 0.f, bim[pad_index(t)] = t < L ? float(g1[t]) : 0.f;
    const float tail_re = L == N / 2 + 1 ? float(g0[N / 2]) : 0.f, tail_im = L == N / 2 + 1 ? float(g1[N / 2]) : 0.f;
    // all twiddles are fetched up front, exactly like the kernel does
    using TL = TwLayout<LOGN>;
    std::vector<Cx2> wall(size_t(NT) * TL::TOTAL);
    for (int tid = 0; tid < NT; ++tid) {
        Cx2* w = &wall[size_t(tid) * TL::TOTAL];
        int ns = 16;
        for (int p = 1; p <= P::NPASS - 2; ++p, ns *= 16) pass_twiddles<LOGN, 16, false>(w + TL::fwd(p), tid, ns, tw.data());
        pass_twiddles<LOGN, LAST, false>(w + TL::fwd_last(), tid, ns, tw.data());
        ns = LAST;
        for (int p = 1; p <= P::NPASS - 1; ++p, ns *= 16) pass_twiddles<LOGN, 16, true>(w + TL::inv(p), tid, ns, tw.data());
    }
    std::vector<Cx2> regs(size_t(NT) * 16), kv(size_t(NT) * 16);
    auto R = [&](int tid) { return &regs[size_t(tid) * 16]; };
    auto W = [&](int tid) { return &wall[size_t(tid) * TL::TOTAL]; };
    // forward passes
    int Ns = 1;
    for (int p = 0; p < P::NPASS - 1; ++p) {
        if (p == 0) {
            for (int tid = 0; tid < NT; ++tid) pass_first_lower<LOGN>(bre.data(), bim.data(), R(tid), tid);
        } else {
            for (int tid = 0; tid < NT; ++tid) pass_load<LOGN, 16>(bre.data(), bim.data(), R(tid), tid);
            for (int tid = 0; tid < NT; ++tid) pass_compute_w<LOGN, 16, false>(R(tid), tid, true, W(tid) + TL::fwd(p));
        }
        for (int tid = 0; tid < NT; ++tid) pass_store<LOGN, 16>(bre.data(), bim.data(), R(tid), tid, Ns);
        Ns *= 16;
    }
    for (int tid = 0; tid < NT; ++tid) spectrum_fetch<LOGN, LAST>(&kv[size_t(tid) * 16], tid, kff.data());
    for (int tid = 0; tid < NT; ++tid) pass_load<LOGN, LAST>(bre.data(), bim.data(), R(tid), tid);
    for (int tid = 0; tid < NT; ++tid) pass_compute_w<LOGN, LAST, false>(R(tid), tid, true, W(tid) + TL::fwd_last());
    for (int tid = 0; tid < NT; ++tid) spectrum_multiply_and_first_inverse_v<LOGN, LAST>(R(tid), tid, &kv[size_t(tid) * 16], tail_re, tail_im);
    for (int tid = 0; tid < NT; ++tid) pass_store<LOGN, LAST>(bre.data(), bim.data(), R(tid), tid, 1);
    Ns = LAST;
    for (int p = 1; p <= P::NPASS - 1; ++p) {
        for (int tid = 0; tid < NT; ++tid) pass_load<LOGN, 16>(bre.data(), bim.data(), R(tid), tid);
        if constexpr (PassGeom<LOGN, 16>::FULL) {
            if (p == P::NPASS - 1) {   // like the kernel: only the lower output half (+ element N/2) of the last pass
                for (int tid = 0; tid < NT; ++tid) pass_compute_last_inverse_lower<LOGN>(R(tid), tid, W(tid) + TL::inv(p));
                for (int i = N / 2 + 1; i < N; ++i) bre[pad_index(i)] = bim[pad_index(i)] = 1e30f;   // never written, never used
                for (int tid = 0; tid < NT; ++tid) pass_store_lower<LOGN>(bre.data(), bim.data(), R(tid), tid, L == N / 2 + 1);
                continue;
            }
        }
        for (int tid = 0; tid < NT; ++tid) pass_compute_w<LOGN, 16, true>(R(tid), tid, true, W(tid) + TL::inv(p));
        for (int tid = 0; tid < NT; ++tid) pass_store<LOGN, 16>(bre.data(), bim.data(), R(tid), tid, Ns);
        Ns *= 16;
    }
    // reference: direct causal convolution in double
    double max_err = 0, max_ref = 0;
    for (int t = 0; t < L; ++t) {
        double r0 = 0, r1 = 0;
        for (int s = 0; s <= t; ++s) r0 += k[s] * g0[t - s], r1 += k[s] * g1[t - s];
        double y0 = bre[pad_index(t)], y1 = bim[pad_index(t)];
        if (t == 0 && 2 * L - 2 == N) {  // the one aliased term k[L-1]*g[L-1] wraps onto output 0
            y0 -= double(float(k[L - 1])) * double(float(g0[L - 1]));
            y1 -= double(float(k[L - 1])) * double(float(g1[L - 1]));
        }
        max_err = std::fmax(max_err, std::fmax(std::fabs(y0 - r0), std::fabs(y1 - r1)));
        max_ref = std::fmax(max_ref, std::fmax(std::fabs(r0), std::fabs(r1)));
    }
    return max_err / max_ref;
}

template <int LOGN>
static int run(unsigned seed) {
    constexpr int N = 1 << LOGN;
    int fails = 0;
    for (int L : {N / 2 + 1, N / 2, N / 2 - 3, N / 4 + 7, 5}) {
        if (2 * L - 2 > N || L < 1) continue;
        double e = test_conv<LOGN>(L, seed + L);
        std::printf("LOGN=%d L=%d rel_err=%.3e\n", LOGN, L, e);
        if (!(e < 2e-5)) ++fails;
    }
    return fails;
}

int main() {
    int fails = 0;
    fails += run<8>(1);
    fails += run<9>(2);
    fails += run<10>(3);
    fails += run<11>(4);
    fails += run<12>(5);
    fails += run<13>(6);
    fails += run<14>(7);
    std::printf(fails ? "FAIL %d\n" : "ALL OK\n", fails);
    return fails ? 1 : 0;
}
