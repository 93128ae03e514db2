// hyena_conv.hip -- the Hyena mixer's non-GEMM half on gfx950:
//   implicit filter MLP, filter spectrum, short (3-tap) depthwise filter, gates and the causal long convolution.
//
// Reference arithmetic (HyenaDNA remote code):
//   uc = Conv1d(768,768,3,padding=2,groups=768)(z)[..., :L];  x0, x1, v = split(uc, 256)
//   g  = v * x1
//   yc = irfft(rfft(g, 2L) * rfft(k, 2L) / 2L, n=2L, norm="forward")[..., :L] + g * D        (fftconv)
//   y  = yc * x0
// with k[t, c] = MLP_sin(z_pos[t]) * (exp(-t_norm |delta_c|) + 0.05).
//
// Engine formulation: a causal linear convolution is the same operator for any transform size that avoids
// wrap-around, so the FFT-hostile 2L (2L = 16386 for 8 kbp reads) is replaced by a power-of-two COMPLEX FFT of
// N >= 2L-2 points held entirely in LDS, with TWO reads of the same channel packed as real and imaginary part
// (the filter is real, so it acts on both parts independently: no untangling pass, and the filter spectrum
// is fetched once per pair).  N == 2L-2 (the 8193-token case) aliases exactly one product, k[L-1]*g[L-1],
// onto output 0; it is subtracted explicitly.  One workgroup = one (channel, read pair).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "clm_common.h"
#include "fft_passes.h"

namespace clm {
using namespace clmfft;

// ---------------------------------------------------------------------------------------- implicit filter
// One workgroup of 64 threads per position t; double precision inside (evaluated once per distinct L and
// cached by the handle, so its cost is irrelevant and its rounding error is below the reference's own).
__global__ __launch_bounds__(64) void filter_kernel(const float* __restrict__ z, const float* __restrict__ tn,
                                                    const float* __restrict__ w0, const float* __restrict__ b0,
                                                    const float* __restrict__ freq, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const float* __restrict__ w4,
                                                    const float* __restrict__ b4, const float* __restrict__ w6,
                                                    const float* __restrict__ deltas, float* __restrict__ k_out,
                                                    int L) {
    __shared__ double hbuf[2][FORDER];
    const int t = blockIdx.x, j = threadIdx.x;
    const double f = (double)freq[j];
    double acc = (double)b0[j];
#pragma unroll
    for (int e = 0; e < EMB; ++e) acc += (double)w0[j * EMB + e] * (double)z[(size_t)t * EMB + e];
    hbuf[0][j] = sin(f * acc);
    __syncthreads();
    acc = (double)b2[j];
    for (int i = 0; i < FORDER; ++i) acc += (double)w2[j * FORDER + i] * hbuf[0][i];
    hbuf[1][j] = sin(f * acc);
    __syncthreads();
    acc = (double)b4[j];
    for (int i = 0; i < FORDER; ++i) acc += (double)w4[j * FORDER + i] * hbuf[1][i];
    __syncthreads();
    hbuf[0][j] = sin(f * acc);
    __syncthreads();
    const double tt = (double)tn[t];
    for (int c = j; c < D; c += 64) {
        double o = 0.0;
        for (int i = 0; i < FORDER; ++i) o += (double)w6[c * FORDER + i] * hbuf[0][i];
        double decay = exp(-tt * fabs((double)deltas[c]));
        k_out[(size_t)t * D + c] = (float)(o * (decay + 0.05));
    }
}

void launch_filter(const float* z, const float* t, const float* w0, const float* b0, const float* freq,
                   const float* w2, const float* b2, const float* w4, const float* b4, const float* w6,
                   const float* deltas, float* k_out, int L, hipStream_t st) {
    hipLaunchKernelGGL(filter_kernel, dim3(L), dim3(64), 0, st, z, t, w0, b0, freq, w2, b2, w4, b4, w6, deltas,
                       k_out, L);
}

// krev[c][t] = k[L-1-t][c] (+ the skip term D[c] on tap 0, i.e. at t = L-1), channel-major with row stride `stride`:
// the last output of a read, y[L-1] = sum_t g[t] k[L-1-t], as a dot product over ascending t (long reads with a lone tail).
__global__ __launch_bounds__(256) void krev_kernel(const float* __restrict__ k, const float* __restrict__ dskip,
                                                   float* __restrict__ krev, int L, int stride) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= stride) return;
    float v = 0.f;
    if (t < L) {
        v = k[(size_t)(L - 1 - t) * D + c];
        if (t == L - 1) v += dskip[c];
    }
    krev[(size_t)c * stride + t] = v;
}
void launch_filter_reversed(const float* k, const float* dskip, float* krev, int L, int stride, hipStream_t st) {
    hipLaunchKernelGGL(krev_kernel, dim3((stride + 255) / 256, D), dim3(256), 0, st, k, dskip, krev, L, stride);
}

// ---------------------------------------------------------------------------------------- filter spectrum
// kf[c][m] = (1/N) sum_t k[t][c] exp(-2 pi i m t / N), double precision radix-2 in global scratch, one
// workgroup per channel (runs once per distinct L).
// The skip term of fftconv, y += g * D[c], is a convolution with D[c]*delta: it is folded into tap 0 here, so the
// convolution kernel needs neither D nor g after the transform.
// Segment form (long reads): taps [seg_off, seg_off + seg_len) of the filter in the lower half and, when prev_off >= 0, taps
// [prev_off, prev_off + seg_len) in the upper half (overlap-save partition, see hyena_conv_seg_kernel); zero elsewhere.
__global__ __launch_bounds__(256) void spectrum_kernel(const float* __restrict__ k, const float* __restrict__ dskip,
                                                       float2* __restrict__ kf, double2* __restrict__ scratch, int L,
                                                       int logn, int seg_off, int seg_len, int prev_off) {
    const int N = 1 << logn, c = blockIdx.x;
    double2* a = scratch + (size_t)c * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {  // bit-reversed load
        int rv = __brev((unsigned)i) >> (32 - logn);
        const bool upper = prev_off >= 0 && i >= seg_len && i < 2 * seg_len;
        const int t = upper ? prev_off + (i - seg_len) : seg_off + i;
        double v = ((i < seg_len || upper) && t < L) ? (double)k[(size_t)t * D + c] : 0.0;
        if (t == 0 && (i < seg_len || upper)) v += (double)dskip[c];
        a[rv] = make_double2(v, 0.0);
    }
    __syncthreads();
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
            int grp = i >> (s - 1), pos = i & (half - 1);
            int i0 = (grp << s) + pos, i1 = i0 + half;
            double sn, cs;
            sincospi(-(double)pos / (double)half, &sn, &cs);  // exp(-2 pi i pos / 2^s)
            double2 u = a[i0], v = a[i1];
            double2 w = make_double2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
            a[i0] = make_double2(u.x + w.x, u.y + w.y);
            a[i1] = make_double2(u.x - w.x, u.y - w.y);
        }
        __syncthreads();
    }
    const double inv = 1.0 / (double)N;
    for (int i = threadIdx.x; i < N; i += blockDim.x)
        kf[(size_t)c * N + i] = make_float2((float)(a[i].x * inv), (float)(a[i].y * inv));
}

void launch_filter_spectrum(const float* k, const float* dskip, float2* kf, double2* scratch, int L, int logn,
                            int seg_off, int seg_len, int prev_off, hipStream_t st) {
    hipLaunchKernelGGL(spectrum_kernel, dim3(D), dim3(256), 0, st, k, dskip, kf, scratch, L, logn, seg_off, seg_len,
                       prev_off);
}

__global__ void twiddle_kernel(float2* tw, int logn) {
    int m = blockIdx.x * blockDim.x + threadIdx.x, N = 1 << logn;
    if (m < N / 2) {
        double sn, cs;
        sincospi(-2.0 * (double)m / (double)N, &sn, &cs);
        tw[m] = make_float2((float)cs, (float)sn);
    }
}
void launch_twiddles(float2* tw, int logn, hipStream_t st) {
    int n = (1 << logn) / 2;
    hipLaunchKernelGGL(twiddle_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tw, logn);
}

// ---------------------------------------------------------------------------------------- the convolution
int conv_logn_for(int L) {
    if (L < 1) return -1;
    for (int logn = 8; logn <= 14; ++logn)
        if (2 * L - 2 <= (1 << logn)) return logn;
    return -1;  // L > 8193: partitioned path (conv_segments_for)
}

// Long reads: S segments of SEG_LEN tokens, each through the 16384-point transform (hyena_conv_seg_kernel).
// A read of S*SEG_LEN + 1 tokens (every read truncated at a multiple of 8192 bases + [SEP]: 16385, 24577, 32769 = the model
// maximum) does not get a segment of its own for the lone last token: see conv_lone_tail.
int conv_segments_for(int L) { return L <= SEG_LEN + 1 ? 1 : (L - 1 + SEG_LEN - 1) / SEG_LEN; }
bool conv_lone_tail(int L) { return L > SEG_LEN + 1 && (L - 1) % SEG_LEN == 0; }

// 8 consecutive activations (one 16-byte vector for the 16-bit types, two for fp32) as floats
template <typename T>
__device__ __forceinline__ void load8(const T* p, float* o);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* o) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* o) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void load8<f16_t>(const f16_t* p, float* o) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    _Float16 h[8];
    __builtin_memcpy(h, &a, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
}
// 8 x 16-bit -> fp32
template <typename T>
__device__ __forceinline__ void cvt8(const uint4 d, float* o) {
    const unsigned w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T lo, hi;
        lo.bits = (unsigned short)(w[i] & 0xffffu);
        hi.bits = (unsigned short)(w[i] >> 16);
        o[2 * i] = to_float(lo);
        o[2 * i + 1] = to_float(hi);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float* v);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) {
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = from_float<bf16_t>(v[i]).bits;
    *reinterpret_cast<u16x8*>(p) = o;
}
template <>
__device__ __forceinline__ void store8<f16_t>(f16_t* p, const float* v) {
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = from_float<f16_t>(v[i]).bits;
    *reinterpret_cast<u16x8*>(p) = o;
}

// ---- round 4 (fp16c): lo bytes next to the 16-bit rows of z and y (clm_common.h lo8_pack4) --------------------------------------
// z: the gated in_proj stage leaves the lo bytes of x0f | g in rows 512.. of the read's z block, as [2][256][Lp] bytes.
template <typename T>
__device__ __forceinline__ const unsigned char* zlo_row(const T* zb /* z + b * 768 * Lp */, int which /*0: x0f, 1: g*/, int c, int Lp) {
    return reinterpret_cast<const unsigned char*>(zb + (size_t)2 * D * Lp) + (size_t)(which * D + c) * Lp;
}
// o[0..7] += the corrections of 8 lo bytes
__device__ __forceinline__ void add_lo8(const uint2 lo, float* o) {
    float d[4];
    lo8_unpack4(lo.x, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += d[e];
    lo8_unpack4(lo.y, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[4 + e] += d[e];
}
__device__ __forceinline__ float lo1b(unsigned byte) { return __builtin_amdgcn_cvt_f32_bf8((int)byte, 0) * LO2_INV; }
__device__ __forceinline__ float lo1(const unsigned char* row, int t) { return lo1b(row[t]); }
// 8 values -> 8 halfs at p and their 8 lo bytes at lo
__device__ __forceinline__ void store8_hilo(f16_t* p, unsigned char* lo, const float* v) {
    u16x4 h0, h1;
    const unsigned l0 = lo8_pack4(v[0], v[1], v[2], v[3], h0), l1 = lo8_pack4(v[4], v[5], v[6], v[7], h1);
    *reinterpret_cast<u16x8*>(p) = u16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    *reinterpret_cast<uint2*>(lo) = make_uint2(l0, l1);
}
__device__ __forceinline__ void store1_hilo(f16_t* p, unsigned char* lo, float v) {
    u16x4 h;
    const unsigned l = lo8_pack4(v, 0.f, 0.f, 0.f, h);
    p->bits = h[0];
    *lo = (unsigned char)(l & 0xffu);
}
// the kernels' y stores: plain, or hi + lo when LO (fp16c; T = f16_t then)
template <typename T, bool LO>
__device__ __forceinline__ void ystore8(T* p, unsigned char* lo, const float* v) {
    if constexpr (LO) store8_hilo(p, lo, v);
    else store8<T>(p, v);
}
template <typename T, bool LO>
__device__ __forceinline__ void ystore1(T* p, unsigned char* lo, float v) {
    if constexpr (LO) store1_hilo(p, lo, v);
    else *p = from_float<T>(v);
}

template <typename T>
__device__ __forceinline__ void load2(const T* p, float* o);
template <>
__device__ __forceinline__ void load2<float>(const float* p, float* o) {
    float2 a = *reinterpret_cast<const float2*>(p);
    o[0] = a.x; o[1] = a.y;
}
template <>
__device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float* o) {
    unsigned w = *reinterpret_cast<const unsigned*>(p);
    o[0] = __uint_as_float(w << 16);
    o[1] = __uint_as_float(w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void load2<f16_t>(const f16_t* p, float* o) {
    unsigned w = *reinterpret_cast<const unsigned*>(p);
    _Float16 h[2];
    __builtin_memcpy(h, &w, 4);
    o[0] = (float)h[0]; o[1] = (float)h[1];
}

// short filter on 8 consecutive samples of one channel row: out[e] = b + w0*x[t-2] + w1*x[t-1] + w2*x[t]
template <typename T>
__device__ __forceinline__ void short_filter8(const T* row, int t0, float w0, float w1, float w2, float bias,
                                              float* out) {
    float x[10];
    load8<T>(row + t0, x + 2);
    if (t0 > 0) {
        load2<T>(row + t0 - 2, x);   // t0 is a multiple of 8: the pair is naturally aligned
    } else {
        x[0] = 0.f;
        x[1] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = bias + w0 * x[e] + w1 * x[e + 1] + w2 * x[e + 2];
}
template <typename T>
__device__ __forceinline__ float short_filter1(const T* row, int t, float w0, float w1, float w2, float bias) {
    float xm2 = t >= 2 ? to_float(row[t - 2]) : 0.f, xm1 = t >= 1 ? to_float(row[t - 1]) : 0.f;
    return bias + w0 * xm2 + w1 * xm1 + w2 * to_float(row[t]);
}

// Raw samples of one 8-token chunk of one row (+ the two preceding samples), loaded with no control flow in between so
// that ALL of a thread's loads are in flight together: stamps showed the branchy load->wait->filter sequence costing
// six serialized HBM round trips (23k of the kernel's 63k cycles).
template <typename T>
struct Raw;
template <>
struct Raw<float> {
    float4 a, b;
    float2 p;
};
template <>
struct Raw<bf16_t> {
    uint4 d;
    unsigned p;
};
template <>
struct Raw<f16_t> {
    uint4 d;
    unsigned p;
};
__device__ __forceinline__ void raw_load(Raw<float>& r, const float* row, int t0, bool valid) {
    const float* q = row + (valid ? t0 : 0);
    r.a = *reinterpret_cast<const float4*>(q);
    r.b = *reinterpret_cast<const float4*>(q + 4);
    r.p = *reinterpret_cast<const float2*>(row + ((valid && t0 > 0) ? t0 - 2 : 0));
}
template <typename T>
__device__ __forceinline__ void raw_load(Raw<T>& r, const T* row, int t0, bool valid) {
    r.d = *reinterpret_cast<const uint4*>(row + (valid ? t0 : 0));                       // clamped, never out of the row
    r.p = *reinterpret_cast<const unsigned*>(row + ((valid && t0 > 0) ? t0 - 2 : 0));
}
// x[0..1] = samples t0-2, t0-1 (zero at the start of the read), x[2..9] = samples t0 .. t0+7; all zero if !valid
__device__ __forceinline__ void raw_decode(const Raw<float>& r, int t0, bool valid, float* x) {
    const float v = valid ? 1.f : 0.f, vp = (valid && t0 > 0) ? 1.f : 0.f;
    x[0] = vp * r.p.x; x[1] = vp * r.p.y;
    x[2] = v * r.a.x; x[3] = v * r.a.y; x[4] = v * r.a.z; x[5] = v * r.a.w;
    x[6] = v * r.b.x; x[7] = v * r.b.y; x[8] = v * r.b.z; x[9] = v * r.b.w;
}
template <typename T>
__device__ __forceinline__ void raw_decode(const Raw<T>& r, int t0, bool valid, float* x) {
    const unsigned w[5] = {(valid && t0 > 0) ? r.p : 0u, valid ? r.d.x : 0u, valid ? r.d.y : 0u, valid ? r.d.z : 0u,
                           valid ? r.d.w : 0u};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        T lo, hi;
        lo.bits = (unsigned short)(w[i] & 0xffffu);
        hi.bits = (unsigned short)(w[i] >> 16);
        x[2 * i] = to_float(lo);
        x[2 * i + 1] = to_float(hi);
    }
}
// IDS source: x[a3][0..9] = table row a3 of the ten token ids t0-2 .. t0+7 (zero outside the read)
// SELECT: load every entry and select afterwards (ids are always inside the 16-row table).  The segmented kernel needs that
// form: there hipcc turns the conditional loads into one divergent branch per element (1,600 spilled registers).
// ROWS: bit a3 set = row a3 wanted (the others are left untouched)
template <bool SELECT = false, int ROWS = 7>
__device__ __forceinline__ void ids_decode(uint2 d, unsigned short p, int t0, bool valid, const float* zt, float (*x)[10]) {
    const unsigned w[3] = {p, d.x, d.y};
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int sh = j < 2 ? 8 * j : 8 * ((j - 2) & 3);
        const unsigned id = (w[j < 2 ? 0 : 1 + ((j - 2) >> 2)] >> sh) & 15u;
        const bool ok = j < 2 ? (valid && t0 > 0) : valid;
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3) {
            if (!((ROWS >> a3) & 1)) continue;
            if constexpr (SELECT) {
                const float val = zt[a3 * 16 + id];
                x[a3][j] = ok ? val : 0.f;
            } else {
                x[a3][j] = ok ? zt[a3 * 16 + id] : 0.f;
            }
        }
    }
}
__device__ __forceinline__ void fir3(const float* x /*[10]*/, float w0, float w1, float w2, float bias, float* out /*[8]*/) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = bias + w0 * x[e] + w1 * x[e + 1] + w2 * x[e + 2];
}

// the same filter on two rows at once (read A and read B of the pair share the channel's taps): packed fp32, one v_pk_fma per
// two multiply-adds
__device__ __forceinline__ void fir3_pair(const float* xa, const float* xb, float w0, float w1, float w2, float bias, float* oa,
                                          float* ob) {
    using clmfft::V2;
    const V2 W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2}, Bv = {bias, bias};
    V2 x[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) x[e] = V2{xa[e], xb[e]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const V2 r = Bv + W0 * x[e] + W1 * x[e + 1] + W2 * x[e + 2];
        oa[e] = r.x;
        ob[e] = r.y;
    }
}

// 8 consecutive floats to / from an LDS array whose padding never splits an aligned group of 8
__device__ __forceinline__ void lds_store8(float* dst, const float* v) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void lds_load8(const float* src, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// One workgroup = one (channel, pair of reads); NT = N/32 threads, each owning 32 complex points per pass.
//   phase 0  issue every long-latency read that does not depend on data: all pass twiddles (registers)
//   phase A  z -> short filter -> gate -> LDS (natural order, two reads packed as re/im); x0 stays in registers
//   phase B  forward passes; last forward pass fused with the spectrum product and the first inverse pass;
//            inverse passes (the filter spectrum bins are fetched one pass ahead)
//   phase C  LDS -> * x0 -> y
// STAMP: developer build (CLM_DEBUG=stamp, 16384-point f16 instantiation only) recording s_memtime at phase boundaries.
constexpr int CONV_NSTAMP = 16;
// IDS: first block only.  The residual stream entering block 0 is the embedding row of the token id, so the block's
// in_proj output is one of 16 precomputed rows (ztab, fp32): the kernel reads the ids (1 byte per token, shared by all 256
// channel workgroups of a read) and looks x0 / x1 / v up instead of reading z -- in_proj of block 0 is never launched.
// GATED: z holds x0f (row c) and g = x1f * vf (row 256 + c), filtered and gated by the producer (see hyena_conv_pers_kernel).
// LO (fp16c, T = f16_t): y leaves as hi + lo bytes (ylo [B][256][Lp]); GATED: x0f and g are read as hi + lo (zlo_row).
template <int LOGN, typename T, bool STAMP = false, bool IDS = false, bool GATED = false, bool LO = false>
__global__ __launch_bounds__(Plan<LOGN>::NT) void hyena_conv_kernel(
    const T* __restrict__ z, T* __restrict__ y, const float2* __restrict__ kf, const float2* __restrict__ tw,
    const float* __restrict__ ktime, const float* __restrict__ short_w, const float* __restrict__ short_b, int B, int L,
    int Lp, unsigned long long* stamps, const unsigned char* __restrict__ ids8, const float* __restrict__ ztab,
    unsigned char* __restrict__ ylo) {
    static_assert(!LO || std::is_same<T, f16_t>::value, "lo bytes exist in the compensated fp16 mode only");
#define CLM_STAMP_AT(k)                                                                                   \
    do {                                                                                                  \
        if (STAMP && threadIdx.x == 0)                                                                    \
            stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * CONV_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    using P = Plan<LOGN>;
    using TL = TwLayout<LOGN>;
    constexpr int N = P::N, NT = P::NT, LAST = P::LAST, HALF = N / 2;
    constexpr int CH = (HALF / 8 + NT - 1) / NT;        // 8-token chunks of the lower half per thread (2, or 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bre = reinterpret_cast<float*>(smem);             // read A of the pair = real parts
    float* bim = bre + padded_size(N);                       // read B            = imaginary parts

    const int tid = threadIdx.x;
    const int c = blockIdx.y, pair = blockIdx.x;
    const int bA = 2 * pair, bB = 2 * pair + 1;
    const bool hasB = bB < B;
    const T* zA = z + (size_t)bA * D3 * Lp;
    const T* zB = z + (size_t)(hasB ? bB : bA) * D3 * Lp;
    const float2* kfc = kf + (size_t)c * N;

    // ---------------------------------------------------------------- phase 0: twiddles of every pass
    CLM_STAMP_AT(0);
    if (STAMP && threadIdx.x == 0)    // 100-MHz counter at both ends of the unit: the shader clock the kernel runs at
        stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * CONV_NSTAMP + 15] = __builtin_amdgcn_s_memrealtime();
    Cx2 wall[TL::TOTAL];
    {
        int ns = 16;
#pragma unroll
        for (int p = 1; p <= P::NPASS - 2; ++p) {
            pass_twiddles<LOGN, 16, false>(wall + TL::fwd(p), tid, ns, tw);
            ns *= 16;
        }
        pass_twiddles<LOGN, LAST, false>(wall + TL::fwd_last(), tid, ns, tw);
        ns = LAST;
#pragma unroll
        for (int p = 1; p <= P::NPASS - 1; ++p) {
            pass_twiddles<LOGN, 16, true>(wall + TL::inv(p), tid, ns, tw);
            ns *= 16;
        }
    }

    // per-channel constants: short filter taps of channels c (x0), 256+c (x1), 512+c (v)
    float sw[3][3], sb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int e = 0; e < 3; ++e) sw[q][e] = short_w[(q * D + c) * 3 + e];
        sb[q] = short_b[q * D + c];
    }

    CLM_STAMP_AT(12);
    // ---------------------------------------------------------------- phase A: load, short filter, gate
    // every global load of the phase is issued before the first use (see Raw<T>)
    constexpr int TAIL_TID = HALF / 8 - 1 - (CH - 1) * NT;   // owner of tokens [HALF-8, HALF): also computes token HALF
    const bool tail = (L == HALF + 1);
    static_assert(!(GATED && (IDS || std::is_same<T, float>::value)), "the gated hand-over exists in the fused 16-bit path only");
    float x0A[CH][8], x0B[CH][8];
    float x0At = 0.f, gAt = 0.f, x0Bt = 0.f, gBt = 0.f;
    if constexpr (GATED) {
        uint4 gr[CH][2][2];                                  // [chunk][read][x0f, g]
        uint2 grl[LO ? CH : 1][2][2];                        // LO: their lo bytes
        T gtl[2][2];
        float gtll[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                const T* row = (rd == 0 ? zA : zB) + (size_t)(a2 * D + c) * Lp;
                const unsigned char* rowl = zlo_row(rd == 0 ? zA : zB, a2, c, Lp);
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (tid + ch * NT), tc = (t0 < HALF && t0 < L) ? t0 : 0;
                    gr[ch][rd][a2] = *reinterpret_cast<const uint4*>(row + tc);
                    if constexpr (LO) grl[ch][rd][a2] = *reinterpret_cast<const uint2*>(rowl + tc);
                }
                gtl[rd][a2] = row[(tail && tid == TAIL_TID) ? HALF : 0];
                if constexpr (LO) gtll[rd][a2] = lo1(rowl, (tail && tid == TAIL_TID) ? HALF : 0);
            }
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const int t0 = 8 * (tid + ch * NT);
            if (t0 < HALF) {
                float gA[8], gB[8];
                cvt8<T>(gr[ch][0][0], x0A[ch]);
                cvt8<T>(gr[ch][1][0], x0B[ch]);
                cvt8<T>(gr[ch][0][1], gA);
                cvt8<T>(gr[ch][1][1], gB);
                if constexpr (LO) {
                    add_lo8(grl[ch][0][0], x0A[ch]);
                    add_lo8(grl[ch][1][0], x0B[ch]);
                    add_lo8(grl[ch][0][1], gA);
                    add_lo8(grl[ch][1][1], gB);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gA[e] = (t0 + e < L) ? gA[e] : 0.f;
                    gB[e] = (hasB && t0 + e < L) ? gB[e] : 0.f;
                }
                lds_store8(bre + pad_index(t0), gA);
                lds_store8(bim + pad_index(t0), gB);
            }
        }
        if (tail && tid == TAIL_TID) {
            x0At = to_float(gtl[0][0]) + gtll[0][0];
            gAt = to_float(gtl[0][1]) + gtll[0][1];
            x0Bt = to_float(gtl[1][0]) + gtll[1][0];
            gBt = hasB ? to_float(gtl[1][1]) + gtll[1][1] : 0.f;
        }
    } else {
    Raw<T> raw[IDS ? 1 : CH][2][IDS ? 1 : 3];
    T ztail[2][3];
    uint2 idd[CH][2];                                        // IDS: 8 token ids of the chunk
    unsigned short idp[CH][2];                               //      and the two before it
    unsigned char idt[2] = {0, 0};
    float* zt = bim + padded_size(N) + 4;                    // IDS: [3][16] rows x0 | x1 | v of this channel
    if constexpr (IDS) {
        if (tid < 48) zt[tid] = ztab[(size_t)(tid & 15) * D3 + (tid >> 4) * D + c];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const unsigned char* ir = ids8 + (size_t)(rd == 0 ? bA : (hasB ? bB : bA)) * Lp;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int t0 = 8 * (tid + ch * NT);
                const bool valid = t0 < HALF && t0 < L;
                idd[ch][rd] = *reinterpret_cast<const uint2*>(ir + (valid ? t0 : 0));
                idp[ch][rd] = *reinterpret_cast<const unsigned short*>(ir + ((valid && t0 > 0) ? t0 - 2 : 0));
            }
            idt[rd] = ir[(tail && tid == TAIL_TID) ? HALF : 0];
        }
        __syncthreads();                                     // zt visible
    } else {
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const T* zr = rd == 0 ? zA : zB;
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3) {
                const T* row = zr + (size_t)(a3 * D + c) * Lp;
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (tid + ch * NT);
                    raw_load(raw[ch][rd][a3], row, t0, t0 < HALF && t0 < L);
                }
                ztail[rd][a3] = row[(tail && tid == TAIL_TID) ? HALF : 0];
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int t0 = 8 * (tid + ch * NT);
        if (t0 < HALF) {
            const bool valid = t0 < L;
            float xa[3][10], xb[3][10], gA[8], gB[8], x1[8], v[8];
            if constexpr (IDS) {
                ids_decode(idd[ch][0], idp[ch][0], t0, valid, zt, xa);
                ids_decode(idd[ch][1], idp[ch][1], t0, valid && hasB, zt, xb);
            } else {
#pragma unroll
                for (int a3 = 0; a3 < 3; ++a3) {
                    raw_decode(raw[ch][0][a3], t0, valid, xa[a3]);
                    raw_decode(raw[ch][1][a3], t0, valid && hasB, xb[a3]);
                }
            }
            float x1b[8], vb[8];
            fir3_pair(xa[0], xb[0], sw[0][0], sw[0][1], sw[0][2], sb[0], x0A[ch], x0B[ch]);
            fir3_pair(xa[1], xb[1], sw[1][0], sw[1][1], sw[1][2], sb[1], x1, x1b);
            fir3_pair(xa[2], xb[2], sw[2][0], sw[2][1], sw[2][2], sb[2], v, vb);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gA[e] = (t0 + e < L) ? v[e] * x1[e] : 0.f;
                gB[e] = (hasB && t0 + e < L) ? vb[e] * x1b[e] : 0.f;
            }
            lds_store8(bre + pad_index(t0), gA);
            lds_store8(bim + pad_index(t0), gB);
            if (ch == CH - 1 && tail && tid == TAIL_TID) {   // token HALF: taps are x[8], x[9] of this chunk and z[HALF]
                float ta[3], tb[3];
#pragma unroll
                for (int a3 = 0; a3 < 3; ++a3) {
                    const float za = IDS ? zt[a3 * 16 + (idt[0] & 15)] : to_float(ztail[0][a3]);
                    const float zb = IDS ? zt[a3 * 16 + (idt[1] & 15)] : to_float(ztail[1][a3]);
                    ta[a3] = sb[a3] + sw[a3][0] * xa[a3][8] + sw[a3][1] * xa[a3][9] + sw[a3][2] * za;
                    tb[a3] = sb[a3] + sw[a3][0] * xb[a3][8] + sw[a3][1] * xb[a3][9] + sw[a3][2] * zb;
                }
                x0At = ta[0];
                gAt = ta[1] * ta[2];
                x0Bt = tb[0];
                gBt = hasB ? tb[1] * tb[2] : 0.f;
            }
        }
    }
    }
    CLM_STAMP_AT(13);
    // (the upper half of the transform input is zero padding: it is neither written nor read -- pass_first_lower)
    CLM_STAMP_AT(14);
    float* gtail = bim + padded_size(N);                     // g[HALF] of both reads, for the alias correction by thread 0
    CLM_STAMP_AT(1);
    __syncthreads();
    if (tail && tid == TAIL_TID) {      // token N/2 exists only when L == N/2 + 1 (8193 tokens in a 16384-point transform):
        gtail[0] = gAt;                 // it joins in the frequency domain (spectrum product) and is needed again for the
        gtail[1] = gBt;                 // alias correction of output 0
    }
    __syncthreads();
    CLM_STAMP_AT(2);

    // ---------------------------------------------------------------- phase B: FFT, spectrum product, inverse FFT
    Cx2 v[16];   // 32 points per thread as 16 SoA pairs
    {
        int Ns = 1;
#pragma unroll
        for (int p = 0; p < P::NPASS - 1; ++p) {
            if (p == 0) {
                pass_first_lower<LOGN>(bre, bim, v, tid);     // zero upper half: half the loads, a shorter butterfly
            } else {
                pass_load<LOGN, 16>(bre, bim, v, tid);
                pass_compute_w<LOGN, 16, false>(v, tid, true, wall + TL::fwd(p));
            }
            __syncthreads();
            pass_store<LOGN, 16>(bre, bim, v, tid, Ns);
            __syncthreads();
            CLM_STAMP_AT(3 + p);
            Ns *= 16;
        }
    }
    {
        Cx2 kv[16];
        spectrum_fetch<LOGN, LAST>(kv, tid, kfc);          // L2 latency overlaps the LDS loads and the butterfly
        pass_load<LOGN, LAST>(bre, bim, v, tid);
        pass_compute_w<LOGN, LAST, false>(v, tid, true, wall + TL::fwd_last());
        spectrum_multiply_and_first_inverse_v<LOGN, LAST>(v, tid, kv, tail ? gtail[0] : 0.f, tail ? gtail[1] : 0.f);
    }
    __syncthreads();
    pass_store<LOGN, LAST>(bre, bim, v, tid, 1);
    __syncthreads();
    CLM_STAMP_AT(7);
    {
        int Ns = LAST;
#pragma unroll
        for (int p = 1; p <= P::NPASS - 1; ++p) {
            pass_load<LOGN, 16>(bre, bim, v, tid);
            bool lower = false;
            if constexpr (PassGeom<LOGN, 16>::FULL) {
                if (p == P::NPASS - 1) {   // outputs beyond N/2 are discarded: half-output butterfly, half the stores
                    lower = true;
                    pass_compute_last_inverse_lower<LOGN>(v, tid, wall + TL::inv(p));
                    __syncthreads();
                    pass_store_lower<LOGN>(bre, bim, v, tid, tail);
                }
            }
            if (!lower) {
                pass_compute_w<LOGN, 16, true>(v, tid, true, wall + TL::inv(p));
                __syncthreads();
                pass_store<LOGN, 16>(bre, bim, v, tid, Ns);
            }
            __syncthreads();
            CLM_STAMP_AT(7 + p);
            Ns *= 16;
        }
    }

    // ---------------------------------------------------------------- phase C: gate with x0, store
    T* yA = y + ((size_t)bA * D + c) * Lp;
    T* yB = y + ((size_t)(hasB ? bB : bA) * D + c) * Lp;
    unsigned char* ylA = LO ? ylo + ((size_t)bA * D + c) * Lp : nullptr;
    unsigned char* ylB = LO ? ylo + ((size_t)(hasB ? bB : bA) * D + c) * Lp : nullptr;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int t0 = 8 * (tid + ch * NT);
        if (t0 < HALF && t0 < Lp) {
            float oA[8], oB[8];
            lds_load8(bre + pad_index(t0), oA);
            lds_load8(bim + pad_index(t0), oB);
            if (tail && t0 == 0) {  // remove the one wrapped product k[L-1]*g[L-1] from output 0
                const float kl = ktime[(size_t)(L - 1) * D + c];
                oA[0] -= kl * gtail[0];
                oB[0] -= kl * gtail[1];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = t0 + e < L;
                oA[e] = ok ? oA[e] * x0A[ch][e] : 0.f;
                oB[e] = ok ? oB[e] * x0B[ch][e] : 0.f;
            }
            ystore8<T, LO>(yA + t0, ylA + t0, oA);
            if (hasB) ystore8<T, LO>(yB + t0, ylB + t0, oB);
        }
    }
    if (tail && tid == TAIL_TID) {
        ystore1<T, LO>(yA + HALF, ylA + HALF, bre[pad_index(HALF)] * x0At);
        if (hasB) ystore1<T, LO>(yB + HALF, ylB + HALF, bim[pad_index(HALF)] * x0Bt);
    }
    CLM_STAMP_AT(11);
    if (STAMP && threadIdx.x == 0)
        stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * CONV_NSTAMP + 6] = __builtin_amdgcn_s_memrealtime();
#undef CLM_STAMP_AT
}


// ================================================================================================ 8k reads: persistent units
// hyena_conv_kernel has one workgroup per CU (LDS) and starts every unit with an HBM round trip for its z rows that nothing
// overlaps (stamps: phase A 19 % of a unit).  Round 1 tried a persistent loop that prefetched the next unit's six rows into
// registers: 40-66 spilled registers and slower.  This variant changes WHAT is held across the transform instead:
//   * x0 (needed only by the output gate) is no longer computed in phase A and carried through the passes (32 registers):
//     its two rows are requested just before the last inverse pass and filtered in phase C;
//   * the x1 / v rows of the NEXT unit of this workgroup are requested at the same point (40 registers, live only across the
//     last pass, phase C and the next phase A -- where the transform's 64 data registers are dead);
//   * the pass twiddles are loaded once per workgroup instead of once per unit.
// Units are taken round-robin by gridDim.x = #CUs workgroups; LOGN = 14 (reads of 4098 .. 8193 tokens) only.
// GATED (the fused tail kernel's hand-over, gemm16.hip inproj_blocks_gated): z row c holds x0f, row 256 + c holds g = x1f * vf --
// already filtered and gated by the producer -- so phase A is a load + convert of ONE row and phase C a multiply with one row.
template <typename T, bool IDS, bool GATED = false, bool LO = false>
struct ConvGateRaw {                                   // x1 / v rows of one unit, both reads (IDS: the token ids instead)
    Raw<T> r[GATED ? 1 : 2][2][GATED ? 1 : 2];         // [chunk][read][x1, v]
    T ztail[2][2];
    uint2 idd[2][2];
    unsigned short idp[2][2];
    unsigned char idt[2];
    uint4 g[2][2];                                     // GATED: 8 samples of g per [chunk][read]
    T gtail[2];
    uint2 gl[2][2];                                    // GATED + LO: their lo bytes
    unsigned char gltail[2];
};

// LO (fp16c): as hyena_conv_kernel -- y leaves as hi + lo bytes, the gated rows are read as hi + lo.
template <typename T, bool IDS, bool GATED = false, bool LO = false>
__global__ __launch_bounds__(Plan<14>::NT) void hyena_conv_pers_kernel(
    const T* __restrict__ z, T* __restrict__ y, const float2* __restrict__ kf /*lane-packed*/, const float2* __restrict__ tw,
    const float* __restrict__ ktime, const float* __restrict__ short_w, const float* __restrict__ short_b, int B, int L,
    int Lp, const unsigned char* __restrict__ ids8, const float* __restrict__ ztab, int use_xcd, unsigned char* __restrict__ ylo) {
    static_assert(!LO || std::is_same<T, f16_t>::value, "lo bytes exist in the compensated fp16 mode only");
    constexpr int LOGN = 14;
    using P = Plan<LOGN>;
    using TL = TwLayout<LOGN>;
    constexpr int N = P::N, NT = P::NT, LAST = P::LAST, HALF = N / 2, CH = 2;
    static_assert(HALF / 8 / NT == CH, "two 8-token chunks per thread");
    constexpr int TAIL_TID = HALF / 8 - 1 - (CH - 1) * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bre = reinterpret_cast<float*>(smem);
    float* bim = bre + padded_size(N);
    float* gtail = bim + padded_size(N);
    float* zt = gtail + 4;

    const int tid = threadIdx.x;
    const int pairs = (B + 1) / 2, n_units = pairs * D;
    const bool tail = (L == HALF + 1);
    // Unit order.  Workgroups go round-robin to the 8 XCDs (id % 8), each with its own L2; a channel's spectrum (128 KB) is
    // shared by all read pairs of the channel.  XCD x therefore takes the channels c = x (mod 8), and its workgroups walk
    // (channel, pair) channel-major: a spectrum is fetched from HBM by one XCD, once, instead of by all eight.
    const bool xcd_order = use_xcd && (gridDim.x % XCDS) == 0;
    const int xcd = blockIdx.x % XCDS, wg_in_xcd = blockIdx.x / XCDS, wgs_per_xcd = gridDim.x / XCDS;
    const int first = xcd_order ? wg_in_xcd : (int)blockIdx.x, stride = xcd_order ? wgs_per_xcd : (int)gridDim.x;
    const int n_mine = xcd_order ? n_units / XCDS : n_units;            // D is a multiple of XCDS
    auto unit_of = [&](int q, int& c, int& pair) {
        c = xcd_order ? XCDS * (q / pairs) + xcd : q / pairs;
        pair = q % pairs;
    };

    Cx2 wall[TL::TOTAL];                                // twiddles of every pass: unit-independent
    {
        int ns = 16;
#pragma unroll
        for (int p = 1; p <= P::NPASS - 2; ++p) {
            pass_twiddles<LOGN, 16, false>(wall + TL::fwd(p), tid, ns, tw);
            ns *= 16;
        }
        pass_twiddles<LOGN, LAST, false>(wall + TL::fwd_last(), tid, ns, tw);
        ns = LAST;
#pragma unroll
        for (int p = 1; p <= P::NPASS - 1; ++p) {
            pass_twiddles<LOGN, 16, true>(wall + TL::inv(p), tid, ns, tw);
            ns *= 16;
        }
    }
    static_assert(!(GATED && IDS), "block 0 looks its rows up by token id: nothing to hand over");
    // requests of the gate rows (x1, v) of unit u: no control flow between the loads (all in flight together)
    auto request_gate = [&](int q, ConvGateRaw<T, IDS, GATED, LO>& g, int ltid) {
        int c, pair;
        unit_of(q, c, pair);
        const int bA = 2 * pair, bB = (2 * pair + 1 < B) ? 2 * pair + 1 : bA;
        if constexpr (GATED) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const T* row = z + ((size_t)(rd == 0 ? bA : bB) * D3 + D + c) * Lp;
                const unsigned char* rowl = zlo_row(z + (size_t)(rd == 0 ? bA : bB) * D3 * Lp, 1, c, Lp);
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (ltid + ch * NT);
                    g.g[ch][rd] = *reinterpret_cast<const uint4*>(row + (t0 < L ? t0 : 0));      // clamped, never out of the row
                    if constexpr (LO) g.gl[ch][rd] = *reinterpret_cast<const uint2*>(rowl + (t0 < L ? t0 : 0));
                }
                g.gtail[rd] = row[(tail && ltid == TAIL_TID) ? HALF : 0];
                if constexpr (LO) g.gltail[rd] = rowl[(tail && ltid == TAIL_TID) ? HALF : 0];
            }
        } else if constexpr (IDS) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const unsigned char* ir = ids8 + (size_t)(rd == 0 ? bA : bB) * Lp;
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (ltid + ch * NT);
                    const bool valid = t0 < L;
                    g.idd[ch][rd] = *reinterpret_cast<const uint2*>(ir + (valid ? t0 : 0));
                    g.idp[ch][rd] = *reinterpret_cast<const unsigned short*>(ir + ((valid && t0 > 0) ? t0 - 2 : 0));
                }
                g.idt[rd] = ir[(tail && ltid == TAIL_TID) ? HALF : 0];
            }
        } else {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const T* zr = z + (size_t)(rd == 0 ? bA : bB) * D3 * Lp;
#pragma unroll
                for (int a3 = 1; a3 < 3; ++a3) {
                    const T* row = zr + (size_t)(a3 * D + c) * Lp;
#pragma unroll
                    for (int ch = 0; ch < CH; ++ch) {
                        const int t0 = 8 * (ltid + ch * NT);
                        raw_load(g.r[ch][rd][a3 - 1], row, t0, t0 < L);
                    }
                    g.ztail[rd][a3 - 1] = row[(tail && ltid == TAIL_TID) ? HALF : 0];
                }
            }
        }
    };

    ConvGateRaw<T, IDS, GATED, LO> cur;
    request_gate(first < n_mine ? first : 0, cur, tid);

#pragma unroll 1
    for (int u = first; u < n_mine; u += stride) {
        int ltid = tid;                                  // opaque per trip: keeps the pass addresses from being hoisted and held
        asm volatile("" : "+v"(ltid));
#pragma unroll
        for (int i = 0; i < TL::TOTAL; ++i)
            asm volatile("" : "+v"(wall[i].re.x), "+v"(wall[i].re.y), "+v"(wall[i].im.x), "+v"(wall[i].im.y));
        int c, pair;
        unit_of(u, c, pair);
        const int bA = 2 * pair, bB = 2 * pair + 1;
        const bool hasB = bB < B;
        // the channel's spectrum, lane-packed (launch_spectrum_lanepack): quad (e, thread) = (re, re, im, im) of the two bins of
        // butterfly pair e -- one 16-byte load straight into the registers of the product, no shuffles
        const float4* kfc = reinterpret_cast<const float4*>(kf) + (size_t)c * (N / 2);
        float sw[3][3], sb[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int e = 0; e < 3; ++e) sw[q][e] = short_w[(q * D + c) * 3 + e];
            sb[q] = short_b[q * D + c];
        }
        if constexpr (IDS) {
            if (ltid < 48) zt[ltid] = ztab[(size_t)(ltid & 15) * D3 + (ltid >> 4) * D + c];
            __syncthreads();
        }
        // ---------------------------------------------------------------- phase A: gate from the rows requested a unit ago
        // FULL (uniform): every token a thread touches exists in both reads of the pair (L >= N/2, two reads) -- true for every
        // unit of an even batch of 8k-bp reads: the ~120 per-element selects of the general form drop out
        const bool full_unit = L >= HALF && hasB;
        auto phase_a = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            float gAt = 0.f, gBt = 0.f;
            if constexpr (GATED) {
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (ltid + ch * NT);
                    float gA[8], gB[8];
                    cvt8<T>(cur.g[ch][0], gA);
                    cvt8<T>(cur.g[ch][1], gB);
                    if constexpr (LO) {
                        add_lo8(cur.gl[ch][0], gA);
                        add_lo8(cur.gl[ch][1], gB);
                    }
                    if constexpr (!FULL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            gA[e] = t0 + e < L ? gA[e] : 0.f;
                            gB[e] = (hasB && t0 + e < L) ? gB[e] : 0.f;
                        }
                    }
                    lds_store8(bre + pad_index(t0), gA);
                    lds_store8(bim + pad_index(t0), gB);
                }
                if (tail && ltid == TAIL_TID) {
                    float l0 = 0.f, l1 = 0.f;
                    if constexpr (LO) l0 = lo1b(cur.gltail[0]), l1 = lo1b(cur.gltail[1]);
                    gtail[0] = to_float(cur.gtail[0]) + l0;
                    gtail[1] = hasB ? to_float(cur.gtail[1]) + l1 : 0.f;
                }
            } else {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int t0 = 8 * (ltid + ch * NT);
                const bool valid = FULL ? true : t0 < L, validB = FULL ? true : (valid && hasB);
                float xa[3][10], xb[3][10], gA[8], gB[8], x1[8], v[8], x1b[8], vb[8];
                if constexpr (IDS) {
                    ids_decode<true, 6>(cur.idd[ch][0], cur.idp[ch][0], t0, valid, zt, xa);
                    ids_decode<true, 6>(cur.idd[ch][1], cur.idp[ch][1], t0, validB, zt, xb);
                } else {
#pragma unroll
                    for (int a3 = 1; a3 < 3; ++a3) {
                        raw_decode(cur.r[ch][0][a3 - 1], t0, valid, xa[a3]);
                        raw_decode(cur.r[ch][1][a3 - 1], t0, validB, xb[a3]);
                    }
                }
                fir3_pair(xa[1], xb[1], sw[1][0], sw[1][1], sw[1][2], sb[1], x1, x1b);
                fir3_pair(xa[2], xb[2], sw[2][0], sw[2][1], sw[2][2], sb[2], v, vb);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gA[e] = (FULL || t0 + e < L) ? v[e] * x1[e] : 0.f;
                    gB[e] = (FULL || (hasB && t0 + e < L)) ? vb[e] * x1b[e] : 0.f;
                }
                lds_store8(bre + pad_index(t0), gA);
                lds_store8(bim + pad_index(t0), gB);
                if (ch == CH - 1 && tail && ltid == TAIL_TID) {   // token HALF: taps are x[8], x[9] of this chunk and z[HALF]
                    float ta[3] = {0.f, 0.f, 0.f}, tb[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int a3 = 1; a3 < 3; ++a3) {
                        const float za = IDS ? zt[a3 * 16 + (cur.idt[0] & 15)] : to_float(cur.ztail[0][a3 - 1]);
                        const float zb = IDS ? zt[a3 * 16 + (cur.idt[1] & 15)] : to_float(cur.ztail[1][a3 - 1]);
                        ta[a3] = sb[a3] + sw[a3][0] * xa[a3][8] + sw[a3][1] * xa[a3][9] + sw[a3][2] * za;
                        tb[a3] = sb[a3] + sw[a3][0] * xb[a3][8] + sw[a3][1] * xb[a3][9] + sw[a3][2] * zb;
                    }
                    gAt = ta[1] * ta[2];
                    gBt = hasB ? tb[1] * tb[2] : 0.f;
                }
            }
            // (gtail is its own LDS words: the previous unit's phase C, its last reader, is behind the barrier that ended that unit)
            if (tail && ltid == TAIL_TID) {
                gtail[0] = gAt;
                gtail[1] = gBt;
            }
            }
        };
        if (full_unit) phase_a(std::true_type{});
        else phase_a(std::false_type{});
        __syncthreads();

        // ---------------------------------------------------------------- phase B: FFT, spectrum product, inverse FFT
        Cx2 v[16];
        {
            int Ns = 1;
#pragma unroll
            for (int p = 0; p < P::NPASS - 1; ++p) {
                if (p == 0) {
                    pass_first_lower<LOGN>(bre, bim, v, ltid);
                } else {
                    pass_load<LOGN, 16>(bre, bim, v, ltid);
                    pass_compute_w<LOGN, 16, false>(v, ltid, true, wall + TL::fwd(p));
                }
                __syncthreads();
                pass_store<LOGN, 16>(bre, bim, v, ltid, Ns);
                __syncthreads();
                Ns *= 16;
            }
        }
        {
            Cx2 kv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float4 q4 = kfc[e * NT + ltid];
                kv[e] = Cx2{make_v2(q4.x, q4.y), make_v2(q4.z, q4.w)};
            }
            pass_load<LOGN, LAST>(bre, bim, v, ltid);
            pass_compute_w<LOGN, LAST, false>(v, ltid, true, wall + TL::fwd_last());
            spectrum_multiply_and_first_inverse_v<LOGN, LAST>(v, ltid, kv, tail ? gtail[0] : 0.f, tail ? gtail[1] : 0.f);
        }
        __syncthreads();
        pass_store<LOGN, LAST>(bre, bim, v, ltid, 1);
        __syncthreads();
        {
            int Ns = LAST;
#pragma unroll
            for (int p = 1; p < P::NPASS - 1; ++p) {
                pass_load<LOGN, 16>(bre, bim, v, ltid);
                pass_compute_w<LOGN, 16, true>(v, ltid, true, wall + TL::inv(p));
                __syncthreads();
                pass_store<LOGN, 16>(bre, bim, v, ltid, Ns);
                __syncthreads();
                Ns *= 16;
            }
        }
        // ---- requests that have the last pass to land: x0 rows of this unit, gate rows of this workgroup's next unit
        Raw<T> x0r[CH][2];
        T x0tail[2];
        uint2 x0l[CH][2];                                // GATED + LO: the lo bytes of x0f
        unsigned char x0ltail[2] = {0, 0};
        const T* zA = z + (size_t)bA * D3 * Lp;
        const T* zB = z + (size_t)(hasB ? bB : bA) * D3 * Lp;
        if constexpr (GATED) {                           // (x0f: no history needed -- only the 16-byte vector of Raw<T> is used)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const T* row = (rd == 0 ? zA : zB) + (size_t)c * Lp;
                const unsigned char* rowl = zlo_row(rd == 0 ? zA : zB, 0, c, Lp);
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (ltid + ch * NT);
                    x0r[ch][rd].d = *reinterpret_cast<const uint4*>(row + (t0 < L ? t0 : 0));
                    if constexpr (LO) x0l[ch][rd] = *reinterpret_cast<const uint2*>(rowl + (t0 < L ? t0 : 0));
                }
                x0tail[rd] = row[(tail && ltid == TAIL_TID) ? HALF : 0];
                if constexpr (LO) x0ltail[rd] = rowl[(tail && ltid == TAIL_TID) ? HALF : 0];
            }
        } else if constexpr (!IDS) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const T* row = (rd == 0 ? zA : zB) + (size_t)c * Lp;
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = 8 * (ltid + ch * NT);
                    raw_load(x0r[ch][rd], row, t0, t0 < L);
                }
                x0tail[rd] = row[(tail && ltid == TAIL_TID) ? HALF : 0];
            }
        }
        ConvGateRaw<T, IDS, GATED, LO> nxt;
        {
            const int un = u + stride;
            request_gate(un < n_mine ? un : u, nxt, ltid);       // clamped: unconditional loads
        }
        pass_load<LOGN, 16>(bre, bim, v, ltid);
        pass_compute_last_inverse_lower<LOGN>(v, ltid, wall + TL::inv(P::NPASS - 1));
        __syncthreads();
        pass_store_lower<LOGN>(bre, bim, v, ltid, tail);
        __syncthreads();

        // ---------------------------------------------------------------- phase C: x0 through the short filter, gate, store
        auto phase_c = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            T* yA = y + ((size_t)bA * D + c) * Lp;
            T* yB = y + ((size_t)(hasB ? bB : bA) * D + c) * Lp;
            unsigned char* ylA = LO ? ylo + ((size_t)bA * D + c) * Lp : nullptr;
            unsigned char* ylB = LO ? ylo + ((size_t)(hasB ? bB : bA) * D + c) * Lp : nullptr;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int t0 = 8 * (ltid + ch * NT);
                const bool valid = FULL ? true : t0 < L, validB = FULL ? true : (valid && hasB);
                float xa[10] = {}, xb[10] = {}, x0a[8], x0b[8];
                if constexpr (GATED) {
                    cvt8<T>(x0r[ch][0].d, x0a);
                    cvt8<T>(x0r[ch][1].d, x0b);
                    if constexpr (LO) {
                        add_lo8(x0l[ch][0], x0a);
                        add_lo8(x0l[ch][1], x0b);
                    }
                } else if constexpr (IDS) {
                    float xa3[3][10], xb3[3][10];
                    // opaque copies: otherwise the 40 extracted ids (common subexpressions with phase A) are kept across the
                    // transform -- in scratch
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd)
                        asm volatile("" : "+v"(cur.idd[ch][rd].x), "+v"(cur.idd[ch][rd].y), "+v"(cur.idp[ch][rd]));
                    ids_decode<true, 1>(cur.idd[ch][0], cur.idp[ch][0], t0, valid, zt, xa3);
                    ids_decode<true, 1>(cur.idd[ch][1], cur.idp[ch][1], t0, validB, zt, xb3);
#pragma unroll
                    for (int e = 0; e < 10; ++e) xa[e] = xa3[0][e], xb[e] = xb3[0][e];
                } else {
                    raw_decode(x0r[ch][0], t0, valid, xa);
                    raw_decode(x0r[ch][1], t0, validB, xb);
                }
                if constexpr (!GATED) fir3_pair(xa, xb, sw[0][0], sw[0][1], sw[0][2], sb[0], x0a, x0b);
                if (t0 < Lp) {
                    float oA[8], oB[8];
                    lds_load8(bre + pad_index(t0), oA);
                    lds_load8(bim + pad_index(t0), oB);
                    if (tail && t0 == 0) {  // remove the one wrapped product k[L-1]*g[L-1] from output 0
                        const float kl = ktime[(size_t)(L - 1) * D + c];
                        oA[0] -= kl * gtail[0];
                        oB[0] -= kl * gtail[1];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = FULL || t0 + e < L;
                        oA[e] = ok ? oA[e] * x0a[e] : 0.f;
                        oB[e] = ok ? oB[e] * x0b[e] : 0.f;
                    }
                    ystore8<T, LO>(yA + t0, ylA + t0, oA);
                    if (hasB) ystore8<T, LO>(yB + t0, ylB + t0, oB);
                }
                if (ch == CH - 1 && tail && ltid == TAIL_TID) {     // token HALF: x0 from x[8], x[9] of this chunk and z[HALF]
                    float za = IDS ? zt[cur.idt[0] & 15] : to_float(x0tail[0]);
                    float zb = IDS ? zt[cur.idt[1] & 15] : to_float(x0tail[1]);
                    if constexpr (GATED && LO) za += lo1b(x0ltail[0]), zb += lo1b(x0ltail[1]);
                    const float x0At = GATED ? za : sb[0] + sw[0][0] * xa[8] + sw[0][1] * xa[9] + sw[0][2] * za;
                    const float x0Bt = GATED ? zb : sb[0] + sw[0][0] * xb[8] + sw[0][1] * xb[9] + sw[0][2] * zb;
                    ystore1<T, LO>(yA + HALF, ylA + HALF, bre[pad_index(HALF)] * x0At);
                    if (hasB) ystore1<T, LO>(yB + HALF, ylB + HALF, bim[pad_index(HALF)] * x0Bt);
                }
            }
        };
        if (full_unit) phase_c(std::true_type{});
        else phase_c(std::false_type{});
        cur = nxt;
        __syncthreads();                                  // the buffer (and zt / gtail) are rewritten by the next unit
    }
}

template <typename T, bool IDS, bool GATED = false, bool LO = false>
static void launch_conv_pers_inst(const void* z, void* y, const float2* kf, const float2* tw, const float* ktime,
                                  const float* short_w, const float* short_b, int B, int L, int Lp, const unsigned char* ids8,
                                  const float* ztab, int use_xcd, hipStream_t st, unsigned char* ylo = nullptr) {
    using P = Plan<14>;
    constexpr size_t lds = (size_t)2 * padded_size(P::N) * sizeof(float) + 256;
    auto kern = hyena_conv_pers_kernel<T, IDS, GATED, LO>;
    CLM_SET_LDS(kern, lds);
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int n_units = ((B + 1) / 2) * D;
    dim3 grid(n_units < cus ? n_units : cus), block(P::NT);
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const T*>(z), reinterpret_cast<T*>(y), kf, tw, ktime, short_w,
                       short_b, B, L, Lp, ids8, ztab, use_xcd, ylo);
}

// (Round 2 built the 16384-point convolution of 8k reads as two 8192-point problems over the even / odd bins -- hyena_conv_eo_kernel --
//  and round 3 the same decomposition for 16,384-token segments of long reads -- hyena_conv_seg16_kernel.  Both were correct and both
//  MEASURED SLOWER than the kernels in this file (8k: 15.4 vs 13.5 ms per batch; 32k: 2.18 vs 1.90 ms per launch at 4.7 instead of
//  5.8 GB); they were removed in round 4.  The numbers and the reasons are in HISTORY.md section 4.6 / 8 and profiles/r03_seg16.txt;
//  the code is in the history at commit 8ca4f7f.)

typedef unsigned v4u32 __attribute__((vector_size(16)));   // the type the buffer builtins take and return
// buffer descriptor over [p, p + bytes): every input wave-uniform, made provably so with readfirstlane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// Partition spectrum j of every channel from natural bin order (src [256][N]) into the lane-packed layout the segmented kernel
// reads (dst [256][KS][16][512] quads): quad (e, t) = (re, re, im, im) of bins t + 512 ka and t + 512 (ka + 1),
// ka = 2 (e / 4) + 8 (e % 4) -- the two butterflies thread t holds in pair e of the last forward pass (radix 4, 512 threads).
__global__ __launch_bounds__(512) void spectrum_lanepack_kernel(const float2* __restrict__ src, float4* __restrict__ dst, int KS, int j) {
    constexpr int N = 16384, NT = 512, LAST = Plan<14>::LAST;
    static_assert(LAST == 4 && Plan<14>::NT == NT, "pair geometry of the 16384-point plan");
    const int c = blockIdx.x, t = threadIdx.x;
    const float2* s = src + (size_t)c * N;
    float4* d = dst + ((size_t)c * KS + j) * (N / 2);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ka = 2 * (e / LAST) + (N / LAST / NT) * (e % LAST);
        const float2 a = s[t + NT * ka], b = s[t + NT * (ka + 1)];
        d[e * NT + t] = make_float4(a.x, b.x, a.y, b.y);
    }
}
void launch_spectrum_lanepack(const float2* src, float2* dst, int KS, int j, hipStream_t st) {
    hipLaunchKernelGGL(spectrum_lanepack_kernel, dim3(D), dim3(512), 0, st, src, reinterpret_cast<float4*>(dst), KS, j);
}

// ================================================================================================ long reads
// L > 8193 tokens does not fit one LDS-resident transform.  Uniformly partitioned convolution over S segments of Ls = 8192:
//     y[m*Ls + t] = IFFT( sum_{i <= m} FFT(g_i) . K'_{m-i} )[t],   0 <= t < Ls
// with g_i the i-th segment of the gated signal (zero-padded to N = 2 Ls) and K'_j the spectrum of the filter taps
// [j Ls, (j+1) Ls) in the lower and [(j-1) Ls, j Ls) in the upper half of the transform (launch_filter_spectrum with prev_off):
// the circular wrap of that upper half is exactly what plain overlap-add would carry over from the previous output segment
// (K'_j = K_j + (-1)^k K_{j-1}), so nothing is carried in the time domain, only the lower half of each inverse transform is
// needed (pruned last pass) and a segment moves 384 KB instead of 512 KB through HBM.  One workgroup walks the segments of
// its (channel, read pair) in order; the segment spectra G_i live in a global scratch that each thread only ever re-reads
// where it wrote itself (no fences needed); the last segment's spectrum is never stored.
// LONE: L = S*SEG_LEN + 1.  The S segments produce the outputs t < L-1 (causal: they need neither g[L-1] nor the taps beyond
// S*SEG_LEN); the one remaining output is the full-length dot product y[L-1] = x0[L-1] * sum_t g[t] krev[c][t], whose partial
// sums ride along with phase A of every segment -- instead of a whole transform pipeline for a single token (S+1 segments:
// +25 % of the kernel at 32769 tokens, +50 % at 16385).
// IDS (16-bit modes, block 0): x0 | x1 | v looked up in ztab by token id, as in hyena_conv_kernel.
// GATED: z holds x0f / g, filtered and gated by the producer (see hyena_conv_pers_kernel).
// LO (fp16c): as hyena_conv_kernel -- y leaves as hi + lo bytes, the gated rows are read as hi + lo.
template <typename T, bool LONE, bool IDS, bool GATED = false, bool LO = false>
__global__ __launch_bounds__(Plan<14>::NT) void hyena_conv_seg_kernel(
    const T* __restrict__ z, T* __restrict__ y, const float2* __restrict__ kf /*[256][KS][N] lane-packed (launch_spectrum_lanepack), KS >= S partitions stored*/,
    int KS, const float2* __restrict__ tw, const float* __restrict__ short_w, const float* __restrict__ short_b,
    float2* __restrict__ gscratch /*[pairs][256][S][N]*/, int B, int L, int Lp, int S,
    const float* __restrict__ krev /*[256][krev_stride], LONE only*/, int krev_stride,
    const unsigned char* __restrict__ ids8 /*[B][Lp], IDS only*/, const float* __restrict__ ztab /*[16][768]*/, int use_xcd,
    unsigned char* __restrict__ ylo, const SegPrefix pfx) {
    static_assert(!LO || std::is_same<T, f16_t>::value, "lo bytes exist in the compensated fp16 mode only");
    constexpr int LOGN = 14;
    using P = Plan<LOGN>;
    using TL = TwLayout<LOGN>;
    constexpr int N = P::N, NT = P::NT, LAST = P::LAST, HALF = N / 2, CH = HALF / 8 / NT;
    static_assert(HALF == SEG_LEN && CH == 2, "segment = half transform");
    static_assert(NT == SEG_DOT_THREADS, "one partial dot product per thread in SegPrefix::dots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bre = reinterpret_cast<float*>(smem);             // read A of the pair = real parts
    float* bim = bre + padded_size(N);                       // read B            = imaginary parts

    const int tid = threadIdx.x;
    // XCD-aware order (see hyena_conv_pers_kernel): all read pairs of a channel on one XCD, so that its S partition spectra
    // (S x 128 KB, read (S+1)S/2 times per unit) are served by that XCD's L2 instead of being fetched by all eight
    const int pairs = (B + 1) / 2, xcd = blockIdx.x % XCDS, slot = blockIdx.x / XCDS;
    const int c = use_xcd ? XCDS * (slot / pairs) + xcd : (int)blockIdx.x / pairs, pair = use_xcd ? slot % pairs : (int)blockIdx.x % pairs;
    // the two reads of the pair: batch order, or (pfx.perm, round 5) the order of descending [PAD] prefix -- reads with long prefixes
    // share a pair, and a pair skips the segments inside BOTH prefixes
    const bool hasB = 2 * pair + 1 < B;
    const int bA = pfx.perm ? pfx.perm[2 * pair] : 2 * pair;
    const int bB = hasB ? (pfx.perm ? pfx.perm[2 * pair + 1] : 2 * pair + 1) : bA;
    const T* zA = z + (size_t)bA * D3 * Lp;
    const T* zB = z + (size_t)(hasB ? bB : bA) * D3 * Lp;
    T* yA = y + ((size_t)bA * D + c) * Lp;
    T* yB = y + ((size_t)(hasB ? bB : bA) * D + c) * Lp;
    unsigned char* ylA = LO ? ylo + ((size_t)bA * D + c) * Lp : nullptr;
    unsigned char* ylB = LO ? ylo + ((size_t)(hasB ? bB : bA) * D + c) * Lp : nullptr;
    const float2* kfc = kf + (size_t)c * KS * N;
    float2* gs = gscratch + ((size_t)pair * D + c) * S * N;
    const float* kr = LONE ? krev + (size_t)c * krev_stride : nullptr;
    float dotA = 0.f, dotB = 0.f;
    // Round 5 ([PAD]-prefix reuse, pad_prefix.hip): the segments that lie wholly inside the [PAD] prefix of BOTH reads of the pair
    // are not transformed -- their spectra are the same in every pair (the all-[PAD] table's, times 1 + i for the two reads packed
    // as re / im: pfx.tab holds them in that form and the partition products read them there instead of in this unit's scratch),
    // their outputs lie inside tail tiles nobody computes, and their share of the last token's dot product is the table's
    // per-thread partial sum.  m_start = the segment
    // that holds the tile BEFORE the pair's first non-prefix tile (the tail kernel computes that tile for its filter history).
    int m_start = 0;
    if (pfx.p0 != nullptr && hasB) {
        const int pa = pfx.p0[bA], pb = pfx.p0[bB], pm = pa < pb ? pa : pb;
        m_start = pm > 0 ? ((pm - 1) * 128) / SEG_LEN : 0;
        if (m_start > S - 1) m_start = S - 1;
    }
    m_start = __builtin_amdgcn_readfirstlane(m_start);
    if constexpr (LONE) {
        if (m_start > 0) dotA = dotB = pfx.dots_in[((size_t)c * pfx.dots_segs + (m_start - 1)) * NT + threadIdx.x];
    }

    float sw[3][3], sb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int e = 0; e < 3; ++e) sw[q][e] = short_w[(q * D + c) * 3 + e];
        sb[q] = short_b[q * D + c];
    }
    float* zt = bim + padded_size(N) + 4;                    // IDS: [3][16] rows x0 | x1 | v of this channel
    const unsigned char* irA = IDS ? ids8 + (size_t)bA * Lp : nullptr;
    const unsigned char* irB = IDS ? ids8 + (size_t)(hasB ? bB : bA) * Lp : nullptr;
    if constexpr (IDS) {
        if (tid < 48) zt[tid] = ztab[(size_t)(tid & 15) * D3 + (tid >> 4) * D + c];
        __syncthreads();
    }

    // buffer descriptors of this unit's spectrum scratch and this channel's partition spectra (uniform: blockIdx + arguments)
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(gs, (size_t)S * N * sizeof(float2));
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kfc, (size_t)KS * N * sizeof(float2));
    // (the table's spectra of this channel, pair form; without a table: the scratch again -- never read, m_start is 0 then)
    const __amdgpu_buffer_rsrc_t t_rs = pfx.tab != nullptr ? make_rsrc(pfx.tab + (size_t)c * pfx.tab_segs * N, (size_t)pfx.tab_segs * N * sizeof(float2))
                                                            : g_rs;
    // GATED (round 4): the g rows of segment m + 1 are requested before the last inverse pass of segment m and
    // consumed in its phase A -- the HBM round trip that opened every segment (the prefetch that took the 8k kernel from 0.81 to
    // 0.57 ms) -- unconditionally: the last segment re-requests itself (a conditional request makes the waitcnt pass drain to zero)
    uint4 gcur[GATED ? CH : 1][2];
    uint2 glcur[(GATED && LO) ? CH : 1][2];
    auto request_g = [&](int mseg, uint4 (&g)[GATED ? CH : 1][2], uint2 (&gl)[(GATED && LO) ? CH : 1][2], int ltid_) {
        if constexpr (GATED) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const T* row = (rd == 0 ? zA : zB) + (size_t)(D + c) * Lp;
                const unsigned char* rowl = zlo_row(rd == 0 ? zA : zB, 1, c, Lp);
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) {
                    const int t0 = mseg * SEG_LEN + 8 * (ltid_ + ch * NT);
                    g[ch][rd] = *reinterpret_cast<const uint4*>(row + (t0 < L ? t0 : 0));
                    if constexpr (LO) gl[ch][rd] = *reinterpret_cast<const uint2*>(rowl + (t0 < L ? t0 : 0));
                }
            }
        }
    };
    request_g(m_start, gcur, glcur, tid);
#pragma unroll 1
    for (int m = m_start; m < S; ++m) {
        const int seg0 = m * SEG_LEN;
        // Launder the thread index once per segment: otherwise LICM hoists every LDS/global address of all seven
        // passes out of this loop and ~1000 VGPRs spill.
        int ltid = tid;
        asm volatile("" : "+v"(ltid));
        int lz = 0;                        // same for the id table and the id rows (IDS): an opaque zero added to their bases
        asm volatile("" : "+v"(lz));
        const float* ztm = zt + lz;
        const unsigned char* irAm = IDS ? irA + lz : nullptr;
        const unsigned char* irBm = IDS ? irB + lz : nullptr;
        // ---- phase A: segment m of the gated signal into the lower half, zeros above
        // (x0 only gates the output: its rows are requested before the last inverse pass and filtered in phase C -- held from
        //  here, its 32 registers were the bulk of 80-135 spilled registers per thread, i.e. scratch traffic per segment)
        uint2 idd[CH][2];                                    // IDS: 8 token ids of the chunk
        unsigned short idp[CH][2];                           //      and the two before it
        static_assert(!(GATED && (IDS || std::is_same<T, float>::value)), "the gated hand-over exists in the fused 16-bit path only");
        if constexpr (GATED) {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int tl = 8 * (ltid + ch * NT), t0 = seg0 + tl;
                float gA[8], gB[8];
                cvt8<T>(gcur[ch][0], gA);
                cvt8<T>(gcur[ch][1], gB);
                if constexpr (LO) {
                    add_lo8(glcur[ch][0], gA);
                    add_lo8(glcur[ch][1], gB);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gA[e] = (t0 + e < L) ? gA[e] : 0.f;
                    gB[e] = (hasB && t0 + e < L) ? gB[e] : 0.f;
                }
                lds_store8(bre + pad_index(tl), gA);
                lds_store8(bim + pad_index(tl), gB);
            }
        } else if constexpr (!std::is_same<T, float>::value) {
            // all loads of the segment up front, no control flow in between (see Raw<T>): four serialized HBM round trips otherwise
            Raw<T> raw[IDS ? 1 : CH][2][IDS ? 1 : 3];
            if constexpr (IDS) {
#pragma unroll
                for (int rd = 0; rd < 2; ++rd)
#pragma unroll
                    for (int ch = 0; ch < CH; ++ch) {
                        const int t0 = seg0 + 8 * (ltid + ch * NT);
                        const unsigned char* ir = rd == 0 ? irAm : irBm;
                        idd[ch][rd] = *reinterpret_cast<const uint2*>(ir + (t0 < L ? t0 : 0));
                        idp[ch][rd] = *reinterpret_cast<const unsigned short*>(ir + ((t0 < L && t0 > 0) ? t0 - 2 : 0));
                    }
            } else {
#pragma unroll
                for (int rd = 0; rd < 2; ++rd)
#pragma unroll
                    for (int a3 = 1; a3 < 3; ++a3) {
                        const T* row = (rd == 0 ? zA : zB) + (size_t)(a3 * D + c) * Lp;
#pragma unroll
                        for (int ch = 0; ch < CH; ++ch) {
                            const int t0 = seg0 + 8 * (ltid + ch * NT);
                            raw_load(raw[ch][rd][a3], row, t0, t0 < L);
                        }
                    }
            }
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int tl = 8 * (ltid + ch * NT), t0 = seg0 + tl;
                const bool valid = t0 < L;
                float xa[3][10], xb[3][10], gA[8], gB[8], x1[8], v[8];
                if constexpr (IDS) {
                    ids_decode<true, 6>(idd[ch][0], idp[ch][0], t0, valid, ztm, xa);
                    ids_decode<true, 6>(idd[ch][1], idp[ch][1], t0, valid && hasB, ztm, xb);
                } else {
#pragma unroll
                    for (int a3 = 1; a3 < 3; ++a3) {
                        raw_decode(raw[ch][0][a3], t0, valid, xa[a3]);
                        raw_decode(raw[ch][1][a3], t0, valid && hasB, xb[a3]);
                    }
                }
                float x1b[8], vb[8];
                fir3_pair(xa[1], xb[1], sw[1][0], sw[1][1], sw[1][2], sb[1], x1, x1b);
                fir3_pair(xa[2], xb[2], sw[2][0], sw[2][1], sw[2][2], sb[2], v, vb);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gA[e] = (t0 + e < L) ? v[e] * x1[e] : 0.f;
                    gB[e] = (hasB && t0 + e < L) ? vb[e] * x1b[e] : 0.f;
                }
                lds_store8(bre + pad_index(tl), gA);
                lds_store8(bim + pad_index(tl), gB);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int tl = 8 * (ltid + ch * NT), t0 = seg0 + tl;
                float gA[8], gB[8];
                if (t0 < L) {
                    float x1[8], v[8];
                    short_filter8<T>(zA + (size_t)(1 * D + c) * Lp, t0, sw[1][0], sw[1][1], sw[1][2], sb[1], x1);
                    short_filter8<T>(zA + (size_t)(2 * D + c) * Lp, t0, sw[2][0], sw[2][1], sw[2][2], sb[2], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gA[e] = (t0 + e < L) ? v[e] * x1[e] : 0.f;
                    if (hasB) {
                        short_filter8<T>(zB + (size_t)(1 * D + c) * Lp, t0, sw[1][0], sw[1][1], sw[1][2], sb[1], x1);
                        short_filter8<T>(zB + (size_t)(2 * D + c) * Lp, t0, sw[2][0], sw[2][1], sw[2][2], sb[2], v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gB[e] = (t0 + e < L) ? v[e] * x1[e] : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gB[e] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gA[e] = 0.f, gB[e] = 0.f;
                }
                lds_store8(bre + pad_index(tl), gA);
                lds_store8(bim + pad_index(tl), gB);
            }
        }
        // Pass twiddles: NOT held across the segment (88 registers that the spectrum product loop needs for its load buffers):
        // the forward ones are requested once phase A has used its row registers (the last token's dot product and the barrier
        // cover the L2 round trip); the inverse ones after the product loop.
        Cx2 wall[TL::TOTAL];
        {
            int ns = 16;
#pragma unroll
            for (int p = 1; p <= P::NPASS - 2; ++p) {
                pass_twiddles<LOGN, 16, false>(wall + TL::fwd(p), ltid, ns, tw);
                ns *= 16;
            }
            pass_twiddles<LOGN, LAST, false>(wall + TL::fwd_last(), ltid, ns, tw);
        }
        if constexpr (LONE) {
            // this segment's share of the last output's dot product: each thread re-reads the 16 gated samples it has just
            // written (own data: no barrier needed) -- in phase A itself the 16 extra registers doubled the spills
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const int tl = 8 * (ltid + ch * NT);
                float gA[8], gB[8], kk[8];
                lds_load8(bre + pad_index(tl), gA);
                lds_load8(bim + pad_index(tl), gB);
                load8<float>(kr + seg0 + tl, kk);
#pragma unroll
                for (int e = 0; e < 8; ++e) dotA = fmaf(gA[e], kk[e], dotA), dotB = fmaf(gB[e], kk[e], dotB);
            }
            // the forward that fills an all-[PAD] table (one read: read A) leaves the running sum after every segment but the last
            if (pfx.dots_out != nullptr && m + 1 < S) pfx.dots_out[((size_t)c * pfx.dots_segs + m) * NT + ltid] = dotA;
        }
        __syncthreads();                 // (upper half = zero padding: never written, never read -- pass_first_lower)

        // ---- forward transform
        Cx2 v[16];
        {
            int Ns = 1;
#pragma unroll
            for (int p = 0; p < P::NPASS - 1; ++p) {
                if (p == 0) {
                    pass_first_lower<LOGN>(bre, bim, v, ltid);
                } else {
                    pass_load<LOGN, 16>(bre, bim, v, ltid);
                    pass_compute_w<LOGN, 16, false>(v, ltid, true, wall + TL::fwd(p));
                }
                __syncthreads();
                pass_store<LOGN, 16>(bre, bim, v, ltid, Ns);
                __syncthreads();
                Ns *= 16;
            }
        }
        // ---- spectrum bookkeeping: keep G_m, form P_m = sum_i G_i K_{m-i}, first inverse butterfly
        using G = PassGeom<LOGN, LAST>;
        static_assert(G::NP * LAST == 16 && G::IT == 2 * G::NP, "16 full pairs per thread");
        // A thread's 32 bins are ltid + 512 k, k = 0 .. 31; pair e (one Cx2) holds k = 2p + 8r and 2p + 1 + 8r, p = e / LAST,
        // r = e % LAST.  Both the unit's scratch (G) and the channel's partition spectra (K', launch_spectrum_lanepack) are laid
        // out LANE-PACKED: [segment][e][ltid] quads (re_a, re_b, im_a, im_b) = a Cx2 as the butterflies hold it -- one 16-byte
        // load per pair, no register shuffles between a load and its use.
        // Buffer addressing: descriptor (scalar) + ONE lane offset register + the row (segment, e) in the scalar offset.  As 32
        // per-bin 64-bit flat addresses the compiler kept them live around the whole product loop: 80-135 spilled registers.
        static_assert(NT == 512 && (N / LAST) % NT == 0 && sizeof(Cx2) == 16, "pair row = 512 quads");
        const int loff = ltid * (int)sizeof(Cx2);
        auto quad = [&](const __amdgpu_buffer_rsrc_t& rs, int seg, int e) -> Cx2 {
            const v4u32 r = __builtin_amdgcn_raw_buffer_load_b128(rs, loff, (seg * (N / 2) + NT * e) * (int)sizeof(Cx2), 0);
            const unsigned r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];   // (__builtin_bit_cast straight from a vector element
            return Cx2{make_v2(__uint_as_float(r0), __uint_as_float(r1)),       //  reads element 0 with hipcc 7.2)
                       make_v2(__uint_as_float(r2), __uint_as_float(r3))};
        };
        auto quad_w = [&](const __amdgpu_buffer_rsrc_t& rs, int seg, int e, Cx2 val) {
            v4u32 r;
            r[0] = __float_as_uint(val.re.x), r[1] = __float_as_uint(val.re.y);
            r[2] = __float_as_uint(val.im.x), r[3] = __float_as_uint(val.im.y);
            __builtin_amdgcn_raw_buffer_store_b128(r, rs, loff, (seg * (N / 2) + NT * e) * (int)sizeof(Cx2), 0);
        };
        Cx2 kv[16];                                // K'_0: requested ahead of the last forward pass
#pragma unroll
        for (int e = 0; e < 16; ++e) kv[e] = quad(k_rs, 0, e);
        pass_load<LOGN, LAST>(bre, bim, v, ltid);
        pass_compute_w<LOGN, LAST, false>(v, ltid, true, wall + TL::fwd_last());
        // P_m = G_m K'_0 + sum_{i < m} G_i K'_{m-i}, a quarter of the thread's pairs (4 of G, 4 of K') per step, the loads of a
        // step issued one step ahead of its multiply-adds (two 32-register buffers; in-order return: vmcnt(8)).
        // sched_barrier + the empty asm on the accumulators pin that order (the asm keeps the IR passes from sinking the
        // multiply-adds to the end of the body, the barrier keeps the machine scheduler from gathering a whole iteration's
        // loads at the top: either way 128 registers of buffers and spills).
        auto fetch = [&](Cx2* ga, Cx2* kb, int i, int qtr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ga[e] = quad(i < m_start ? t_rs : g_rs, i, 4 * qtr + e);      // (uniform select of the descriptor)
                kb[e] = quad(k_rs, m - i, 4 * qtr + e);
            }
        };
        auto accumulate = [&](int qtr, const Cx2* ga, const Cx2* kb) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Cx2& acc = v[4 * qtr + e];
                acc = Cx2::add(acc, Cx2::mul(ga[e], kb[e]));
                asm volatile("" : "+v"(acc.re), "+v"(acc.im));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        Cx2 a0[4], b0[4];
        if (m > 0) fetch(a0, b0, 0, 0);
        if (m + 1 < S) {                          // keep G_m (uniform branch; the last segment's spectrum is never read again)
#pragma unroll
            for (int e = 0; e < 16; ++e) quad_w(g_rs, m, e, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = Cx2::mul(v[e], kv[e]);   // P_m = G_m K'_0
        // (each thread re-reads only quads it wrote itself: no fence.  The last step is peeled: with the request for step i + 1
        //  under a condition, the wait counts at the join are those of the path WITHOUT it -- every step then drains to vmcnt(0))
        auto step = [&](int i, auto next) {
            Cx2 a1[4], b1[4];
            fetch(a1, b1, i, 1);
            accumulate(0, a0, b0);
            fetch(a0, b0, i, 2);
            accumulate(1, a1, b1);
            fetch(a1, b1, i, 3);
            accumulate(2, a0, b0);
            if constexpr (decltype(next)::value) fetch(a0, b0, i + 1, 0);
            accumulate(3, a1, b1);
        };
#pragma unroll 1
        for (int i = 0; i + 1 < m; ++i) step(i, std::true_type{});
        if (m > 0) step(m - 1, std::false_type{});
        {
            int ns = LAST;
#pragma unroll
            for (int p = 1; p <= P::NPASS - 1; ++p) {
                pass_twiddles<LOGN, 16, true>(wall + TL::inv(p), ltid, ns, tw);
                ns *= 16;
            }
        }
#pragma unroll
        for (int p = 0; p < G::NP; ++p) Dft<LAST, true>::run(v + p * LAST);
        __syncthreads();
        pass_store<LOGN, LAST>(bre, bim, v, ltid, 1);
        __syncthreads();
        Raw<T> x0r[CH][2];
        uint2 x0l[CH][2];
        {
            int Ns = LAST;
#pragma unroll
            for (int p = 1; p <= P::NPASS - 1; ++p) {
                if (p == P::NPASS - 1) {   // x0 rows of this segment (and the g rows of the next): the last pass to land
                    if constexpr (GATED) {
                        request_g(m + 1 < S ? m + 1 : m, gcur, glcur, ltid);
#pragma unroll
                        for (int rd = 0; rd < 2; ++rd) {
                            const T* row = (rd == 0 ? zA : zB) + (size_t)c * Lp;
                            const unsigned char* rowl = zlo_row(rd == 0 ? zA : zB, 0, c, Lp);
#pragma unroll
                            for (int ch = 0; ch < CH; ++ch) {
                                const int t0 = seg0 + 8 * (ltid + ch * NT);
                                x0r[ch][rd].d = *reinterpret_cast<const uint4*>(row + (t0 < L ? t0 : 0));
                                if constexpr (LO) x0l[ch][rd] = *reinterpret_cast<const uint2*>(rowl + (t0 < L ? t0 : 0));
                            }
                        }
                    } else if constexpr (!IDS && !std::is_same<T, float>::value) {
#pragma unroll
                        for (int rd = 0; rd < 2; ++rd) {
                            const T* row = (rd == 0 ? zA : zB) + (size_t)c * Lp;
#pragma unroll
                            for (int ch = 0; ch < CH; ++ch) {
                                const int t0 = seg0 + 8 * (ltid + ch * NT);
                                raw_load(x0r[ch][rd], row, t0, t0 < L);
                            }
                        }
                    }
                }
                pass_load<LOGN, 16>(bre, bim, v, ltid);
                if (p == P::NPASS - 1) {   // only the lower half of the outputs is used: half-output butterfly, half the stores
                    pass_compute_last_inverse_lower<LOGN>(v, ltid, wall + TL::inv(p));
                    __syncthreads();
                    pass_store_lower<LOGN>(bre, bim, v, ltid, false);
                } else {
                    pass_compute_w<LOGN, 16, true>(v, ltid, true, wall + TL::inv(p));
                    __syncthreads();
                    pass_store<LOGN, 16>(bre, bim, v, ltid, Ns);
                }
                __syncthreads();
                Ns *= 16;
            }
        }
        // ---- phase C: the lower half is output segment m -> gate with x0 -> y
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const int tl = 8 * (ltid + ch * NT), t0 = seg0 + tl;
            const bool valid = t0 < L;
            float x0A[8], x0B[8];
            if constexpr (GATED) {
                cvt8<T>(x0r[ch][0].d, x0A);
                cvt8<T>(x0r[ch][1].d, x0B);
                if constexpr (LO) {
                    add_lo8(x0l[ch][0], x0A);
                    add_lo8(x0l[ch][1], x0B);
                }
            } else if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x0A[e] = 0.f, x0B[e] = 0.f;
                if (valid) {
                    short_filter8<T>(zA + (size_t)c * Lp, t0, sw[0][0], sw[0][1], sw[0][2], sb[0], x0A);
                    if (hasB) short_filter8<T>(zB + (size_t)c * Lp, t0, sw[0][0], sw[0][1], sw[0][2], sb[0], x0B);
                }
            } else {
                float xa[10], xb[10];
                if constexpr (IDS) {
                    float xa3[3][10], xb3[3][10];
                    // opaque copies: otherwise the 40 extracted ids (common subexpressions with phase A) are kept across the
                    // transform -- in scratch
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) asm volatile("" : "+v"(idd[ch][rd].x), "+v"(idd[ch][rd].y), "+v"(idp[ch][rd]));
                    ids_decode<true, 1>(idd[ch][0], idp[ch][0], t0, valid, ztm, xa3);
                    ids_decode<true, 1>(idd[ch][1], idp[ch][1], t0, valid && hasB, ztm, xb3);
#pragma unroll
                    for (int e = 0; e < 10; ++e) xa[e] = xa3[0][e], xb[e] = xb3[0][e];
                } else {
                    raw_decode(x0r[ch][0], t0, valid, xa);
                    raw_decode(x0r[ch][1], t0, valid && hasB, xb);
                }
                fir3_pair(xa, xb, sw[0][0], sw[0][1], sw[0][2], sb[0], x0A, x0B);
            }
            float oA[8], oB[8];
            lds_load8(bre + pad_index(tl), oA);
            lds_load8(bim + pad_index(tl), oB);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = t0 + e < L;
                oA[e] = ok ? oA[e] * x0A[e] : 0.f;
                oB[e] = ok ? oB[e] * x0B[e] : 0.f;
            }
            if (t0 < Lp) {
                ystore8<T, LO>(yA + t0, ylA + t0, oA);
                if (hasB) ystore8<T, LO>(yB + t0, ylB + t0, oB);
            }
        }
        __syncthreads();   // the LDS buffer is refilled by the next segment
    }
    if constexpr (LONE) {
        // y[L-1]: partial dot products of the 512 threads -> wave sums -> one thread; fixed order, deterministic
        const float a = wave_sum(dotA), b = wave_sum(dotB);
        if ((tid & 63) == 0) {
            bre[tid >> 6] = a;
            bim[tid >> 6] = b;
        }
        __syncthreads();
        if (tid == 0) {
            const int t = L - 1;
            float sa = 0.f, sbb = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) sa += bre[w], sbb += bim[w];
            const float k0 = kr[t];                          // tap 0 (+ skip term)
            // short filter of the one token t (t >= 2 here) on row a3 of read rd: from z, or from the id table
            auto filt = [&](int rd, int a3) {
                if constexpr (GATED) {        // a3 = 0: x0f; a3 = 1: g (called as filt(., 2) * filt(., 1): the v slot counts as 1)
                    if (a3 == 2) return 1.0f;
                    float v1 = to_float(((rd == 0 ? zA : zB) + (size_t)(a3 * D + c) * Lp)[t]);
                    if constexpr (LO) v1 += lo1(zlo_row(rd == 0 ? zA : zB, a3, c, Lp), t);
                    return v1;
                } else if constexpr (IDS) {
                    const unsigned char* ir = rd == 0 ? irA : irB;
                    const float* r = zt + a3 * 16;
                    return sb[a3] + sw[a3][0] * r[ir[t - 2] & 15] + sw[a3][1] * r[ir[t - 1] & 15] + sw[a3][2] * r[ir[t] & 15];
                } else {
                    return short_filter1<T>((rd == 0 ? zA : zB) + (size_t)(a3 * D + c) * Lp, t, sw[a3][0], sw[a3][1], sw[a3][2],
                                            sb[a3]);
                }
            };
            ystore1<T, LO>(yA + t, ylA + t, fmaf(filt(0, 2) * filt(0, 1), k0, sa) * filt(0, 0));
            if (hasB) ystore1<T, LO>(yB + t, ylB + t, fmaf(filt(1, 2) * filt(1, 1), k0, sbb) * filt(1, 0));
        }
        for (int t = L + tid; t < Lp; t += NT) {             // padding columns stay zero
            ystore1<T, LO>(yA + t, ylA + t, 0.f);
            if (hasB) ystore1<T, LO>(yB + t, ylB + t, 0.f);
        }
    }
}

template <typename T, bool LONE, bool IDS, bool GATED = false, bool LO = false>
static void launch_conv_seg_inst(const void* z, void* y, const float2* kf, int KS, const float2* tw, const float* short_w,
                              const float* short_b, float2* gscratch, int B, int L, int Lp, int S,
                              const float* krev, int krev_stride, const unsigned char* ids8, const float* ztab,
                              int use_xcd, hipStream_t st, unsigned char* ylo = nullptr, const SegPrefix& pfx = SegPrefix{}) {
    using P = Plan<14>;
    constexpr size_t lds = (size_t)2 * padded_size(P::N) * sizeof(float) + 256;   // + the 3x16 id table
    auto kern = hyena_conv_seg_kernel<T, LONE, IDS, GATED, LO>;
    CLM_SET_LDS(kern, lds);
    static_assert(D % XCDS == 0, "channels split evenly over the XCDs");
    dim3 grid(((B + 1) / 2) * D), block(P::NT);
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const T*>(z), reinterpret_cast<T*>(y), kf, KS, tw,
                       short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, ids8, ztab, use_xcd, ylo, pfx);
}
// LO: fp16c (T = f16_t) with a lo plane for y -- the gated rows then carry lo bytes too
template <typename T, bool LO = false>
static void launch_conv_seg_t(const void* z, void* y, const float2* kf, int KS, const float2* tw, const float* short_w,
                              const float* short_b, float2* gscratch, int B, int L, int Lp, int S, const float* krev,
                              int krev_stride, const unsigned char* ids8, const float* ztab, int use_xcd, hipStream_t st,
                              bool gated, unsigned char* ylo = nullptr, const SegPrefix& pfx = SegPrefix{}) {
#define CLM_SEG(LONE, IDS)                                                                                               \
    launch_conv_seg_inst<T, LONE, IDS, false, LO>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, ids8, ztab, use_xcd, st, ylo, pfx)
    if constexpr (std::is_same<T, float>::value) {           // fp32 mode never takes the id path
        if (krev) CLM_SEG(true, false);
        else CLM_SEG(false, false);
    } else {
        const bool ids = ids8 != nullptr && ztab != nullptr;
        if (gated && !ids) {
            if (krev) launch_conv_seg_inst<T, true, false, true, LO>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, nullptr, nullptr, use_xcd, st, ylo, pfx);
            else launch_conv_seg_inst<T, false, false, true, LO>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, nullptr, nullptr, use_xcd, st, ylo, pfx);
            return;
        }
        if (krev && ids) CLM_SEG(true, true);
        else if (krev) CLM_SEG(true, false);
        else if (ids) CLM_SEG(false, true);
        else CLM_SEG(false, false);
    }
#undef CLM_SEG
}

void launch_hyena_conv_seg(int prec, const void* z, void* y, const float2* kf, int KS, const float2* tw, const float* short_w,
                           const float* short_b, float2* gscratch, int B, int L, int Lp, int S, const float* krev,
                           int krev_stride, const unsigned char* ids8, const float* ztab, hipStream_t st, int flags,
                           unsigned char* ylo, const SegPrefix& pfx) {
    const int use_xcd = !(flags & CONV_NO_XCD);
    const bool gated = (flags & CONV_GATED) != 0;
    if (prec == PREC_F16C && ylo)
        launch_conv_seg_t<f16_t, true>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, ids8, ztab, use_xcd, st, gated, ylo, pfx);
    else if (prec == PREC_F32)
        launch_conv_seg_t<float>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, nullptr, nullptr, use_xcd, st, false, nullptr, pfx);
    else if (prec == PREC_BF16)
        launch_conv_seg_t<bf16_t>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, ids8, ztab, use_xcd, st, gated, nullptr, pfx);
    else
        launch_conv_seg_t<f16_t>(z, y, kf, KS, tw, short_w, short_b, gscratch, B, L, Lp, S, krev, krev_stride, ids8, ztab, use_xcd, st, gated, nullptr, pfx);
}

// Round 5: the all-[PAD] table's segment spectra (one read: G, the convolution scratch of the forward that filled the table) in the form a
// PAIR of such reads has -- the two packed as re / im: (1 + i) G = (re - im, re + im) on the lane-packed quads (re_a, re_b, im_a, im_b).
// In place, once per table; hyena_conv_seg_kernel reads the result where a segment lies inside both reads' [PAD] prefix.
__global__ __launch_bounds__(256) void spectra_pair_form_kernel(float4* __restrict__ tab, size_t nquads) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (size_t)gridDim.x * 256) {
        const float4 t = tab[q];
        tab[q] = make_float4(t.x - t.z, t.y - t.w, t.x + t.z, t.y + t.w);
    }
}
void launch_spectra_pair_form(float2* table, int S_T, hipStream_t st) {
    static_assert(Plan<14>::N == 16384, "segment transform size");
    const size_t nquads = (size_t)D * S_T * (16384 / 2);
    hipLaunchKernelGGL(spectra_pair_form_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<float4*>(table), nquads);
}

// ztab[id][n] = in_proj(LN1(embedding[id]))[n] of block 0, fp32: one workgroup per token id
__global__ __launch_bounds__(256) void ztab_kernel(const float* __restrict__ emb, const float* __restrict__ g,
                                                   const float* __restrict__ bta, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ ztab, float eps) {
    __shared__ float xn[D], red[4];
    const int id = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float x = emb[(size_t)id * D + tid];
    float s = wave_sum(x);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / D);
    __syncthreads();
    const float d = x - mean;
    s = wave_sum(d * d);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float var = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / D);
    xn[tid] = d * (1.0f / sqrtf(var + eps)) * g[tid] + bta[tid];
    __syncthreads();
    for (int n = tid; n < D3; n += 256) {
        const float* wr = w + (size_t)n * D;
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc = fmaf(wr[k], xn[k], acc);
        ztab[(size_t)id * D3 + n] = acc + bias[n];
    }
}
void launch_ztab(const float* emb, const float* g, const float* bta, const float* w, const float* bias, float* ztab,
                 float eps, hipStream_t st) {
    hipLaunchKernelGGL(ztab_kernel, dim3(VOCAB), dim3(256), 0, st, emb, g, bta, w, bias, ztab, eps);
}

static unsigned long long* s_conv_stamp_buf = nullptr;
static size_t s_conv_stamp_wgs = 0;
void conv_dump_stamps() {
    if (!s_conv_stamp_buf || !s_conv_stamp_wgs) return;
    std::vector<unsigned long long> hst(s_conv_stamp_wgs * CONV_NSTAMP);
    if (hipMemcpy(hst.data(), s_conv_stamp_buf, hst.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    double sum[CONV_NSTAMP] = {};
    size_t n = 0;
    for (size_t w = 0; w < s_conv_stamp_wgs; ++w) {
        const unsigned long long* p = &hst[w * CONV_NSTAMP];
        if (!p[0] || !p[11]) continue;
        for (int k = 1; k <= 11; ++k)
            if (k != 6) sum[k] += double(p[k] - p[k == 7 ? 5 : k - 1]);   // slot 6 holds the 100-MHz end stamp
        ++n;
    }
    const char* names[12] = {"", "tw+phaseA", "barriers", "fwd0", "fwd1", "fwd2", "(unused)", "fwd3+kf+inv0", "inv1", "inv2", "inv3", "phaseC"};
    {   // finer split of phase A: 0 -> 12 (twiddle issue + filter taps) -> 13 (z loads, FIR, gate, LDS) -> 14 (zero fill) -> 1 (tail)
        double a[4] = {}; size_t m = 0;
        for (size_t w = 0; w < s_conv_stamp_wgs; ++w) {
            const unsigned long long* p = &hst[w * CONV_NSTAMP];
            if (!p[0] || !p[14]) continue;
            a[0] += double(p[12] - p[0]); a[1] += double(p[13] - p[12]); a[2] += double(p[14] - p[13]); a[3] += double(p[1] - p[14]); ++m;
        }
        if (m) std::fprintf(stderr, "[conv stamps] phase A split: setup %.0f  loads+fir+lds %.0f  zerofill %.0f  tail %.0f\n", a[0] / m, a[1] / m, a[2] / m, a[3] / m);
        double mid = 0; size_t mm = 0;
        for (size_t w = 0; w < s_conv_stamp_wgs; ++w) { const unsigned long long* p = &hst[w * CONV_NSTAMP]; if (p[5] && p[7]) { mid += double(p[7] - p[5]); ++mm; } }
        if (mm) std::fprintf(stderr, "[conv stamps] fwd3+kf+inv0 = %.0f\n", mid / mm);
    }
    {
        double dm = 0, dr = 0;
        for (size_t w = 0; w < s_conv_stamp_wgs; ++w) {
            const unsigned long long* p = &hst[w * CONV_NSTAMP];
            if (!p[0] || !p[11] || !p[6] || !p[15]) continue;
            dm += double(p[11] - p[0]);
            dr += double(p[6] - p[15]);
        }
        if (dr > 0) std::fprintf(stderr, "[conv stamps] s_memtime / s_memrealtime = %.3f -> shader clock %.0f MHz\n", dm / dr, dm / dr * 100.0);
    }
    double tot = 0;
    for (int k = 1; k <= 11; ++k) tot += sum[k] / (n ? n : 1);
    std::fprintf(stderr, "[conv stamps] %zu workgroups, mean s_memtime ticks per phase (total %.0f):\n", n, tot);
    for (int k = 1; k <= 11; ++k)
        if (k != 6) std::fprintf(stderr, "  %-14s %9.0f  %5.1f %%\n", names[k], sum[k] / (n ? n : 1), 100.0 * sum[k] / (n ? n : 1) / tot);
}

// one launch site = one static: the dynamic-LDS attribute is set once per kernel instantiation and device
#define CLM_CONV_LAUNCH(KERN, ...)                                    \
    do {                                                              \
        auto kern_ = KERN;                                            \
        CLM_SET_LDS(kern_, lds);                                      \
        hipLaunchKernelGGL(kern_, grid, block, lds, st, __VA_ARGS__); \
    } while (0)

// LO: fp16c (T = f16_t) with a lo plane for y -- the gated rows then carry lo bytes too
template <int LOGN, typename T, bool LO = false>
static void launch_conv_t(const void* z, void* y, const float2* kf, const float2* tw, const float* ktime,
                          const float* short_w, const float* short_b, int B, int L, int Lp, const unsigned char* ids8,
                          const float* ztab, hipStream_t st, bool gated, unsigned char* ylo = nullptr) {
    using P = Plan<LOGN>;
    constexpr size_t lds = (size_t)2 * padded_size(P::N) * sizeof(float) + 256;   // + g[N/2] pair + the 3x16 id table
    dim3 grid((B + 1) / 2, D), block(P::NT);
    const T* zt = reinterpret_cast<const T*>(z);
    T* yt = reinterpret_cast<T*>(y);
    unsigned long long* const no_stamps = nullptr;
    const unsigned char* const no_ids = nullptr;
    const float* const no_tab = nullptr;
    if constexpr (!std::is_same<T, float>::value) {
        if (gated && !ids8) {
            CLM_CONV_LAUNCH((hyena_conv_kernel<LOGN, T, false, false, true, LO>), zt, yt, kf, tw, ktime, short_w, short_b, B, L, Lp,
                            no_stamps, no_ids, no_tab, ylo);
            return;
        }
    }
    if (ids8) {     // block 0: z looked up by token id (exact fp32 too, round 4: its block 0 then needs no in_proj launch)
        CLM_CONV_LAUNCH((hyena_conv_kernel<LOGN, T, false, true, false, LO>), (const T*)nullptr, yt, kf, tw, ktime, short_w, short_b,
                        B, L, Lp, no_stamps, ids8, ztab, ylo);
        return;
    }
    if constexpr (LOGN == 14 && std::is_same<T, f16_t>::value && !LO) {
        static const bool stamp = debug_flag("stamp");
        if (stamp) {
            const size_t wgs = (size_t)grid.x * grid.y;
            if (wgs > s_conv_stamp_wgs) {
                if (s_conv_stamp_buf) (void)hipFree(s_conv_stamp_buf);
                (void)hipMalloc((void**)&s_conv_stamp_buf, wgs * CONV_NSTAMP * 8);
                s_conv_stamp_wgs = wgs;
            }
            (void)hipMemsetAsync(s_conv_stamp_buf, 0, wgs * CONV_NSTAMP * 8, st);
            CLM_CONV_LAUNCH((hyena_conv_kernel<LOGN, T, true>), zt, yt, kf, tw, ktime, short_w, short_b, B, L, Lp, s_conv_stamp_buf,
                            no_ids, no_tab, (unsigned char*)nullptr);
            return;
        }
    }
    CLM_CONV_LAUNCH((hyena_conv_kernel<LOGN, T, false, false, false, LO>), zt, yt, kf, tw, ktime, short_w, short_b, B, L, Lp, no_stamps,
                    no_ids, no_tab, ylo);
}

template <int LOGN>
static void launch_conv_p(int prec, const void* z, void* y, const float2* kf, const float2* tw, const float* ktime,
                          const float* short_w, const float* short_b, int B, int L, int Lp, const unsigned char* ids8,
                          const float* ztab, hipStream_t st, bool gated, unsigned char* ylo) {
    if (prec == PREC_F16C && ylo)
        launch_conv_t<LOGN, f16_t, true>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, st, gated, ylo);
    else if (prec == PREC_F32)
        launch_conv_t<LOGN, float>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, st, false);
    else if (prec == PREC_BF16)
        launch_conv_t<LOGN, bf16_t>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, st, gated);
    else
        launch_conv_t<LOGN, f16_t>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, st, gated);
}

void launch_hyena_conv(int prec, const void* z, void* y, const float2* kf, const float2* tw, const float* ktime,
                       const float* short_w, const float* short_b, int B, int L, int Lp, int logn,
                       const unsigned char* ids8, const float* ztab, hipStream_t st, int flags, const float2* kf_packed,
                       unsigned char* ylo) {
    // 16384-point class, 16-bit activations: persistent workgroups with next-unit requests (CONV_ONESHOT: one workgroup per
    // unit -- A/B runs; the developer stamps live in that kernel only)
    static const bool stamp = debug_flag("stamp");
    const bool ids = ids8 != nullptr && ztab != nullptr;
    const bool gated = (flags & CONV_GATED) != 0 && !ids;
    const bool lo = prec == PREC_F16C && ylo != nullptr;
    if (logn == 14 && !(flags & CONV_ONESHOT) && !(stamp && !gated) && kf_packed) {
        kf = kf_packed;
        const int xcd = !(flags & CONV_NO_XCD);
        if (prec == PREC_F32) {     // round 5: the exact / fp16x3 engine's rows (raw x0 | x1 | v, fp32) through the persistent form too
            if (ids) launch_conv_pers_inst<float, true>(nullptr, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, xcd, st);
            else launch_conv_pers_inst<float, false>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st);
        } else if (prec == PREC_BF16) {
            if (ids) launch_conv_pers_inst<bf16_t, true>(nullptr, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, xcd, st);
            else if (gated) launch_conv_pers_inst<bf16_t, false, true>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st);
            else launch_conv_pers_inst<bf16_t, false>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st);
        } else if (lo) {
            if (ids) launch_conv_pers_inst<f16_t, true, false, true>(nullptr, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, xcd, st, ylo);
            else if (gated) launch_conv_pers_inst<f16_t, false, true, true>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st, ylo);
            else launch_conv_pers_inst<f16_t, false, false, true>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st, ylo);
        } else {
            if (ids) launch_conv_pers_inst<f16_t, true>(nullptr, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, xcd, st);
            else if (gated) launch_conv_pers_inst<f16_t, false, true>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st);
            else launch_conv_pers_inst<f16_t, false>(z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, nullptr, nullptr, xcd, st);
        }
        return;
    }
#define CLM_CONV_CASE(n) \
    case n: launch_conv_p<n>(prec, z, y, kf, tw, ktime, short_w, short_b, B, L, Lp, ids8, ztab, st, gated, ylo); break;
    switch (logn) {
        CLM_CONV_CASE(8)
        CLM_CONV_CASE(9)
        CLM_CONV_CASE(10)
        CLM_CONV_CASE(11)
        CLM_CONV_CASE(12)
        CLM_CONV_CASE(13)
        CLM_CONV_CASE(14)
        default: break;
    }
#undef CLM_CONV_CASE
}

}  // namespace clm
