// gemm_common.h -- shared pieces of the MFMA GEMM kernels (gemm.hip: generic/fp32 + score; gemm16.hip: the tuned
// 16-bit kernels).  See gemm.hip for the design notes and the MFMA register maps.
#pragma once
#include "clm_common.h"

namespace clm {

template <int PREC>
struct CT;
template <>
struct CT<PREC_F32> {
    using elem = float;
    using frag = float;
    static constexpr int MFMA_K = 2, BM = 64, RS = 257;
};
template <>
struct CT<PREC_BF16> {
    using elem = bf16_t;
    using frag = u16x8;
    static constexpr int MFMA_K = 16, BM = 128, RS = 264;
};
template <>
struct CT<PREC_F16> {
    using elem = f16_t;
    using frag = u16x8;
    static constexpr int MFMA_K = 16, BM = 128, RS = 264;
};

template <>
struct CT<PREC_F16C> {
    using elem = f16_t;
    using frag = u16x8;
    static constexpr int MFMA_K = 16, BM = 128, RS = 264;
};
// fragment slots per k-step in the packed stream: 1, or 2 in the compensated mode (per group of four k-steps: four hi
// fragments, the lo bytes of the K = 64 fp8 MFMA in two slots, two slots unused -- pack_weight_split_kernel)
template <int PREC>
constexpr int WFR = PREC == PREC_F16C ? 2 : 1;
// lo half of the compensated mode: e4m3((w - hi) * 2^17), undone by the MFMA's block scale 2^-17 (E8M0 byte 127 - 17)
constexpr float LO8_SCALE = 131072.f;
constexpr int LO8_E8M0 = 127 - 17;
typedef int i32x8 __attribute__((ext_vector_type(8)));
// acc += w_lo . a: `w8` the lane's 32 e4m3 bytes of the weight tile, `a8` its 32 e5m2 bytes of the activation tile (the same 64
// k in the same order); W_IS_A: the weights are the MFMA's A operand (accumulator rows = output features)
template <bool W_IS_A>
__device__ __forceinline__ f32x16 mfma_lo8(i32x8 w8, i32x8 a8, f32x16 c) {
    if (W_IS_A) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8, c, 0 /*e4m3*/, 1 /*e5m2*/, 0, LO8_E8M0, 0, 127);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, c, 1, 0, 0, 127, 0, LO8_E8M0);
}
// the 8 halfs of an activation fragment -> 8 e5m2 bytes (two registers of the MFMA's 8-register operand) BY TRUNCATION: the upper
// byte of an IEEE half (sign, 5 exponent bits, 2 mantissa bits) IS an e5m2 number, so the "conversion" is a byte gather -- two
// v_perm_b32 per fragment instead of the four v_cvt_scalef32_pk_bf8_f16 of round 2, with no dependence on the operand register's
// previous contents.  The conversions were a third of the tail kernel's VALU instructions (every wave converts the whole
// activation tile for itself): same box, tail kernel per step 21.3 -> 20.45 ms (round 3, tools/build_variant.sh A/B).
// Truncation is biased towards zero: the mean of trunc(a) / a over log-uniform mantissas is (1/5 + 1/6 + 1/7 + 1/8) / ln 2 =
// 0.9154 (0.9158 measured on N(0,1) halfs) -- folded into the packed lo weights (LO8_TRUNC_GAIN, pack_weight_split_kernel), after
// which the relative error of the operand is 5.5 % rms, that of round-to-nearest e5m2 5.3 %: token-random noise on a term that
// enters at 2^-11.  Infinities / NaN keep their meaning; fp16 subnormals become e5m2 subnormals.
// (Also measured and not kept, round 3: the lo product in the 4/6-bit class of the scaled MFMA -- e2m1 activations x e2m3
//  weights with a block scale run it in 32 cycles instead of 64, tools/micro/mfma_fp6_lo.cpp -- tail 22.8 -> 22.2 ms per step
//  against 20.45 here: the four conversions per fragment, not the MFMA cycles, were what the lo half cost.)
constexpr float LO8_TRUNC_GAIN = LO8_TRUNC_GAIN_V;
__device__ __forceinline__ void frag_to_e5m2t(u16x8 af, int& w0, int& w1) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = __builtin_bit_cast(u32x4, af);
    const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];   // (scalars first: bit_cast of a vector ELEMENT reads element 0, hipcc 7.2)
    w0 = (int)__builtin_amdgcn_perm(d1, d0, 0x07050301u);         // bytes 1, 3 of d0 then of d1: the upper bytes of halfs 0..3
    w1 = (int)__builtin_amdgcn_perm(d3, d2, 0x07050301u);
}

// ---- round 4: the lo half of an ACTIVATION operand ------------------------------------------------------------------------------
// With the weights compensated, what is left of the fp16c mode's error is the fp16 rounding of the GEMM activation operands (and of
// z / y in HBM): token-random, 2^-12 relative per element (tests/error_model.py).  The producer of an operand
// tile (LayerNorm from the accumulators; the convolution for y) has the fp32 value in registers, so it also leaves
//     lo8 = e5m2((x - fp16(x)) * 2^10 / 0.9155)          one byte per element, in a second tile
// and the product gets a third term per 64-deep group:  acc += lo8(x) . e5m2_trunc(w_hi)  on the same K = 64 block-scaled MFMA
// (block scale 2^-10).  The weight operand of that term needs no storage and no load: the upper bytes of the fp16 hi fragments a
// wave already holds ARE e5m2 numbers (frag_to_e5m2t: two v_perm per fragment, once per weight set); their truncation bias (0.9155,
// see LO8_TRUNC_GAIN) is folded into the producer's scale.  2^10: x - fp16(x) is at most 2^-11 |x|, and |x| < 65504, so the scaled
// lo never exceeds 16384 < 57344 = the e5m2 maximum -- no saturation logic; fp16 values down to 2^-4 keep a NORMAL lo (relative
// precision 2^-3 on a 2^-12 term), smaller ones lose it to e5m2's subnormals, where the absolute error is below 2^-27.
// (LO2_SCALE, lo8_pack4, lo8_unpack4: clm_common.h -- the convolution kernels write y's and read z's lo bytes with them.)
// acc += lo8(activations) . e5m2_trunc(w_hi): `w8h` the lane's 32 truncated weight bytes (frag_to_e5m2t of the set's four hi
// fragments), `alo` its 32 activation lo bytes, both in the k order of the fp16 fragments (byte 8 s + j = k-step s, element j)
template <bool W_IS_A>
__device__ __forceinline__ f32x16 mfma_lo2(i32x8 w8h, i32x8 alo, f32x16 c) {
    if (W_IS_A) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8h, alo, c, 1 /*e5m2*/, 1 /*e5m2*/, 0, 127, 0, LO2_E8M0);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(alo, w8h, c, 1, 1, 0, LO2_E8M0, 0, 127);
}

template <int PREC>
__device__ __forceinline__ f32x16 mfma(typename CT<PREC>::frag a, typename CT<PREC>::frag b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma<PREC_F32>(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma<PREC_BF16>(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma<PREC_F16>(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
}

template <>
__device__ __forceinline__ f32x16 mfma<PREC_F16C>(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
}

// ---------------------------------------------------------------------------------------- the kernel
enum { A_LN = 0, A_TM = 1, A_CM = 2 };
enum { E_CM = 0, E_GELU_TM = 1, E_RESID = 2, E_SCORE = 3 };

struct GemmArgs {
    const float* h_in;    // A_LN source [B, L, 256] fp32
    const void* a_in;     // A_TM: [B, L, K] ; A_CM: [B, 256, Lp]   (compute dtype)
    const float* ln_g;
    const float* ln_b;
    const void* w;        // packed
    const float* bias;    // [N]
    void* out;            // E_CM: z [B, N, Lp] ; E_GELU_TM: u [B, L, N]
    float* h_out;         // E_RESID: residual stream [B, L, 256], updated in place
    const float* w2;      // E_SCORE: attention.2.weight [256]
    const float* b2;      // E_SCORE: attention.2.bias [1]
    float* scores;        // E_SCORE: [B, L]
    int B, L, Lp;
    float eps;
};

template <typename E>
__device__ __forceinline__ void store4(E* dst, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* dst, float a, float b, float c, float d) {
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, float a, float b, float c, float d) {
    u16x4 p = {from_float<bf16_t>(a).bits, from_float<bf16_t>(b).bits, from_float<bf16_t>(c).bits,
               from_float<bf16_t>(d).bits};
    *reinterpret_cast<u16x4*>(dst) = p;
}
template <>
__device__ __forceinline__ void store4<f16_t>(f16_t* dst, float a, float b, float c, float d) {
    u16x4 p = {from_float<f16_t>(a).bits, from_float<f16_t>(b).bits, from_float<f16_t>(c).bits,
               from_float<f16_t>(d).bits};
    *reinterpret_cast<u16x4*>(dst) = p;
}

// ---- A-tile staging: BM tokens x 256 reduction columns of chunk kc into LDS, in the compute dtype ----------
template <int PREC, int ASRC, int K, int NW = 4>
__device__ __forceinline__ void stage_a_tile(const GemmArgs& a, typename CT<PREC>::elem* As, int b, int t0, int kc) {
    using C = CT<PREC>;
    using elem = typename C::elem;
    constexpr int BM = C::BM, RS = C::RS, KC = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, L = a.L, Lp = a.Lp;
    if (ASRC == A_LN) {
        // All row loads of a batch are issued before the first reduction: the staging phase is HBM-latency bound
        // (every wave of the workgroup is in it at once), so 16 rows in flight per wave instead of 4 cut it ~4x.
        const float4 g4 = *reinterpret_cast<const float4*>(a.ln_g + lane * 4);
        const float4 b4 = *reinterpret_cast<const float4*>(a.ln_b + lane * 4);
        constexpr int ROWS = BM / NW, BATCH = ROWS < 16 ? ROWS : 16;
        static_assert(ROWS % BATCH == 0, "rows per wave must be a multiple of the batch");
#pragma unroll 1
        for (int r0 = 0; r0 < ROWS; r0 += BATCH) {
            float4 x[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int t = t0 + wave + (r0 + i) * NW;      // clamped, branch-free: all loads of the batch stay in flight
                x[i] = *reinterpret_cast<const float4*>(a.h_in + ((size_t)b * L + (t < L ? t : L - 1)) * D + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int r = wave + (r0 + i) * NW, t = t0 + r;
                float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
                if (t < L) {  // wave-uniform
                    float mean = wave_sum((x[i].x + x[i].y) + (x[i].z + x[i].w)) * (1.0f / D);
                    float d0 = x[i].x - mean, d1 = x[i].y - mean, d2 = x[i].z - mean, d3 = x[i].w - mean;
                    float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
                    float rstd = 1.0f / sqrtf(var + a.eps);
                    y0 = d0 * rstd * g4.x + b4.x;
                    y1 = d1 * rstd * g4.y + b4.y;
                    y2 = d2 * rstd * g4.z + b4.z;
                    y3 = d3 * rstd * g4.w + b4.w;
                }
                store4<elem>(As + r * RS + lane * 4, y0, y1, y2, y3);
            }
        }
    } else if (ASRC == A_TM) {
        const elem* src = reinterpret_cast<const elem*>(a.a_in);
        if (PREC == PREC_F32) {
            for (int r = wave; r < BM; r += NW) {
                int t = t0 + r;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < L)
                    x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) +
                                                         ((size_t)b * L + t) * K + kc * KC + lane * 4);
                float* d = reinterpret_cast<float*>(As) + r * RS + lane * 4;
                d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
            }
        } else {
#pragma unroll 4
            for (int r = tid >> 5; r < BM; r += 2 * NW) {
                int t = t0 + r, c8 = (tid & 31) * 8;
                uint4 x = make_uint4(0, 0, 0, 0);
                if (t < L) x = *reinterpret_cast<const uint4*>(src + ((size_t)b * L + t) * K + kc * KC + c8);
                *reinterpret_cast<uint4*>(As + r * RS + c8) = x;
            }
        }
    } else {  // A_CM: y [B, 256, Lp] channel-major -> LDS [token][channel]   (scalar LDS scatter)
        const elem* src = reinterpret_cast<const elem*>(a.a_in);
        constexpr int TPV = (PREC == PREC_F32) ? 4 : 8;   // tokens per 16-byte vector
        static_assert(BM / TPV == 16, "tile is 256 B per channel");
#pragma unroll 2
        for (int c = tid >> 4; c < KC; c += 4 * NW) {
            int tk = (tid & 15) * TPV;
            elem v[TPV];
            if (t0 + tk < Lp) {
                uint4 x = *reinterpret_cast<const uint4*>(src + ((size_t)b * D + c) * Lp + t0 + tk);
                __builtin_memcpy(v, &x, 16);
            } else {
#pragma unroll
                for (int e = 0; e < TPV; ++e) v[e] = from_float<elem>(0.f);
            }
#pragma unroll
            for (int e = 0; e < TPV; ++e) As[(tk + e) * RS + c] = v[e];
        }
    }
}

// ---- epilogue of one 256-wide output block --------------------------------------------------------------
template <int PREC, int EPI, int N, bool MASK>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x16 (&acc)[CT<PREC>::BM / 32][2], int b, int t0,
                                         int nb, unsigned char* smem) {
    using C = CT<PREC>;
    using elem = typename C::elem;
    constexpr int BM = C::BM, MT = BM / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, L = a.L, Lp = a.Lp;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nbase = nb * 256 + wave * 64;
    // Addresses are ONE per-lane base pointer plus compile-time offsets: anything fancier gets hoisted out of the
    // block loop by LICM and its 64-bit address registers then live (and spill) across the MFMA phases.
    if (EPI == E_CM) {
        // acc rows = output channel (nbase + nt*32 + rowoff(r) + 4*lhalf), cols = token (t0 + mt*32 + lrow)
        elem* zb = reinterpret_cast<elem*>(a.out) + ((size_t)b * N + nbase + 4 * lhalf) * Lp + t0 + lrow;
        const float* bb = a.bias + nbase + 4 * lhalf;
        const int tl = t0 + lrow;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = nt * 32 + (r & 3) + 8 * (r >> 2);
                const float bias = bb[nrow];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (!MASK || tl + mt * 32 < L)
                        zb[(size_t)nrow * Lp + mt * 32] = from_float<elem>(acc[mt][nt][r] + bias);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (EPI == E_GELU_TM) {
        // acc rows = token (t0 + mt*32 + rowoff(r) + 4*lhalf), cols = hidden unit (nbase + nt*32 + lrow)
        elem* ub = reinterpret_cast<elem*>(a.out) + ((size_t)b * L + t0 + 4 * lhalf) * N + nbase + lrow;
        const int tl = t0 + 4 * lhalf;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bias = a.bias[nbase + nt * 32 + lrow];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int trow = mt * 32 + (r & 3) + 8 * (r >> 2);
                    if (!MASK || tl + trow < L)
                        ub[(size_t)trow * N + nt * 32] = from_float<elem>(gelu_tanh(acc[mt][nt][r] + bias));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (EPI == E_RESID) {
        float* hb = a.h_out + ((size_t)b * L + t0 + 4 * lhalf) * D + nbase + lrow;
        const int tl = t0 + 4 * lhalf;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bias = a.bias[nbase + nt * 32 + lrow];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int trow = mt * 32 + (r & 3) + 8 * (r >> 2);
                    if (!MASK || tl + trow < L) {
                        float* p = hb + (size_t)trow * D + nt * 32;
                        *p = *p + (acc[mt][nt][r] + bias);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // at most 16 residual loads in flight per group
            }
        }
    } else {  // E_SCORE: s[t] = sum_n w2[n] * gelu_erf(acc[t][n] + b1[n]) + b2, deterministic reduction
        __syncthreads();  // every wave is done reading the A tile; reuse LDS for the partials
        float* part = reinterpret_cast<float*>(smem);  // [4][BM]
        float w2v[2], b1v[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int n = nbase + nt * 32 + lrow;
            w2v[nt] = a.w2[n];
            b1v[nt] = a.bias[n];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = gelu_erf(acc[mt][0][r] + b1v[0]) * w2v[0] + gelu_erf(acc[mt][1][r] + b1v[1]) * w2v[1];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (lrow == 0) part[wave * BM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf] = s;
            }
        __syncthreads();
        if (tid < BM && t0 + tid < L)
            a.scores[(size_t)b * L + t0 + tid] =
                ((part[tid] + part[BM + tid]) + (part[2 * BM + tid] + part[3 * BM + tid])) + a.b2[0];
    }
}

// one "set" of weight fragments: both column tiles of a wave x SETK fragments (16 bytes per lane each) = SETK k-steps,
// or SETK / 2 k-steps of (hi, lo) pairs in the compensated mode; a 256-deep reduction chunk is NPARTS sets
constexpr int SETK = 8;
// the arithmetic of the two MLP products inside a kernel of mode PREC (see tail16_kernel): plain fp16 under fp16c
template <int PREC>
constexpr int MLP_PREC = PREC == PREC_F16C ? (int)PREC_F16 : PREC;
template <int PREC>
constexpr int KPS = SETK / WFR<PREC>;                       // k-steps per set
template <int PREC>
constexpr int NPARTS = (256 / 16) / KPS<PREC>;              // sets per 256-deep chunk (16-bit types): 2, or 4
template <int PREC, int K, int NT = 2>
__device__ __forceinline__ void load_set(const typename CT<PREC>::frag* wp, int nb, int kc, int part, int wave, int lane,
                                         typename CT<PREC>::frag (&dst)[NT][SETK]) {
    constexpr int KSTEPS = 256 / CT<PREC>::MFMA_K, KSTEPS_ALL = K / CT<PREC>::MFMA_K;
    constexpr int FR = CT<PREC>::MFMA_K == 16 ? WFR<PREC> : 1, KP = SETK / FR;
    if constexpr (lab::NOW) {
        if (kc != 0 || part != 0) return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const typename CT<PREC>::frag* p =
            wp + ((size_t)(nb * 8 + wave * NT + nt) * KSTEPS_ALL + kc * KSTEPS + part * KP) * (FR * 64) + lane;
        if constexpr (lab::W0) p = wp + (size_t)(wave * NT + nt) * KSTEPS_ALL * (FR * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < SETK; ++ks)
            if (PREC != PREC_F16C || ks < 6) dst[nt][ks] = p[(size_t)ks * 64];     // compensated mode: slots 6, 7 are unused
    }
}
template <int PREC, int EPI>
__device__ __forceinline__ void compute_set(const typename CT<PREC>::elem* As, int part, int lrow, int lhalf,
                                            const typename CT<PREC>::frag (&src)[2][SETK],
                                            f32x16 (&acc)[CT<PREC>::BM / 32][2]) {
    using C = CT<PREC>;
    using frag = typename C::frag;
    constexpr int MT = C::BM / 32, RS = C::RS;
#pragma unroll
    for (int ks = 0; ks < SETK; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            frag af = *reinterpret_cast<const frag*>(As + (mt * 32 + lrow) * RS + (part * SETK + ks) * 16 + lhalf * 8);
            if (EPI == E_CM) {
                acc[mt][0] = mfma<PREC>(src[0][ks], af, acc[mt][0]);
                acc[mt][1] = mfma<PREC>(src[1][ks], af, acc[mt][1]);
            } else {
                acc[mt][0] = mfma<PREC>(af, src[0][ks], acc[mt][0]);
                acc[mt][1] = mfma<PREC>(af, src[1][ks], acc[mt][1]);
            }
        }
    }
}

}  // namespace clm
