// tf_model.hip -- SequenceCNNTransformer forward on MI355X: kernels + engine + C ABI.
//
// Reference: /root/reference/chimeralm/models/components/transformer.py
//   :28-86  modules   Embedding(12, 256, padding_idx 4) ; 3 x [Conv1d(256, 256, k=3, padding=1), ReLU, MaxPool1d(2, 2)] ;
//                     SinusoidalPositionalEncoding ; LayerNorm ; TransformerEncoder(12 x post-norm layer: 8 heads of 32,
//                     feed-forward 1024, ReLU) ; Linear(256, 1) softmax pooling ; Linear(256, 128), ReLU, Linear(128, 2)
//   :88-104 forward   (no masks anywhere: pads are ordinary tokens)
// configured by /root/reference/configs/model/transformer.yaml:3-12.  Oracle: oracle/transformer_oracle.py (pinned against the
// reference module itself, tests/golden/transformer_golden.npz).
//
// First correct version: every stage is its own kernel (the Hyena path's fusion lessons are not applied yet); the dense work is
// on MFMA with the same LDS-resident 128-token activation tile / register-resident weight half-set scheme as gemm16.hip, the
// attention is attention.hip.  Data layout, one batch of B reads (L tokens -> L3 = L / 8 positions, M = B * L3 rows):
//   x1 [B, L/2, 256], x2 [B, L/4, 256], x3 [B, L3, 256]     16-bit, token-major          conv stack
//   h  [M, 256] fp32 (residual stream, post-norm: always a LayerNorm output)   hx [M, 256] 16-bit copy = next GEMM operand
//   qkv [M, 768], att [M, 256], u [M, 1024]                  16-bit
// The convolution is a GEMM with K = 3 * 256: one 130-row tile of the input (1 halo row each side) feeds three passes with the
// row offset 0 / 1 / 2 and the weight slice W[:, :, dk]; ReLU and the max-pool happen on the accumulator quads (4 consecutive
// positions of one channel) before anything is written.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "chimeralm_hip.h"
#include "gemm16_common.h"

namespace clm {
namespace tf {

constexpr int TFF = 1024, TQKV = 768, TVOC = 12, TCH = 128 /* classifier hidden */;
constexpr int ZRS = 40;   // row stride (elements) of the wave-private [token][32 features] output staging tiles: 80 bytes

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void conv_w_split_kernel(const float* __restrict__ w, float* __restrict__ out) {   // [co][ci][3] -> [3][co][ci]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * D * 3) return;
    const int dk = i % 3, ci = (i / 3) % D, co = i / (3 * D);
    out[((size_t)dk * D + co) * D + ci] = w[i];
}

// wave-private [128 tokens][32 features] 16-bit tile -> global rows (64 bytes per token and wave), 16 rows per instruction
template <typename E>
__device__ __forceinline__ void store_wave_tile(const E* zs, E* out, size_t row0, size_t nrows_valid, int ld, int col0, int lane,
                                                int ntok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 16 + (lane >> 2), piece = lane & 3;
        if (row < ntok && (size_t)row < nrows_valid) {
            const uint4 v = *reinterpret_cast<const uint4*>(zs + row * ZRS + piece * 8);
            *reinterpret_cast<uint4*>(out + (row0 + row) * ld + col0 + piece * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv + ReLU + max-pool
template <int PREC, bool FROM_IDS>
__global__ __launch_bounds__(512) void conv3_relu_pool_kernel(const unsigned char* __restrict__ ids8, int ids_stride,
                                                              const float* __restrict__ emb,
                                                              const typename CT<PREC>::elem* __restrict__ xin,
                                                              const void* __restrict__ wpk, const float* __restrict__ bias,
                                                              typename CT<PREC>::elem* __restrict__ xout, int Lin, int Lout) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);                 // [130][RS16]: input rows t0 - 1 .. t0 + 128
    elem* Zs = As + 130 * RS16;                               // 8 x [64 pooled positions][ZRS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * 128;
    const frag* wp = reinterpret_cast<const frag*>(wpk);
    constexpr size_t WSLICE = (size_t)D * D / 8 * WFR<PREC>;  // fragments per [256 x 256] slice
    f32x16 acc[4];
    frag bs[2][1][SETK];
    load_set<PREC, D, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
    __builtin_amdgcn_sched_barrier(0);
    {
        const int piece = tid & 31;
#pragma unroll 1
        for (int r = tid >> 5; r < 130; r += 16) {
            const int tg = t0 - 1 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tg >= 0 && tg < Lin) {
                if (FROM_IDS) {
                    const int id = ids8[(size_t)b * ids_stride + tg];
                    const float* e = emb + (size_t)(id < TVOC ? id : TVOC - 1) * D + piece * 8;
                    const float4 a = *reinterpret_cast<const float4*>(e), c4 = *reinterpret_cast<const float4*>(e + 4);
                    u16x8 pk = {to_bits<PREC>(a.x), to_bits<PREC>(a.y), to_bits<PREC>(a.z), to_bits<PREC>(a.w),
                                to_bits<PREC>(c4.x), to_bits<PREC>(c4.y), to_bits<PREC>(c4.z), to_bits<PREC>(c4.w)};
                    v = __builtin_bit_cast(uint4, pk);
                } else {
                    v = *reinterpret_cast<const uint4*>(xin + ((size_t)b * Lin + tg) * D + piece * 8);
                }
            }
            *reinterpret_cast<uint4*>(As + r * RS16 + piece * 8) = v;
        }
    }
    __syncthreads();
    zero_acc(acc);
#pragma unroll
    for (int dk = 0; dk < 3; ++dk)                            // y[t] += W[:, :, dk] x[t + dk - 1]
        phase_tm<PREC, D, D, false>(As + dk * RS16, wp + dk * WSLICE, 0, 0, wp + (dk < 2 ? dk + 1 : 0) * WSLICE, 0, 0, wave, lane, bs,
                                    acc);
    // rows = positions (register quads = 4 consecutive positions), lane = output channel wave*32 + lrow
    elem* zs = Zs + wave * 64 * ZRS;
    const float bv = bias[wave * 32 + lrow];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v0 = fmaxf(acc[mt][4 * g + 0] + bv, 0.f), v1 = fmaxf(acc[mt][4 * g + 1] + bv, 0.f);
            const float v2 = fmaxf(acc[mt][4 * g + 2] + bv, 0.f), v3 = fmaxf(acc[mt][4 * g + 3] + bv, 0.f);
            const int pl = 16 * mt + 4 * g + 2 * lhalf;       // pooled position inside the tile
            zs[pl * ZRS + lrow] = from_float<elem>(fmaxf(v0, v1));
            zs[(pl + 1) * ZRS + lrow] = from_float<elem>(fmaxf(v2, v3));
        }
    const size_t p0 = (size_t)(t0 / 2);
    const size_t valid = (size_t)Lout > p0 ? (size_t)Lout - p0 : 0;
    store_wave_tile<elem>(zs, xout + (size_t)b * Lout * D, p0, valid, D, wave * 32, lane, 64);
}

// ------------------------------------------------------------------------------------------------ + PE, LayerNorm
template <int PREC>
__global__ __launch_bounds__(256) void pe_ln_kernel(const typename CT<PREC>::elem* __restrict__ x, const float* __restrict__ pe,
                                                    const float* __restrict__ g, const float* __restrict__ bta,
                                                    float* __restrict__ h, typename CT<PREC>::elem* __restrict__ hx,
                                                    size_t M, int L3, float eps) {
    using elem = typename CT<PREC>::elem;
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int t = (int)(row % (size_t)L3);
    const u16x4 raw = *reinterpret_cast<const u16x4*>(x + row * D + lane * 4);
    const float4 p = *reinterpret_cast<const float4*>(pe + (size_t)t * D + lane * 4);
    elem e0, e1, e2, e3;
    e0.bits = raw[0]; e1.bits = raw[1]; e2.bits = raw[2]; e3.bits = raw[3];
    const float x0 = to_float(e0) + p.x, x1 = to_float(e1) + p.y, x2 = to_float(e2) + p.z, x3 = to_float(e3) + p.w;
    const float mean = wave_sum((x0 + x1) + (x2 + x3)) * (1.0f / D);
    const float d0 = x0 - mean, d1 = x1 - mean, d2 = x2 - mean, d3 = x3 - mean;
    const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 g4 = *reinterpret_cast<const float4*>(g + lane * 4), b4 = *reinterpret_cast<const float4*>(bta + lane * 4);
    const float y0 = d0 * rstd * g4.x + b4.x, y1 = d1 * rstd * g4.y + b4.y, y2 = d2 * rstd * g4.z + b4.z,
                y3 = d3 * rstd * g4.w + b4.w;
    *reinterpret_cast<float4*>(h + row * D + lane * 4) = make_float4(y0, y1, y2, y3);
    store4<elem>(hx + row * D + lane * 4, y0, y1, y2, y3);
}

// ------------------------------------------------------------------------------------------------ dense layers
struct LinArgs {
    const void* a;            // [M, K] 16-bit, token-major
    const void* w;            // packed [N, K]
    const float* bias;        // [N]
    void* out16;              // E_ACT: [M, N] 16-bit
    float* h;                 // E_RES_LN: residual in, LayerNorm(residual + a W^T + bias) out, [M, 256] fp32
    void* hx;                 // E_RES_LN: 16-bit copy of the new h
    const float *ln_g, *ln_b;
    size_t M;
    float eps;
    int relu;
};
enum { E_ACT = 0, E_RES_LN = 1 };

template <int PREC>
__device__ __forceinline__ void stage_rows16(const typename CT<PREC>::elem* a, size_t row0, size_t M, int K, int kc,
                                             typename CT<PREC>::elem* As, int tid) {
    const int piece = tid & 31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (tid >> 5) + 16 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < M) v = *reinterpret_cast<const uint4*>(a + (row0 + r) * K + kc * 256 + piece * 8);
        *reinterpret_cast<uint4*>(As + r * RS16 + piece * 8) = v;
    }
}

// acc (rows = this wave's 32 features, lane = token) + bias + residual h -> post-norm LayerNorm -> h (fp32, whole lines through
// the XOR-swizzled LDS transpose) and hx (16-bit, rows of the normalised tile as they lie in LDS).  `smem`: >= 128 KiB staging
// area starting at the tile As; P1 / P2 behind it.
template <int PREC>
__device__ __forceinline__ int res_ln(f32x16 (&acc)[4], const float* __restrict__ bias, const float* __restrict__ h,
                                      const float* ln_g, const float* ln_b, float eps, size_t row0, size_t M,
                                      typename CT<PREC>::elem* As, float* P1, float* P2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lhalf = lane >> 5;
    // r = acc + bias + residual (this lane's token, 4 consecutive features per quad)
    const float* bp = bias + wave * 32 + 4 * lhalf;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const size_t row = row0 + mt * 32 + lrow;
        const float* hr = (h ? h : bias) + (h && row < M ? row : 0) * D + wave * 32 + 4 * lhalf;   // h == nullptr: no residual here
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bp + 8 * q);
            float4 hv = *reinterpret_cast<const float4*>(hr + 8 * q);
            if (!h) hv = make_float4(0.f, 0.f, 0.f, 0.f);
            acc[mt][4 * q + 0] += bb.x + hv.x;
            acc[mt][4 * q + 1] += bb.y + hv.y;
            acc[mt][4 * q + 2] += bb.z + hv.z;
            acc[mt][4 * q + 3] += bb.w + hv.w;
        }
    }
    const int l_valid = (int)std::min<size_t>(M > row0 ? M - row0 : 0, 128);
    ln_acc_to_tile<PREC, true>(acc, P1, P2, ln_g, ln_b, eps, As, 0, l_valid, wave, lrow, lhalf);
    return l_valid;
}

// the normalised tile (As, 16-bit) -> hx rows; the normalised accumulators -> h (fp32, whole 128-byte lines through the
// XOR-swizzled LDS transpose, which overwrites the first 128 KiB of `smem`, As included)
template <int PREC>
__device__ __forceinline__ void store_h_hx(const f32x16 (&acc)[4], float* __restrict__ h, typename CT<PREC>::elem* __restrict__ hx,
                                           size_t row0, int l_valid, unsigned char* smem, const typename CT<PREC>::elem* As) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    {
        const int piece = tid & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (tid >> 5) + 16 * i;
            if (r < l_valid)
                *reinterpret_cast<uint4*>(hx + (row0 + r) * D + piece * 8) = *reinterpret_cast<const uint4*>(As + r * RS16 + piece * 8);
        }
    }
    __syncthreads();
    float* rs = reinterpret_cast<float*>(smem) + wave * (128 * 32);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int tok = mt * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int chunk = (2 * q + lhalf) ^ (tok & 7);
            *reinterpret_cast<float4*>(rs + tok * 32 + 4 * chunk) =
                make_float4(acc[mt][4 * q + 0], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
        }
    }
    const int c = lane & 7, rsub = lane >> 3;
    float* hrow = h + row0 * D + wave * 32 + 4 * c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tr = i * 8 + rsub;
        const float4 v = *reinterpret_cast<const float4*>(rs + tr * 32 + 4 * (c ^ (tr & 7)));
        if (tr < l_valid) *reinterpret_cast<float4*>(hrow + (size_t)tr * D) = v;
    }
}

// N / 256 blocks of a bias (+ReLU) linear layer over the staged tile As, 16-bit token-major output through the wave-private
// tiles at `zs`.  On entry bs[0] holds the first half-set of block 0.
template <int PREC, int N>
__device__ __forceinline__ void act_blocks(const typename CT<PREC>::elem* As, typename CT<PREC>::elem* zs, const u16x8* wp,
                                           const float* __restrict__ bias, typename CT<PREC>::elem* out, size_t row0, size_t M,
                                           bool relu, u16x8 (&bs)[2][1][SETK], f32x16 (&acc)[4]) {
    using elem = typename CT<PREC>::elem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll 1
    for (int nb = 0; nb < N / 256; ++nb) {
        zero_acc(acc);
        phase_tm<PREC, D, D, true>(As, wp, nb, 0, wp, nb + 1 < N / 256 ? nb + 1 : 0, 0, wave, lane, bs, acc);
        // rows = features (register quads), lane = token
        const float* bp = bias + nb * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bp + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float v0 = acc[mt][4 * q + 0] + bb.x, v1 = acc[mt][4 * q + 1] + bb.y, v2 = acc[mt][4 * q + 2] + bb.z,
                      v3 = acc[mt][4 * q + 3] + bb.w;
                if (relu) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f), v2 = fmaxf(v2, 0.f), v3 = fmaxf(v3, 0.f);
                u16x4 pk = {to_bits<PREC>(v0), to_bits<PREC>(v1), to_bits<PREC>(v2), to_bits<PREC>(v3)};
                *reinterpret_cast<u16x4*>(zs + (mt * 32 + lrow) * ZRS + 8 * q + 4 * lhalf) = pk;
            }
        }
        const size_t valid = M > row0 ? M - row0 : 0;
        store_wave_tile<elem>(zs, out, row0, valid, N, nb * 256 + wave * 32, lane, 128);
    }
}

template <int PREC, int EPI, int K, int N>
__global__ __launch_bounds__(512) void linear16_kernel(LinArgs m) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    static_assert(K % 256 == 0 && N % 256 == 0, "256-wide chunks");
    static_assert(EPI == E_ACT ? K == 256 : N == 256, "E_ACT keeps one activation tile; E_RES_LN normalises one 256-wide row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const size_t row0 = (size_t)blockIdx.x * 128;
    const elem* a = reinterpret_cast<const elem*>(m.a);
    const frag* wp = reinterpret_cast<const frag*>(m.w);
    f32x16 acc[4];
    frag bs[2][1][SETK];
    if constexpr (EPI == E_ACT) {
        elem* zs = As + 128 * RS16 + wave * 128 * ZRS;        // wave-private [128 tokens][32 features]
        load_set<PREC, K, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
        __builtin_amdgcn_sched_barrier(0);
        stage_rows16<PREC>(a, row0, m.M, K, 0, As, tid);
        __syncthreads();
        act_blocks<PREC, N>(As, zs, wp, m.bias, reinterpret_cast<elem*>(m.out16), row0, m.M, m.relu != 0, bs, acc);
    } else {
        float* P1 = reinterpret_cast<float*>(smem + (size_t)2 * 128 * RS16 * 2);   // tables behind the 128 KiB staging area
        float* P2 = P1 + 16 * 128;
        zero_acc(acc);
        load_set<PREC, K, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
#pragma unroll 1
        for (int kc = 0; kc < K / 256; ++kc) {                 // (bs[0] holds the chunk's first set: requested a chunk ahead)
            load_set<PREC, K, 1>(wp, 0, kc, 1, wave, lane, bs[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (kc > 0) __syncthreads();                       // every wave is done with the previous chunk of the tile
            stage_rows16<PREC>(a, row0, m.M, K, kc, As, tid);
            __syncthreads();
            phase_tm<PREC, K, K, true, true>(As, wp, 0, kc, wp, 0, kc + 1 < K / 256 ? kc + 1 : 0, wave, lane, bs, acc);
        }
        const int l_valid = res_ln<PREC>(acc, m.bias, m.h, m.ln_g, m.ln_b, m.eps, row0, m.M, As, P1, P2);
        store_h_hx<PREC>(acc, m.h, reinterpret_cast<elem*>(m.hx), row0, l_valid, smem, As);
    }
}

// ------------------------------------------------------------------------------------------------ fused feed-forward
// h = LN2(h + relu(hx W1^T + b1) W2^T + b2): the 1024-wide activations never reach HBM -- per 256-wide hidden chunk the fc1
// accumulator goes through bias + ReLU into an LDS tile that is the A operand of fc2's reduction chunk (same scheme as mlp16 in
// gemm16.hip); the post-norm LayerNorm runs on the fc2 accumulators and the new h / hx leave as whole lines.
struct FfnArgs {
    const void* hx_in;        // [M, 256] 16-bit
    const void *w1, *w2;      // packed [1024, 256], [256, 1024]
    const float *b1, *b2, *ln_g, *ln_b;
    float* h;                 // [M, 256] fp32, residual in / LayerNorm out
    void* hx_out;             // [M, 256] 16-bit copy of the new h (may alias hx_in: a tile is read completely before it is written)
    size_t M;
    float eps;
    const void* w_qkv;        // next layer's packed in_proj (null after the last layer): QKV of the tile just normalised
    const float* b_qkv;
    void* qkv;                // [M, 768] 16-bit
    // whole-layer form (att != null): the kernel starts from the attention output -- x1 = LN1(h + att W_o^T + b_o) stays in the
    // fc2 accumulators as the feed-forward residual and in LDS as its operand; h is read once and written once per layer
    const void* att;          // [M, 256] 16-bit; fp16c: two planes [2][M, 256] = fp16(64 a), fp16(64 a - hi) (attention.hip, HILO)
    const void* w_o;
    const float *b_o, *ln1_g, *ln1_b;
};

template <int PREC>
__global__ __launch_bounds__(512) void enc_ffn16_kernel(FfnArgs m) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int NCH = TFF / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);
    elem* Hs = As + 128 * RS16;
    float* P1 = reinterpret_cast<float*>(smem + (size_t)2 * 128 * RS16 * 2);
    float* P2 = P1 + 16 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const size_t row0 = (size_t)blockIdx.x * 128;
    const frag* w1 = reinterpret_cast<const frag*>(m.w1);
    const frag* w2 = reinterpret_cast<const frag*>(m.w2);
    f32x16 acc1[4], acc2[4];
    frag bs[2][1][SETK];
    const bool whole = m.att != nullptr;                      // uniform
    if (whole) {
        const frag* wo = reinterpret_cast<const frag*>(m.w_o);
        load_set<PREC, D, 1>(wo, 0, 0, 0, wave, lane, bs[0]);
        load_set<PREC, D, 1>(wo, 0, 0, 1, wave, lane, bs[1]);
        __builtin_amdgcn_sched_barrier(0);
        stage_rows16<PREC>(reinterpret_cast<const elem*>(m.att), row0, m.M, D, 0, As, tid);
        if constexpr (PREC == PREC_F16C) stage_rows16<PREC>(reinterpret_cast<const elem*>(m.att) + m.M * D, row0, m.M, D, 0, Hs, tid);
        __syncthreads();
        zero_acc(acc2);
        if constexpr (PREC == PREC_F16C) {
            // out_proj of both planes of the attention output by the same weights, then the exact 1/64 of their common scale
            phase_tm<PREC, D, D, true, true>(As, wo, 0, 0, wo, 0, 0, wave, lane, bs, acc2);
            phase_tm<PREC, D, D, true>(Hs, wo, 0, 0, w1, 0, 0, wave, lane, bs, acc2);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] *= 1.0f / 64.0f;
        } else {
            phase_tm<PREC, D, D, true, true>(As, wo, 0, 0, w1, 0, 0, wave, lane, bs, acc2);   // (the first fc1 set under its last set)
        }
        res_ln<PREC>(acc2, m.b_o, m.h, m.ln1_g, m.ln1_b, m.eps, row0, m.M, As, P1, P2);   // acc2 = x1, As = x1 (16-bit)
    } else {
        load_set<PREC, D, 1>(w1, 0, 0, 0, wave, lane, bs[0]);
        __builtin_amdgcn_sched_barrier(0);
        stage_rows16<PREC>(reinterpret_cast<const elem*>(m.hx_in), row0, m.M, D, 0, As, tid);
        __syncthreads();
        zero_acc(acc2);
    }
#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
        zero_acc(acc1);
        phase_tm<PREC, D, TFF, true>(As, w1, j, 0, w2, 0, j, wave, lane, bs, acc1);
        __syncthreads();                                       // every wave is done reading the previous chunk of Hs
        {
            const float* b1 = m.b1 + j * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + 8 * q);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    u16x4 pk = {to_bits<PREC>(fmaxf(acc1[mt][4 * q + 0] + bb.x, 0.f)), to_bits<PREC>(fmaxf(acc1[mt][4 * q + 1] + bb.y, 0.f)),
                                to_bits<PREC>(fmaxf(acc1[mt][4 * q + 2] + bb.z, 0.f)), to_bits<PREC>(fmaxf(acc1[mt][4 * q + 3] + bb.w, 0.f))};
                    *reinterpret_cast<u16x4*>(Hs + (mt * 32 + lrow) * RS16 + wave * 32 + 8 * q + 4 * lhalf) = pk;
                }
            }
        }
        __syncthreads();
        phase_tm<PREC, TFF, D, true>(Hs, w2, 0, j, w1, j + 1 < NCH ? j + 1 : 0, 0, wave, lane, bs, acc2);
    }
    if (m.w_qkv) load_set<PREC, D, 1>(reinterpret_cast<const frag*>(m.w_qkv), 0, 0, 0, wave, lane, bs[0]);
    const int l_valid = res_ln<PREC>(acc2, m.b2, whole ? nullptr : m.h, m.ln_g, m.ln_b, m.eps, row0, m.M, As, P1, P2);
    if (m.w_qkv) {   // in_proj of the next layer on the normalised tile while it is in LDS (wave tiles in the Hs + table region)
        elem* zs = Hs + wave * 128 * ZRS;
        act_blocks<PREC, TQKV>(As, zs, reinterpret_cast<const frag*>(m.w_qkv), m.b_qkv, reinterpret_cast<elem*>(m.qkv), row0, m.M,
                               false, bs, acc1);
    }
    store_h_hx<PREC>(acc2, m.h, reinterpret_cast<elem*>(m.hx_out), row0, l_valid, smem, As);
}

// ------------------------------------------------------------------------------------------------ pooling + classifier
// attn_weights = softmax_t(h[t] . w + b) over ALL positions of the read; pooled = sum_t a_t h[t]; Linear(256,128) ReLU Linear(128,2)
__global__ __launch_bounds__(256) void pool_head_kernel(const float* __restrict__ h, const float* __restrict__ pw,
                                                        const float* __restrict__ pb, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, float* __restrict__ scores,
                                                        float* __restrict__ pooled_out, float* __restrict__ logits, int L3) {
    __shared__ float red[4], vec[4][D], xin[D], hid[TCH];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* hb = h + (size_t)b * L3 * D;
    float* sc = scores + (size_t)b * L3;
    const float4 w4 = *reinterpret_cast<const float4*>(pw + lane * 4);
    const float bias = pb[0];
    float mx = -INFINITY;
    for (int t = wave; t < L3; t += 4) {                      // one wave per position
        const float4 x = *reinterpret_cast<const float4*>(hb + (size_t)t * D + lane * 4);
        const float s = wave_sum((x.x * w4.x + x.y * w4.y) + (x.z * w4.z + x.w * w4.w)) + bias;
        if (lane == 0) sc[t] = s;
        mx = fmaxf(mx, s);
    }
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ssum = 0.f;
    for (int t = wave; t < L3; t += 4) {
        const float e = expf(sc[t] - mx);                     // written by lane 0 of this same wave above
        const float4 x = *reinterpret_cast<const float4*>(hb + (size_t)t * D + lane * 4);
        ssum += e;
        a0 = fmaf(e, x.x, a0); a1 = fmaf(e, x.y, a1); a2 = fmaf(e, x.z, a2); a3 = fmaf(e, x.w, a3);
    }
    *reinterpret_cast<float4*>(&vec[wave][lane * 4]) = make_float4(a0, a1, a2, a3);
    if (lane == 0) red[wave] = ssum;
    __syncthreads();
    {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float p = ((vec[0][tid] + vec[1][tid]) + (vec[2][tid] + vec[3][tid])) / tot;
        xin[tid] = p;
        pooled_out[(size_t)b * D + tid] = p;
    }
    __syncthreads();
    if (tid < TCH) {
        float acc = b0[tid];
        const float* wr = w0 + (size_t)tid * D;
        for (int i = 0; i < D; ++i) acc = fmaf(wr[i], xin[i], acc);
        hid[tid] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (tid < NCLS) {
        float acc = b1[tid];
        for (int i = 0; i < TCH; ++i) acc = fmaf(w1[(size_t)tid * TCH + i], hid[i], acc);
        logits[(size_t)b * NCLS + tid] = acc;
    }
}

}  // namespace tf

void launch_attention_fwd(int prec, const void* qkv, void* out, int B, int L, hipStream_t st, bool hilo);   // attention.hip
// tf_fp32.hip: the same forward with exact-fp32 products (the reference's precision), up to the encoder output h
size_t tf32_workspace_floats(int B, int L);
int tf32_forward(const unsigned char* ids8, int ids_stride, int B, int L, int n_layers, float* ws, float* h,
                 const float* (*get)(void*, const std::string&), void* ctx, hipStream_t st, bool unfused, bool x3);

}  // namespace clm

// ================================================================================================ engine + C ABI
using namespace clm;

struct clm_tf_handle {
    int device = 0, prec = PREC_F16, n_layers = 12;
    std::string err;
    std::map<std::string, float*> w;              // fp32 device copies by reference key
    std::map<std::string, std::vector<int64_t>> shape;
    std::map<std::string, void*> packed;
    bool finalized = false;
    // workspace
    size_t cap_rows = 0, cap_tok = 0, cap_B = 0;            // cap_B: reads `pooled` has rows for (it scales with B alone)
    size_t cap16[7] = {};                         // bytes of x1, x2, x3, hx, qkv, att, u
    unsigned char* ids8 = nullptr;
    void *x1 = nullptr, *x2 = nullptr, *x3 = nullptr, *hx = nullptr, *qkv = nullptr, *att = nullptr, *u = nullptr;
    float *h = nullptr, *scores = nullptr, *pooled = nullptr;
    bool referee = false;                         // inside clm_tf_selfcheck's second pass: an fp16x3 handle runs its exact-fp32 kernels
    bool arith_x3 = false;                        // CLM_PREC_F16X3: prec is PREC_F32, the fused fp32-path kernels run on hi + lo halfs
    bool unfused_fp32 = false;                    // CLM_DEBUG=unfused_fp32 at creation: the separate fp32 launches of round 2 (tests cross-check the fused kernels)
    float* ws32 = nullptr;                        // fp32 mode: activations of tf_fp32.hip
    size_t cap_ws32 = 0;
    int last_B = 0, last_L3 = 0;
    // clm_tf_set_fallback (as clm_set_fallback): 0 = the handle's mode; 1 = the next arithmetic inside the gate (16-bit handle: the
    // fp32-path kernels on hi + lo halfs = fp16x3; fp16x3 handle: exact fp32); 2 = exact fp32 (the raw fp32 tensors stay loaded)
    int fallback = 0;
    bool have_x3 = false;                         // the "x3.*" packings exist (fp16x3 and 16-bit handles, fused kernels only)
    float* sc_logits = nullptr;                   // clm_tf_selfcheck: [2][cap] device logits of the two forwards
    int sc_cap = 0;
    // profiling taps (clm_tf_profile_*): HIP events on the launch stream around each stage, stages: 0 conv stack + pe/LN,
    // 1 attention, 2 encoder layer kernel (out_proj + LN1 + FFN + LN2 + next QKV; the unfused pieces count here too), 3 pooling head
    bool prof = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> recs;
    double prof_ms[4] = {};
    int64_t prof_n[4] = {};
};

namespace {

std::string g_tf_create_error;

struct TfTimer {                                   // RAII: one event pair per stage span when profiling is on
    clm_tf_handle* h;
    hipStream_t st;
    int stage;
    hipEvent_t e0{}, e1{};
    bool on;
    TfTimer(clm_tf_handle* h_, hipStream_t st_, int stage_) : h(h_), st(st_), stage(stage_), on(h_->prof && h_->recs.size() < 100000) {
        if (!on) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e0, st);
    }
    ~TfTimer() {
        if (!on) return;
        (void)hipEventRecord(e1, st);
        h->recs.push_back({stage, {e0, e1}});
    }
};

int tf_fail(clm_tf_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_tf_create_error = msg;
    return code;
}
#define TFCHK(h, call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return tf_fail(h, CLM_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

std::string tf_canon(const char* key) {
    std::string k(key);
    if (k.rfind("net.", 0) == 0) k = k.substr(4);
    return k;
}

std::map<std::string, std::vector<int64_t>> tf_expected(int n_layers) {
    std::map<std::string, std::vector<int64_t>> e;
    e["embedding.weight"] = {tf::TVOC, D};
    for (int i : {0, 3, 6}) {
        e["cnn." + std::to_string(i) + ".weight"] = {D, D, 3};
        e["cnn." + std::to_string(i) + ".bias"] = {D};
    }
    e["norm.weight"] = {D}; e["norm.bias"] = {D};
    for (int i = 0; i < n_layers; ++i) {
        const std::string p = "transformer_encoder.layers." + std::to_string(i) + ".";
        e[p + "self_attn.in_proj_weight"] = {tf::TQKV, D}; e[p + "self_attn.in_proj_bias"] = {tf::TQKV};
        e[p + "self_attn.out_proj.weight"] = {D, D}; e[p + "self_attn.out_proj.bias"] = {D};
        e[p + "linear1.weight"] = {tf::TFF, D}; e[p + "linear1.bias"] = {tf::TFF};
        e[p + "linear2.weight"] = {D, tf::TFF}; e[p + "linear2.bias"] = {D};
        e[p + "norm1.weight"] = {D}; e[p + "norm1.bias"] = {D}; e[p + "norm2.weight"] = {D}; e[p + "norm2.bias"] = {D};
    }
    e["attn_pool.weight"] = {1, D}; e["attn_pool.bias"] = {1};
    e["classifier.0.weight"] = {tf::TCH, D}; e["classifier.0.bias"] = {tf::TCH};
    e["classifier.3.weight"] = {NCLS, tf::TCH}; e["classifier.3.bias"] = {NCLS};
    return e;
}

void tf_free_ws(clm_tf_handle* h) {
    for (void* p : {(void*)h->ids8, h->x1, h->x2, h->x3, h->hx, h->qkv, h->att, h->u, (void*)h->h, (void*)h->scores,
                    (void*)h->pooled, (void*)h->ws32})
        if (p) (void)hipFree(p);
    h->ids8 = nullptr; h->x1 = h->x2 = h->x3 = h->hx = h->qkv = h->att = h->u = nullptr;
    h->h = h->scores = h->pooled = h->ws32 = nullptr;
    h->cap_rows = h->cap_tok = h->cap_ws32 = h->cap_B = 0;
    for (size_t& c : h->cap16) c = 0;
}

template <int PREC, int EPI, int K, int N>
void tf_launch_linear(const tf::LinArgs& a, hipStream_t st) {
    constexpr size_t lds = EPI == tf::E_ACT ? (size_t)(128 * RS16 + 8 * 128 * tf::ZRS) * 2
                                            : (size_t)2 * 128 * RS16 * 2 + (size_t)2 * 16 * 128 * 4;
    auto kern = tf::linear16_kernel<PREC, EPI, K, N>;
    CLM_SET_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 127) / 128)), dim3(512), lds, st, a);
}

template <int PREC>
int tf_forward_t(clm_tf_handle* h, const void* ids, int ids_dtype, int64_t stride, int B, int L, float* logits, hipStream_t st) {
    using elem = typename CT<PREC>::elem;
    const int L1 = L / 2, L2 = L1 / 2, L3 = L2 / 2, Lp = (L + 63) / 64 * 64;
    const size_t M = (size_t)B * L3;
    auto W = [&](const std::string& k) { return h->w.at(k); };
    TfTimer* tconv = new TfTimer(h, st, 0);
    launch_embed(ids, ids_dtype, stride, nullptr, nullptr, h->ids8, B, L, Lp, st);      // ids of any dtype -> clamped uint8
    constexpr size_t conv_lds = (size_t)(130 * RS16 + 8 * 64 * tf::ZRS) * 2;
    {
        auto k1 = tf::conv3_relu_pool_kernel<PREC, true>;
        auto k2 = tf::conv3_relu_pool_kernel<PREC, false>;
        CLM_SET_LDS(k1, conv_lds);
        CLM_SET_LDS(k2, conv_lds);
        hipLaunchKernelGGL(k1, dim3((2 * L1 + 127) / 128, B), dim3(512), conv_lds, st, h->ids8, Lp, W("embedding.weight"),
                           (const elem*)nullptr, h->packed.at("cnn.0"), W("cnn.0.bias"), (elem*)h->x1, L, L1);
        hipLaunchKernelGGL(k2, dim3((2 * L2 + 127) / 128, B), dim3(512), conv_lds, st, (const unsigned char*)nullptr, 0,
                           (const float*)nullptr, (const elem*)h->x1, h->packed.at("cnn.3"), W("cnn.3.bias"), (elem*)h->x2, L1, L2);
        hipLaunchKernelGGL(k2, dim3((2 * L3 + 127) / 128, B), dim3(512), conv_lds, st, (const unsigned char*)nullptr, 0,
                           (const float*)nullptr, (const elem*)h->x2, h->packed.at("cnn.6"), W("cnn.6.bias"), (elem*)h->x3, L2, L3);
    }
    hipLaunchKernelGGL(tf::pe_ln_kernel<PREC>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const elem*)h->x3,
                       W("pos_encoder.pe"), W("norm.weight"), W("norm.bias"), h->h, (elem*)h->hx, M, L3, 1e-5f);
    delete tconv;
    for (int i = 0; i < h->n_layers; ++i) {
        const std::string p = "transformer_encoder.layers." + std::to_string(i) + ".";
        tf::LinArgs a{};
        a.M = M; a.eps = 1e-5f;
        // (the unfused A/B forms read the attention output as ONE 16-bit plane: not for fp16c, whose out_proj takes two)
        static const bool unfused_ffn_env = debug_flag("tf_unfused_ffn");
        const bool unfused_ffn = unfused_ffn_env && PREC != PREC_F16C;
        if (i == 0 || unfused_ffn) {   // later layers: computed at the end of the previous layer's feed-forward kernel
            TfTimer t(h, st, 2);
            a.a = h->hx; a.w = h->packed.at(p + "in"); a.bias = W(p + "self_attn.in_proj_bias"); a.out16 = h->qkv; a.relu = 0;
            tf_launch_linear<PREC, tf::E_ACT, D, tf::TQKV>(a, st);
        }
        {
            TfTimer t(h, st, 1);
            launch_attention_fwd(PREC, h->qkv, h->att, B, L3, st, PREC == PREC_F16C);   // fp16c: hi and lo planes (attention.hip)
        }
        TfTimer tl(h, st, 2);
        static const bool unfused_layer_env = debug_flag("tf_unfused_layer");
        const bool unfused_layer = unfused_layer_env && PREC != PREC_F16C;
        const bool whole_layer = !unfused_layer && !unfused_ffn;   // out_proj + LN1 at the head of the feed-forward kernel
        if (!whole_layer) {
            a.a = h->att; a.w = h->packed.at(p + "out"); a.bias = W(p + "self_attn.out_proj.bias"); a.h = h->h; a.hx = h->hx;
            a.ln_g = W(p + "norm1.weight"); a.ln_b = W(p + "norm1.bias");
            tf_launch_linear<PREC, tf::E_RES_LN, D, D>(a, st);
        }
        if (!unfused_ffn) {
            const bool more = i + 1 < h->n_layers;
            const std::string pn = "transformer_encoder.layers." + std::to_string(i + 1) + ".";
            tf::FfnArgs f{h->hx, h->packed.at(p + "ff1"), h->packed.at(p + "ff2"), W(p + "linear1.bias"), W(p + "linear2.bias"),
                          W(p + "norm2.weight"), W(p + "norm2.bias"), h->h, h->hx, M, 1e-5f,
                          more ? h->packed.at(pn + "in") : nullptr, more ? W(pn + "self_attn.in_proj_bias") : nullptr, h->qkv,
                          whole_layer ? h->att : nullptr, h->packed.at(p + "out"), W(p + "self_attn.out_proj.bias"),
                          W(p + "norm1.weight"), W(p + "norm1.bias")};
            constexpr size_t lds = (size_t)2 * 128 * RS16 * 2 + (size_t)2 * 16 * 128 * 4;
            auto kern = tf::enc_ffn16_kernel<PREC>;
            CLM_SET_LDS(kern, lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)((M + 127) / 128)), dim3(512), lds, st, f);
        } else {
            a.a = h->hx; a.w = h->packed.at(p + "ff1"); a.bias = W(p + "linear1.bias"); a.out16 = h->u; a.relu = 1;
            tf_launch_linear<PREC, tf::E_ACT, D, tf::TFF>(a, st);
            a.a = h->u; a.w = h->packed.at(p + "ff2"); a.bias = W(p + "linear2.bias");
            a.ln_g = W(p + "norm2.weight"); a.ln_b = W(p + "norm2.bias");
            tf_launch_linear<PREC, tf::E_RES_LN, tf::TFF, D>(a, st);
        }
    }
    {
        TfTimer t(h, st, 3);
        hipLaunchKernelGGL(tf::pool_head_kernel, dim3(B), dim3(256), 0, st, h->h, W("attn_pool.weight"), W("attn_pool.bias"),
                           W("classifier.0.weight"), W("classifier.0.bias"), W("classifier.3.weight"), W("classifier.3.bias"),
                           h->scores, h->pooled, logits, L3);
    }
    h->last_B = B; h->last_L3 = L3;
    return hipGetLastError() == hipSuccess ? CLM_OK : CLM_E_HIP;
}

}  // namespace

extern "C" {

int clm_tf_create(int device, int precision, int n_layers, clm_tf_handle** out) {
    if (!out || n_layers < 1 || n_layers > 64) return tf_fail(nullptr, CLM_E_INVALID, "clm_tf_create: bad argument");
    if (precision != CLM_PREC_F16 && precision != CLM_PREC_BF16 && precision != CLM_PREC_F32 && precision != CLM_PREC_F16C &&
        precision != CLM_PREC_F16X3)
        return tf_fail(nullptr, CLM_E_UNSUPPORTED,
                       "clm_tf_create: precision must be fp32 (exact, the reference's arithmetic), fp16x3, fp16c, fp16 or bf16");
    if (hipSetDevice(device) != hipSuccess) return tf_fail(nullptr, CLM_E_HIP, "clm_tf_create: hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return tf_fail(nullptr, CLM_E_UNSUPPORTED, "clm_tf_create: this engine is built for gfx950 (MI355X) only");
    clm_tf_handle* h = new clm_tf_handle();
    h->unfused_fp32 = debug_flag("unfused_fp32");
    h->device = device;
    // fp16x3: the exact-fp32 path with its fused kernels multiplying hi + lo halfs (three fp16 MFMAs per product; tail32.hip AR_X3)
    h->arith_x3 = precision == CLM_PREC_F16X3;
    if (h->arith_x3 && h->unfused_fp32) {
        delete h;
        return tf_fail(nullptr, CLM_E_UNSUPPORTED, "clm_tf_create: fp16x3 exists in the fused kernels only (CLM_DEBUG=unfused_fp32 is set)");
    }
    h->prec = precision == CLM_PREC_BF16 ? PREC_BF16 : (precision == CLM_PREC_F32 || h->arith_x3) ? PREC_F32 : precision == CLM_PREC_F16C ? PREC_F16C
                                                                                                                                   : PREC_F16;
    h->n_layers = n_layers;
    *out = h;
    return CLM_OK;
}

int clm_tf_load_weight(clm_tf_handle* h, const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!h || !key || !data || !shape) return tf_fail(h, CLM_E_INVALID, "clm_tf_load_weight: null argument");
    if (dtype != CLM_DT_F32) return tf_fail(h, CLM_E_INVALID, "clm_tf_load_weight: fp32 tensors only");
    const std::string k = tf_canon(key);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (k == "pos_encoder.pe") {                               // buffer [1, max_len, 256]
        if (ndim != 3 || shp[0] != 1 || shp[2] != D) return tf_fail(h, CLM_E_INVALID, "pos_encoder.pe: expected [1, max_len, 256]");
    } else {
        const auto exp = tf_expected(h->n_layers);
        auto it = exp.find(k);
        if (it == exp.end()) return tf_fail(h, CLM_E_INVALID, "clm_tf_load_weight: unknown key " + k);
        if (it->second != shp) return tf_fail(h, CLM_E_INVALID, "clm_tf_load_weight: wrong shape for " + k);
    }
    size_t n = 1;
    for (int64_t s : shp) n *= (size_t)s;
    TFCHK(h, hipSetDevice(h->device));
    if (h->w.count(k)) { (void)hipFree(h->w[k]); h->w.erase(k); }
    float* d = nullptr;
    TFCHK(h, hipMalloc((void**)&d, n * 4));
    TFCHK(h, hipMemcpy(d, data, n * 4, hipMemcpyDefault));
    h->w[k] = d;
    h->shape[k] = shp;
    h->finalized = false;
    return CLM_OK;
}

int clm_tf_finalize(clm_tf_handle* h) {
    if (!h) return CLM_E_INVALID;
    TFCHK(h, hipSetDevice(h->device));
    for (const auto& kv : tf_expected(h->n_layers))
        if (!h->w.count(kv.first)) return tf_fail(h, CLM_E_MISSING, "clm_tf_finalize: missing weight " + kv.first);
    if (!h->w.count("pos_encoder.pe")) return tf_fail(h, CLM_E_MISSING, "clm_tf_finalize: missing buffer pos_encoder.pe");
    for (auto& kv : h->packed) (void)hipFree(kv.second);
    h->packed.clear();
    h->have_x3 = (h->arith_x3 || h->prec != PREC_F32) && !h->unfused_fp32;
    // exact fp32 (the handle's own mode, or the referee / fall-back of a 16-bit handle): the dense layers of the encoder in the fused
    // kernel's packing (tail32.hip enc32_kernel); everything else of tf_fp32.hip reads the fp32 tensors as they are
    for (int i = 0; i < h->n_layers; ++i) {
        const std::string p = "transformer_encoder.layers." + std::to_string(i) + ".", t = "t32." + std::to_string(i) + ".";
        struct { const char* name; const char* key; int n, k; } tw[4] = {{"in", "self_attn.in_proj_weight", tf::TQKV, D}, {"out", "self_attn.out_proj.weight", D, D},
                                                                         {"ff1", "linear1.weight", tf::TFF, D}, {"ff2", "linear2.weight", D, tf::TFF}};
        for (auto& e : tw) {
            void* q = nullptr;
            TFCHK(h, hipMalloc(&q, (size_t)e.n * e.k * 4));
            launch_pack_f32t(h->w.at(p + e.key), q, e.n, e.k, 0);
            h->packed[t + e.name] = q;
            if (h->have_x3) {                                 // fp16x3 (a handle's mode, or a 16-bit handle's first fall-back level): the same weights as hi + lo halfs ("x3." keys); "t32." = the referee
                void* qx = nullptr;
                TFCHK(h, hipMalloc(&qx, (size_t)e.n * e.k * 4));
                launch_pack_x3(h->w.at(p + e.key), qx, e.n, e.k, 0);
                h->packed["x3." + std::to_string(i) + "." + e.name] = qx;
            }
        }
    }
    {   // ... and the three taps of each CNN-stem convolution ([co][ci][3] -> [3][co][ci], each tap packed)
        float* split32 = nullptr;
        TFCHK(h, hipMalloc((void**)&split32, (size_t)3 * D * D * 4));
        for (int i : {0, 3, 6}) {
            const std::string name = "cnn." + std::to_string(i);
            hipLaunchKernelGGL(tf::conv_w_split_kernel, dim3((3 * D * D + 255) / 256), dim3(256), 0, 0, h->w.at(name + ".weight"), split32);
            float* q = nullptr;
            TFCHK(h, hipMalloc((void**)&q, (size_t)3 * D * D * 4));
            for (int dk = 0; dk < 3; ++dk) launch_pack_f32t(split32 + (size_t)dk * D * D, q + (size_t)dk * D * D, D, D, 0);
            h->packed["t32." + name] = q;
            if (h->have_x3) {
                float* qx = nullptr;
                TFCHK(h, hipMalloc((void**)&qx, (size_t)3 * D * D * 4));
                for (int dk = 0; dk < 3; ++dk) launch_pack_x3(split32 + (size_t)dk * D * D, qx + (size_t)dk * D * D, D, D, 0);
                h->packed["x3." + name] = qx;
            }
            TFCHK(h, hipDeviceSynchronize());                 // `split32` is reused by the next layer
        }
        (void)hipFree(split32);
    }
    if (h->prec == PREC_F32) {
        TFCHK(h, hipDeviceSynchronize());
        h->finalized = true;
        return CLM_OK;
    }
    auto pack = [&](const std::string& name, const float* w, int n, int k) -> int {
        void* p = nullptr;
        TFCHK(h, hipMalloc(&p, packed_weight_bytes(h->prec, n, k)));
        launch_pack_weight(h->prec, w, p, n, k, 0);
        h->packed[name] = p;
        return CLM_OK;
    };
    float* split = nullptr;
    TFCHK(h, hipMalloc((void**)&split, (size_t)3 * D * D * 4));
    for (int i : {0, 3, 6}) {
        const std::string name = "cnn." + std::to_string(i);
        hipLaunchKernelGGL(tf::conv_w_split_kernel, dim3((3 * D * D + 255) / 256), dim3(256), 0, 0, h->w.at(name + ".weight"), split);
        void* p = nullptr;
        TFCHK(h, hipMalloc(&p, 3 * packed_weight_bytes(h->prec, D, D)));
        for (int dk = 0; dk < 3; ++dk)
            launch_pack_weight(h->prec, split + (size_t)dk * D * D, (char*)p + dk * packed_weight_bytes(h->prec, D, D), D, D, 0);
        TFCHK(h, hipDeviceSynchronize());                     // `split` is reused by the next layer
        h->packed[name] = p;
    }
    (void)hipFree(split);
    for (int i = 0; i < h->n_layers; ++i) {
        const std::string p = "transformer_encoder.layers." + std::to_string(i) + ".";
        int rc;
        if ((rc = pack(p + "in", h->w.at(p + "self_attn.in_proj_weight"), tf::TQKV, D))) return rc;
        if ((rc = pack(p + "out", h->w.at(p + "self_attn.out_proj.weight"), D, D))) return rc;
        if ((rc = pack(p + "ff1", h->w.at(p + "linear1.weight"), tf::TFF, D))) return rc;
        if ((rc = pack(p + "ff2", h->w.at(p + "linear2.weight"), D, tf::TFF))) return rc;
    }
    TFCHK(h, hipDeviceSynchronize());
    h->finalized = true;
    return CLM_OK;
}

static bool tf_fp32_path_is_x3(const clm_tf_handle* h) {
    if (h->referee || !h->have_x3) return false;
    return h->arith_x3 ? h->fallback == 0 : h->fallback < 2;
}

// One forward in the arithmetic `prec32 ? the fp32 path (exact, or fp16x3: tf_fp32_path_is_x3) : the handle's 16-bit mode` (workspaces of BOTH kinds may be live: the
// self-check runs one after the other on the same ids).
static int tf_run(clm_tf_handle* h, bool prec32, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L,
                  float* logits_out, hipStream_t st) {
    const int L3 = L / 8;
    const size_t M = (size_t)B * L3, tok = (size_t)B * ((L + 63) / 64 * 64);
    if (M > h->cap_rows || tok > h->cap_tok) {                // shared pieces: ids, residual stream, pooling scores
        TFCHK(h, hipDeviceSynchronize());
        tf_free_ws(h);
        TFCHK(h, hipMalloc((void**)&h->ids8, tok));
        TFCHK(h, hipMalloc((void**)&h->h, M * D * 4));
        TFCHK(h, hipMalloc((void**)&h->scores, M * 4));
        h->cap_rows = M; h->cap_tok = tok;
    }
    if ((size_t)B > h->cap_B) {                               // pooled [B][256]: its own capacity -- a batch of MORE, SHORTER reads fits
        TFCHK(h, hipDeviceSynchronize());                     // cap_rows / cap_tok and must still regrow this one
        if (h->pooled) (void)hipFree(h->pooled);
        h->pooled = nullptr; h->cap_B = 0;
        TFCHK(h, hipMalloc((void**)&h->pooled, (size_t)B * D * 4));
        h->cap_B = (size_t)B;
    }
    if (prec32) {
        const size_t need = tf32_workspace_floats(B, L);
        if (need > h->cap_ws32) {
            TFCHK(h, hipDeviceSynchronize());
            if (h->ws32) (void)hipFree(h->ws32);
            h->ws32 = nullptr; h->cap_ws32 = 0;
            TFCHK(h, hipMalloc((void**)&h->ws32, need * 4));
            h->cap_ws32 = need;
        }
        const int Lp = (L + 63) / 64 * 64;
        launch_embed(ids, ids_dtype, ids_row_stride, nullptr, nullptr, h->ids8, B, L, Lp, st);
        // hi + lo halfs on the fp32 path: an fp16x3 handle unless told to fall back; a 16-bit handle at fall-back level 1
        const bool use_x3 = tf_fp32_path_is_x3(h);
        auto get = [](void* ctx, const std::string& k) -> const float* {     // "t32.*": the fused kernels' packed weights ("x3.*" in fp16x3)
            auto* hh = static_cast<clm_tf_handle*>(ctx);
            if (k.rfind("t32.", 0) != 0) return hh->w.at(k);
            return static_cast<const float*>(hh->packed.at(tf_fp32_path_is_x3(hh) ? "x3." + k.substr(4) : k));
        };
        if (tf32_forward(h->ids8, Lp, B, L, h->n_layers, h->ws32, h->h, get, h, st, h->unfused_fp32, use_x3))
            return tf_fail(h, CLM_E_HIP, std::string("clm_tf_forward (fp32): ") + hipGetErrorString(hipGetLastError()));
        auto W = [&](const std::string& k) { return h->w.at(k); };
        hipLaunchKernelGGL(tf::pool_head_kernel, dim3(B), dim3(256), 0, st, h->h, W("attn_pool.weight"), W("attn_pool.bias"),
                           W("classifier.0.weight"), W("classifier.0.bias"), W("classifier.3.weight"), W("classifier.3.bias"),
                           h->scores, h->pooled, logits_out, L3);
        h->last_B = B; h->last_L3 = L3;
        return hipGetLastError() == hipSuccess ? CLM_OK : tf_fail(h, CLM_E_HIP, "clm_tf_forward (fp32): launch failed");
    }
    {   // the 16-bit activations, each buffer grown on its own (x1 / x2 scale with B * (L / 2), B * (L / 4), not with M)
        const size_t need[7] = {(size_t)B * (L / 2) * D * 2, (size_t)B * (L / 4) * D * 2, M * D * 2, M * D * 2, M * tf::TQKV * 2,
                                M * D * 2 * (h->prec == PREC_F16C ? 2 : 1) /* hi + lo planes */, M * tf::TFF * 2};
        void** buf[7] = {&h->x1, &h->x2, &h->x3, &h->hx, &h->qkv, &h->att, &h->u};
        bool grow = false;
        for (int i = 0; i < 7; ++i) grow |= need[i] > h->cap16[i];
        if (grow) {
            TFCHK(h, hipDeviceSynchronize());
            for (int i = 0; i < 7; ++i)
                if (need[i] > h->cap16[i]) {
                    if (*buf[i]) (void)hipFree(*buf[i]);
                    *buf[i] = nullptr; h->cap16[i] = 0;
                    TFCHK(h, hipMalloc(buf[i], need[i]));
                    h->cap16[i] = need[i];
                }
        }
    }
    const int rc = h->prec == PREC_BF16   ? tf_forward_t<PREC_BF16>(h, ids, ids_dtype, ids_row_stride, B, L, logits_out, st)
                   : h->prec == PREC_F16C ? tf_forward_t<PREC_F16C>(h, ids, ids_dtype, ids_row_stride, B, L, logits_out, st)
                                          : tf_forward_t<PREC_F16>(h, ids, ids_dtype, ids_row_stride, B, L, logits_out, st);
    if (rc) return tf_fail(h, rc, std::string("clm_tf_forward: ") + hipGetErrorString(hipGetLastError()));
    return CLM_OK;
}

static int tf_check_args(clm_tf_handle* h, const char* who, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L) {
    if (!h->finalized) return tf_fail(h, CLM_E_STATE, std::string(who) + " before clm_tf_finalize");
    if (!ids || B < 1 || L < 8 || ids_row_stride < L) return tf_fail(h, CLM_E_INVALID, std::string(who) + ": bad argument (L >= 8)");
    if (ids_dtype != CLM_DT_I64 && ids_dtype != CLM_DT_I32 && ids_dtype != CLM_DT_U8)
        return tf_fail(h, CLM_E_INVALID, std::string(who) + ": ids dtype must be i64, i32 or u8");
    if ((int64_t)(L / 8) > h->shape.at("pos_encoder.pe")[1])
        return tf_fail(h, CLM_E_INVALID, std::string(who) + ": Sequence too long (" + std::to_string(L / 8) + " > max_len of pos_encoder.pe)");
    return CLM_OK;
}

int clm_tf_forward(clm_tf_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, float* logits_out,
                   void* stream) {
    if (!h) return CLM_E_INVALID;
    if (!logits_out) return tf_fail(h, CLM_E_INVALID, "clm_tf_forward: bad argument (L >= 8)");
    if (int rc = tf_check_args(h, "clm_tf_forward", ids, ids_dtype, ids_row_stride, B, L)) return rc;
    TFCHK(h, hipSetDevice(h->device));
    return tf_run(h, h->prec == PREC_F32 || h->fallback > 0, ids, ids_dtype, ids_row_stride, B, L, logits_out,
                  reinterpret_cast<hipStream_t>(stream));
}

// The handle's 16-bit mode on trial against the exact-fp32 kernels of the same handle, on the caller's ids (chimeralm_hip.h;
// the counterpart of clm_selfcheck).  Synchronises `stream`.
int clm_tf_selfcheck(clm_tf_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, void* stream,
                     float* max_abs_diff_out, int* labels_differ_out) {
    if (!h) return CLM_E_INVALID;
    if (int rc = tf_check_args(h, "clm_tf_selfcheck", ids, ids_dtype, ids_row_stride, B, L)) return rc;
    if (B > 4096) return tf_fail(h, CLM_E_INVALID, "clm_tf_selfcheck: at most 4096 reads per call");
    TFCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float diff = 0.f;
    int differ = 0;
    if (h->prec != PREC_F32 || h->arith_x3) {                  // (an fp16x3 handle: its own exact-fp32 kernels are the referee)
        if (B > h->sc_cap) {
            TFCHK(h, hipDeviceSynchronize());
            if (h->sc_logits) (void)hipFree(h->sc_logits);
            h->sc_logits = nullptr; h->sc_cap = 0;
            TFCHK(h, hipMalloc((void**)&h->sc_logits, (size_t)2 * B * NCLS * 4));
            h->sc_cap = B;
        }
        float* lm = h->sc_logits;
        float* lx = h->sc_logits + (size_t)h->sc_cap * NCLS;
        const int fallback = h->fallback;
        h->fallback = 0;                                       // the MODE is on trial, whatever clm_tf_set_fallback says
        const int rc1 = tf_run(h, h->arith_x3, ids, ids_dtype, ids_row_stride, B, L, lm, st);
        h->fallback = fallback;
        if (rc1) return rc1;
        h->referee = true;
        const int rc2 = tf_run(h, true, ids, ids_dtype, ids_row_stride, B, L, lx, st);
        h->referee = false;
        if (rc2) return rc2;
        std::vector<float> a((size_t)B * NCLS), b((size_t)B * NCLS);
        TFCHK(h, hipMemcpyAsync(a.data(), lm, a.size() * 4, hipMemcpyDeviceToHost, st));
        TFCHK(h, hipMemcpyAsync(b.data(), lx, b.size() * 4, hipMemcpyDeviceToHost, st));
        TFCHK(h, hipStreamSynchronize(st));
        (void)clm_logit_deviation(a.data(), b.data(), B, NCLS, &diff, &differ);   // (shared with clm_selfcheck: a non-finite logit is sticky +inf)
    }
    if (max_abs_diff_out) *max_abs_diff_out = diff;
    if (labels_differ_out) *labels_differ_out = differ;
    return CLM_OK;
}

int clm_tf_set_fallback(clm_tf_handle* h, int on) {
    if (!h) return CLM_E_INVALID;
    if (on < 0 || on > 2) return tf_fail(h, CLM_E_INVALID, "clm_tf_set_fallback: level must be 0, 1 or 2");
    h->fallback = (h->prec == PREC_F32 && !h->arith_x3) ? 0 : on;
    return CLM_OK;
}

int clm_tf_debug_fetch(clm_tf_handle* h, const char* name, void* host_out, size_t bytes) {
    if (!h || !name || !host_out) return tf_fail(h, CLM_E_INVALID, "clm_tf_debug_fetch: bad argument");
    TFCHK(h, hipSetDevice(h->device));
    TFCHK(h, hipDeviceSynchronize());
    const std::string n(name);
    const size_t M = (size_t)h->last_B * h->last_L3;
    const void* src = nullptr;
    size_t have = 0;
    if (n == "hidden") { src = h->h; have = M * D * 4; }
    else if (n == "scores") { src = h->scores; have = M * 4; }
    else if (n == "pooled") { src = h->pooled; have = (size_t)h->last_B * D * 4; }
    else return tf_fail(h, CLM_E_INVALID, "clm_tf_debug_fetch: unknown name " + n);
    if (bytes > have) return tf_fail(h, CLM_E_INVALID, "clm_tf_debug_fetch: more bytes requested than the last forward produced");
    TFCHK(h, hipMemcpy(host_out, src, bytes, hipMemcpyDeviceToHost));
    return CLM_OK;
}

int clm_tf_profile_enable(clm_tf_handle* h, int on) {
    if (!h) return CLM_E_INVALID;
    h->prof = on != 0;
    return CLM_OK;
}

int clm_tf_profile_read(clm_tf_handle* h, double* ms_out /*[4]*/, int64_t* spans_out /*[4]*/, int reset) {
    if (!h) return CLM_E_INVALID;
    TFCHK(h, hipSetDevice(h->device));
    TFCHK(h, hipDeviceSynchronize());
    for (auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.second.first, r.second.second) == hipSuccess) {
            h->prof_ms[r.first] += ms;
            h->prof_n[r.first] += 1;
        }
        (void)hipEventDestroy(r.second.first);
        (void)hipEventDestroy(r.second.second);
    }
    h->recs.clear();
    for (int i = 0; i < 4; ++i) {
        if (ms_out) ms_out[i] = h->prof_ms[i];
        if (spans_out) spans_out[i] = h->prof_n[i];
        if (reset) h->prof_ms[i] = 0, h->prof_n[i] = 0;
    }
    return CLM_OK;
}

const char* clm_tf_last_error(const clm_tf_handle* h) { return h ? h->err.c_str() : g_tf_create_error.c_str(); }

int clm_tf_destroy(clm_tf_handle* h) {
    if (!h) return CLM_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    tf_free_ws(h);
    if (h->sc_logits) (void)hipFree(h->sc_logits);
    for (auto& r : h->recs) { (void)hipEventDestroy(r.second.first); (void)hipEventDestroy(r.second.second); }
    for (auto& kv : h->w) (void)hipFree(kv.second);
    for (auto& kv : h->packed) (void)hipFree(kv.second);
    delete h;
    return CLM_OK;
}

}  // extern "C"
