// tail32.hip -- the fused kernels of the exact-fp32 engine and of its fp16x3 form (round 4):
//   tail32_kernel   Hyena block tail (this comment)                          enc32_kernel   SequenceCNNTransformer encoder layer
//   conv32_kernel   the transformer's CNN stem (conv k = 3 + ReLU + pool)     pack_f32t / pack_x3 kernels: the weight packings
// all on one 64-token tile, one weight-set pipeline and one MFMA loop, instantiated for two arithmetics (AR_F32: the fp32 MFMA;
// AR_X3: every operand as two halfs, three fp16 MFMAs per product -- see ARITH below).  The fp32 / x3 attention lives in tf_fp32.hip.
//
// One kernel per Hyena block, on v_mfma_f32_32x32x2_f32 throughout, for the part of the block that is token-wise
//   /root/reference/chimeralm/models/components/hyena.py:244-256 -> the backbone's block: x = x + mixer(norm1(x)); x = x + mlp(norm2(x))
//       r  = h + b_out + W_out y                    (y = the convolution's output, channel-major in HBM)
//       h' = r + b_2 + W_2 gelu_tanh(W_1 LN2(r) + b_1)
//       z  = W_in' LN1'(h') + b_in'                 (the NEXT block's in_proj, rows x0 | x1 | v as the fp32 convolution reads them)
// where round 1-3's fp32 path ran five separate GEMM kernels per block with the 1024-wide fc1 output, the normalised tiles and
// the intermediate residual all crossing HBM in fp32 (gemm.hip; kept: block 0's in_proj, the score layer, and the whole path
// under CLM_DEBUG=unfused_fp32, which tests/ uses to cross-check this kernel).
//
// Tile = 64 tokens x 256 features per workgroup (8 waves; a wave owns 32 output features of every product, both 32-token row
// tiles): an fp32 tile of 64 tokens fills the LDS a 16-bit tile of 128 does (As + Hs = 2 x 65 KiB).  Weights are the MFMA's A
// operand (accumulator rows = output features, lane = token), packed per (256-wide block, wave, 8-deep k-step, lane) as one
// float4 -- the four k values a lane feeds to four consecutive MFMAs: the reduction index may be permuted freely, so MFMA j of a
// k-step pairs k = 8 s + j (lanes 0-31) with k = 8 s + 4 + j (lanes 32-63) and both operands are 16-byte loads.  A 64-deep weight
// set (8 float4 = 32 registers) is requested one set ahead of its use; a set is 64 MFMAs of 64 cycles per wave, so neither the
// L2 latency nor the 16 ds_read_b128 of a set are anywhere near the matrix pipe's time: the kernel is MFMA-bound by construction
// (12 blocks x 256 MFMAs x 64 cycles x 2 waves per SIMD = 393k cycles per tile against ~25k of LayerNorm / GELU / epilogue VALU).
#include "chimeralm_hip.h"
#include "clm_common.h"

namespace clm {

namespace {

constexpr int BM32 = 64;          // tokens per tile
constexpr int RS32 = 260;         // row stride (floats) of the token-major tiles: 1040 B -- the 16 rows of a ds_read_b128 lane group
                                  // fall on 16 different 16-byte bank groups (260 = 4 mod 64)
constexpr int RSY = 72;           // row stride (floats) of the k-major y tile: the two half-waves of an MFMA read rows k and k + 4, i.e. 288 floats = banks + 32
constexpr int KS_SET = 8;         // k-steps (of 8) per weight set: 64 deep

using f32x4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const f32x4* wset_ptr(const f32x4* wp, int nb, int ksteps_all, int ks0, int wave, int lane) {
    return wp + ((size_t)(nb * 8 + wave) * ksteps_all + ks0) * 64 + lane;
}
__device__ __forceinline__ void load_wset(const f32x4* p, f32x4 (&ws)[KS_SET]) {
#pragma unroll
    for (int s = 0; s < KS_SET; ++s) ws[s] = p[(size_t)s * 64];
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---- ARITH: the arithmetic of the products.  AR_F32: exact fp32 on v_mfma_f32_32x32x2_f32.  AR_X3 ("fp16x3"): every operand split
// into two halfs, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (~21 bits), and a product as THREE fp16 MFMAs into the fp32
// accumulator -- w_hi a_hi + w_lo a_hi + w_hi a_lo (the dropped w_lo a_lo is 2^-22 of the product) -- at 96 cycles per 16-deep step
// and row tile where the fp32 MFMA takes 512.  A tile row then holds 256 hi halfs | 256 lo halfs in the 1024 bytes of its 256
// floats (same stride, same bank behaviour); a weight "fragment" is 8 halfs, (hi, lo) pairs alternating, so a 64-deep set is again
// 8 fragments of 16 bytes and the set machinery is shared.  Weights are packed x 2^10 (X3_WS), which keeps their lo halfs out of
// fp16's subnormal range down to |w| ~ 2.5e-4; accumulators that also hold unscaled terms (the residual) are scaled before and
// unscaled after their products (powers of two: exact).
enum { AR_F32 = 0, AR_X3 = 1 };
constexpr float X3_WS = 1024.f, X3_WSI = 1.0f / 1024.f;
template <int AR> constexpr float WSCALE = AR == AR_X3 ? X3_WS : 1.0f;
template <int AR> constexpr float WUNSCALE = AR == AR_X3 ? X3_WSI : 1.0f;
using v4i16 = short __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) * lds_v4i16_ptr32;
constexpr int RSKM64 = 96;        // AR_X3: row stride (halfs) of a k-major y plane [256 channels][64 tokens]: 192 B -- the four rows of a
                                  // transposing read fall on four different 64-byte bank groups (0, 192, 128, 64 mod 256)

__device__ __forceinline__ f32x16 mfma16(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// x -> (hi, lo) halfs.  Values beyond fp16's range do not become inf: hi saturates at +-65504 and lo carries the rest (up to twice
// the range; beyond that the pair saturates) -- the mode has no guard, so an outlier must not turn into a NaN logit.
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 clamp_h(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
    return r;
}
__device__ __forceinline__ void split4(f32x4 v, h4_t& hi, h4_t& lo) {
    hi = __builtin_convertvector(clamp_h(v), h4_t);
    lo = __builtin_convertvector(clamp_h(v - __builtin_convertvector(hi, f32x4)), h4_t);
}
// four consecutive features of one token row into a tile (fp32: 16 bytes; x3: 8 bytes of hi halfs + 8 bytes of lo halfs)
template <int AR>
__device__ __forceinline__ void tile_store4(float* T, int row, int col, f32x4 v) {
    if constexpr (AR == AR_F32) {
        *reinterpret_cast<f32x4*>(T + row * RS32 + col) = v;
    } else {
        h4_t hi, lo;
        split4(v, hi, lo);
        char* base = reinterpret_cast<char*>(T + row * RS32) + col * 2;
        *reinterpret_cast<h4_t*>(base) = hi;
        *reinterpret_cast<h4_t*>(base + 512) = lo;
    }
}

// acc[mt] += W-set x T[tokens mt*32.., k = 64 part ..): token-major tile
template <int AR>
__device__ __forceinline__ void compute_set_tm(const float* T, int part, int lrow, int lhalf, const f32x4 (&ws)[KS_SET], f32x16 (&acc)[2]) {
    if constexpr (AR == AR_X3) {
        // (the fragments of item i + 2 requested before the MFMAs of item i, pinned with sched_group_barrier, was measured: 4,121 vs
        //  4,180 reads/s on the Hyena path, 7,269 vs 7,637 on the transformer -- no gain over what hipcc schedules; the simple form stays)
        const char* a0 = reinterpret_cast<const char*>(T + lrow * RS32) + (part * 64 + lhalf * 8) * 2;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 ah = *reinterpret_cast<const f32x4*>(a0 + mt * 32 * RS32 * 4 + s * 32);
                const f32x4 al = *reinterpret_cast<const f32x4*>(a0 + 512 + mt * 32 * RS32 * 4 + s * 32);
                acc[mt] = mfma16(ws[2 * s], ah, acc[mt]);
                acc[mt] = mfma16(ws[2 * s + 1], ah, acc[mt]);
                acc[mt] = mfma16(ws[2 * s], al, acc[mt]);
            }
        return;
    }
    const float* a0 = T + lrow * RS32 + part * 64 + lhalf * 4;
    f32x4 af[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) af[0][mt] = *reinterpret_cast<const f32x4*>(a0 + mt * 32 * RS32);
#pragma unroll
    for (int s = 0; s < KS_SET; ++s) {
        if (s + 1 < KS_SET) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[(s + 1) & 1][mt] = *reinterpret_cast<const f32x4*>(a0 + mt * 32 * RS32 + (s + 1) * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma32(ws[s][j], af[s & 1][mt][j], acc[mt]);
    }
}
// same from the k-major tile (out_proj: y is channel-major).  fp32: Ys[k][token] floats; x3: two planes of halfs [k][RSKM64] read with
// the transposing LDS read (a 16-lane group reads a 4(k) x 16(token) block, lane i receives token i's four k values)
template <int AR>
__device__ __forceinline__ void compute_set_km(const float* Ys, int part, int lane, const f32x4 (&ws)[KS_SET], f32x16 (&acc)[2]) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    if constexpr (AR == AR_X3) {
        const unsigned short* Yh = reinterpret_cast<const unsigned short*>(Ys);
        const int li = lane & 15, g1 = (lane >> 4) & 1, q = li >> 2, p = li & 3;
        const unsigned short* base = Yh + (8 * lhalf + q) * RSKM64 + 16 * g1 + 4 * p;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned short* p0 = base + ((part * 4 + s) * 16) * RSKM64 + mt * 32;
                f32x4 a2[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const unsigned short* pp = p0 + pl * (D * RSKM64);
                    const v4i16 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr32)(pp));
                    const v4i16 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr32)(pp + 4 * RSKM64));
                    typedef short s8 __attribute__((ext_vector_type(8)));
                    const s8 both = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                    a2[pl] = __builtin_bit_cast(f32x4, both);
                }
                acc[mt] = mfma16(ws[2 * s], a2[0], acc[mt]);
                acc[mt] = mfma16(ws[2 * s + 1], a2[0], acc[mt]);
                acc[mt] = mfma16(ws[2 * s], a2[1], acc[mt]);
            }
        return;
    }
    const float* a0 = Ys + (part * 64 + lhalf * 4) * RSY + lrow;
#pragma unroll
    for (int s = 0; s < KS_SET; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma32(ws[s][j], a0[(s * 8 + j) * RSY + mt * 32], acc[mt]);
}

// one 256-deep product: sets 0..3 starting at fragment ks0 of (wp, nb); ws[0] holds set 0 on entry, and on exit the first set of
// what follows (`nxt`, requested under the last set: unconditional)
template <bool KM, int AR = AR_F32>
__device__ __forceinline__ void product256(const float* T, const f32x4* wp, int nb, int ksteps_all, int ks0, const f32x4* nxt,
                                           int wave, int lane, f32x4 (&ws)[2][KS_SET], f32x16 (&acc)[2]) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    static_for<0, 4>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < 3) load_wset(wset_ptr(wp, nb, ksteps_all, ks0 + (p + 1) * KS_SET, wave, lane), ws[(p + 1) & 1]);
        else load_wset(nxt, ws[0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KM) compute_set_km<AR>(T, p, lane, ws[p & 1], acc);
        else compute_set_tm<AR>(T, p, lrow, lhalf, ws[p & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// LayerNorm over the 256 features of the 64 tokens in accumulator layout (rows = this wave's 32 features, lane = token) -> T
// (fp32, token-major); two-pass statistics like every fp32 LayerNorm of the engine (gemm_common.h stage_a_tile).  Ends with a
// barrier (T complete); its first barrier also orders every earlier LDS read of the workgroup before the writes.
// KEEP: the normalised values also replace the accumulator contents (post-norm blocks: they are the next residual).
template <bool KEEP = false, int AR = AR_F32>
__device__ __forceinline__ void ln_to_tile(f32x16 (&acc)[2], float* P1, float* P2, const float* __restrict__ g,
                                           const float* __restrict__ bta, float eps, float* T, int valid, int wave, int lrow, int lhalf) {
    float mean[2], rstd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[mt][r];
        s += __shfl_xor(s, 32, 64);
        if (lhalf == 0) P1[wave * BM32 + mt * 32 + lrow] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += P1[w * BM32 + mt * 32 + lrow];
        mean[mt] = s * (1.0f / D);
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[mt][r] - mean[mt];
            v += d * d;
        }
        v += __shfl_xor(v, 32, 64);
        if (lhalf == 0) P2[wave * BM32 + mt * 32 + lrow] = v;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += P2[w * BM32 + mt * 32 + lrow];
        rstd[mt] = 1.0f / sqrtf(v * (1.0f / D) + eps);
    }
    const float* gp = g + wave * 32 + 4 * lhalf;
    const float* bp = bta + wave * 32 + 4 * lhalf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(gp + 8 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * q);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bool ok = mt * 32 + lrow < valid;          // rows beyond the read leave as zeros
            f32x4 y;
            y[0] = ok ? (acc[mt][4 * q + 0] - mean[mt]) * rstd[mt] * g4.x + b4.x : 0.f;
            y[1] = ok ? (acc[mt][4 * q + 1] - mean[mt]) * rstd[mt] * g4.y + b4.y : 0.f;
            y[2] = ok ? (acc[mt][4 * q + 2] - mean[mt]) * rstd[mt] * g4.z + b4.z : 0.f;
            y[3] = ok ? (acc[mt][4 * q + 3] - mean[mt]) * rstd[mt] * g4.w + b4.w : 0.f;
            if constexpr (KEEP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][4 * q + e] = y[e];
            }
            tile_store4<AR>(T, mt * 32 + lrow, wave * 32 + 8 * q + 4 * lhalf, y);
        }
    }
    __syncthreads();
}

}  // namespace

struct Tail32Args {
    const float* y;           // [B, 256, Lp] fp32
    float* h;                 // [B, L, 256] residual stream, updated in place
    const f32x4 *w_out, *w_fc1, *w_fc2, *w_in;   // packed (pack_f32t_kernel); w_in: the NEXT block's in_proj, or null
    const float *b_out, *b_fc1, *b_fc2, *b_in, *ln2_g, *ln2_b, *n_g, *n_b;
    float* z;                 // [B, 768, Lp] fp32 (NEXT)
    int B, L, Lp, tiles_x;
    float eps;
    const int* p0;            // [B] or null: 128-token tiles wholly inside each read's [PAD] prefix (pad_prefix.hip) -- not computed
    // NEXT = T32_SCORE (the last block): w_in = attention.0.weight (packed like the others), b_in = its bias, n_g / n_b = ln_f
    const float *att_w2, *att_b2;   // attention.2 weight [256] and bias [1]
    float* scores;            // [B, L] pooling scores
    float* partial;           // [B, tiles_x, POOL_PSTRIDE] per-tile online-softmax pooling partials (64-token tiles here)
    // block 0 with the id-table convolution: the residual entering the block is the embedding row of the token id -- gathered here,
    // the embedding kernel's 1 KiB per token never crosses HBM (as in the 16-bit tail kernel)
    const unsigned char* ids8;   // [B, Lp] or null
    const float* emb;            // [16, 256]
};

// NEXT: what follows the block on the tile still in registers -- nothing / the next block's LayerNorm-1 + in_proj / (round 5, the
// last block) ln_f + attention.0 + GELU(erf) + attention.2 = the pooling scores and this tile's online-softmax pooling partial, as
// the 16-bit tail kernel has it (gemm16.hip score_pool_tile; /root/reference/chimeralm/models/components/hyena.py:117-132): the
// separate score GEMM (3.5 ms per 256 x 8,193 launch), softmax-statistics and pooling kernels of rounds 1-4 read the fp32 residual
// stream twice more; head_tiles_kernel (head.hip) merges the partials in a fixed order.
enum { T32_NONE = 0, T32_INPROJ = 1, T32_SCORE = 2 };

template <int NEXT, int AR = AR_F32>
__global__ __launch_bounds__(512) void tail32_kernel(Tail32Args m) {
    constexpr float WS = WSCALE<AR>, WSI = WUNSCALE<AR>;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* As = smem32;                                     // LN2(r) / LN1'(h') tile [64][RS32]      (the y tile aliases As + Hs)
    float* Hs = As + BM32 * RS32;                           // gelu(fc1) chunk [64][RS32]
    float* Ys = smem32;                                     // y tile [256 channels][RSY], k-major
    float* P1 = smem32 + 2 * BM32 * RS32;                   // LayerNorm partials [8][64]
    float* P2 = P1 + 8 * BM32;
    static_assert(D * RSY <= 2 * BM32 * RS32 && 2 * D * RSKM64 * 2 <= 2 * BM32 * RS32 * 4, "the y tile fits As + Hs");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lrow = lane & 31, lhalf = lane >> 5;
    const int b = (int)blockIdx.x / m.tiles_x, t0 = ((int)blockIdx.x % m.tiles_x) * BM32, L = m.L, Lp = m.Lp;
    // a tile wholly inside the read's [PAD] prefix: its rows of z (and of the final h) come from the all-[PAD] table, nothing here
    // reads its neighbours (raw in_proj rows: the short filter runs in the convolution)  -- whole workgroup, before any barrier
    if (m.p0 && t0 + BM32 <= 128 * m.p0[b]) return;
    const int valid = L - t0 < BM32 ? L - t0 : BM32;
    f32x4 ws[2][KS_SET];
    f32x16 acc1[2], acc2[2];
    load_wset(wset_ptr(m.w_out, 0, D / 8, 0, wave, lane), ws[0]);
    // ---- 0. y tile -> LDS (k-major, as it lies in HBM): 16 lanes x 16 bytes per channel row, 32 channels per pass
    {
        const int tk = (tid & 15) * 4;
        const float* src = m.y + (size_t)b * D * Lp + t0 + tk;
        const bool in_row = t0 + tk < Lp;                   // (Lp is a multiple of 64: a 16-byte piece is inside the row or beyond it)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = (tid >> 4) + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_row) v = *reinterpret_cast<const f32x4*>(src + (size_t)c * Lp);
            if constexpr (AR == AR_X3) {                    // two planes of halfs [channel][RSKM64]
                h4_t hi, lo;
                split4(v, hi, lo);
                _Float16* yh = reinterpret_cast<_Float16*>(Ys) + c * RSKM64 + tk;
                *reinterpret_cast<h4_t*>(yh) = hi;
                *reinterpret_cast<h4_t*>(yh + D * RSKM64) = lo;
            } else {
                *reinterpret_cast<f32x4*>(Ys + c * RSY + tk) = v;
            }
        }
    }
    // ---- 1. r = h + b_out + W_out y: the accumulators start from the residual rows
    {
        const float* bo = m.b_out + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int t = t0 + mt * 32 + lrow, tc = t < L ? t : L - 1;
            const float* row = m.ids8 ? m.emb + (size_t)m.ids8[(size_t)b * Lp + tc] * D + wave * 32 + 4 * lhalf
                                      : m.h + ((size_t)b * L + tc) * D + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hv = *reinterpret_cast<const float4*>(row + 8 * q);
                const float4 bb = *reinterpret_cast<const float4*>(bo + 8 * q);
                acc2[mt][4 * q + 0] = (hv.x + bb.x) * WS;
                acc2[mt][4 * q + 1] = (hv.y + bb.y) * WS;
                acc2[mt][4 * q + 2] = (hv.z + bb.z) * WS;
                acc2[mt][4 * q + 3] = (hv.w + bb.w) * WS;
            }
        }
    }
    __syncthreads();
    product256<true, AR>(Ys, m.w_out, 0, D / 8, 0, wset_ptr(m.w_fc1, 0, D / 8, 0, wave, lane), wave, lane, ws, acc2);
    if constexpr (AR == AR_X3) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] *= WSI;
    }
    // ---- 2. LayerNorm-2 -> As
    ln_to_tile<false, AR>(acc2, P1, P2, m.ln2_g, m.ln2_b, m.eps, As, valid, wave, lrow, lhalf);
    if constexpr (AR == AR_X3) {                             // (the fc2 products land on r in the weights' scale)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] *= WS;
    }
    // ---- 3. MLP in four 256-wide chunks of the hidden layer (acc2 holds r)
#pragma unroll 1
    for (int j = 0; j < DI / 256; ++j) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
        product256<false, AR>(As, m.w_fc1, j, D / 8, 0, wset_ptr(m.w_fc2, 0, DI / 8, j * 32, wave, lane), wave, lane, ws, acc1);
        {
            const float* b1 = m.b_fc1 + j * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + 8 * q);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 g = {gelu_tanh(acc1[mt][4 * q + 0] * WSI + bb.x), gelu_tanh(acc1[mt][4 * q + 1] * WSI + bb.y),
                               gelu_tanh(acc1[mt][4 * q + 2] * WSI + bb.z), gelu_tanh(acc1[mt][4 * q + 3] * WSI + bb.w)};
                    tile_store4<AR>(Hs, mt * 32 + lrow, wave * 32 + 8 * q + 4 * lhalf, g);
                }
            }
        }
        __syncthreads();
        const f32x4* nxt = j + 1 < DI / 256 ? wset_ptr(m.w_fc1, j + 1, D / 8, 0, wave, lane)
                                            : wset_ptr(NEXT != T32_NONE ? m.w_in : m.w_fc1, 0, D / 8, 0, wave, lane);
        product256<false, AR>(Hs, m.w_fc2, 0, DI / 8, j * 32, nxt, wave, lane, ws, acc2);
        __syncthreads();                                    // every wave is done reading Hs before the next chunk lands in it
    }
    // ---- 4. h' = acc2 + b_2: 16 bytes per lane and feature quad (a lane = a token row of h)
    {
        const float* b2p = m.b_fc2 + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(b2p + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc2[mt][4 * q + 0] = acc2[mt][4 * q + 0] * WSI + bb.x;
                acc2[mt][4 * q + 1] = acc2[mt][4 * q + 1] * WSI + bb.y;
                acc2[mt][4 * q + 2] = acc2[mt][4 * q + 2] * WSI + bb.z;
                acc2[mt][4 * q + 3] = acc2[mt][4 * q + 3] * WSI + bb.w;
                if (mt * 32 + lrow < valid)
                    *reinterpret_cast<float4*>(m.h + ((size_t)b * L + t0 + mt * 32 + lrow) * D + wave * 32 + 4 * lhalf + 8 * q) =
                        make_float4(acc2[mt][4 * q + 0], acc2[mt][4 * q + 1], acc2[mt][4 * q + 2], acc2[mt][4 * q + 3]);
            }
        }
    }
    // ---- 5. the next block's LayerNorm-1 + in_proj on the tile still in registers: z rows x0 | x1 | v
    if constexpr (NEXT == T32_INPROJ) {
        ln_to_tile<false, AR>(acc2, P1, P2, m.n_g, m.n_b, m.eps, As, valid, wave, lrow, lhalf);
#pragma unroll 1
        for (int nb = 0; nb < D3 / 256; ++nb) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
            product256<false, AR>(As, m.w_in, nb, D / 8, 0, wset_ptr(m.w_in, nb + 1 < D3 / 256 ? nb + 1 : 0, D / 8, 0, wave, lane), wave,
                                  lane, ws, acc1);
            // rows = features: 32 lanes = 32 consecutive tokens of one channel row (128 bytes)
            const float* bi = m.b_in + nb * 256 + wave * 32 + 4 * lhalf;
            float* zb = m.z + ((size_t)b * D3 + nb * 256 + wave * 32 + 4 * lhalf) * Lp + t0 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2);
                const float bias = bi[f];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    if (mt * 32 + lrow < valid) zb[(size_t)f * Lp + mt * 32] = acc1[mt][r] * WSI + bias;
            }
        }
    }
    // ---- 5'. the last block: ln_f -> As, scores, and this tile's pooling partial  m = max s_t, S = sum exp(s_t - m),
    //          vec[c] = sum_t exp(s_t - m) ln_f(h')[t][c]   (softmax over ALL positions, pads included: hyena.py:121-132, mask None)
    if constexpr (NEXT == T32_SCORE) {
        ln_to_tile<false, AR>(acc2, P1, P2, m.n_g, m.n_b, m.eps, As, valid, wave, lrow, lhalf);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
        product256<false, AR>(As, m.w_in, 0, D / 8, 0, wset_ptr(m.w_in, 0, D / 8, 0, wave, lane), wave, lane, ws, acc1);   // (last slot: unused re-request)
        float* P = Hs;                                      // [8][64] score partials of the waves (Hs is dead: every wave is past the MLP)
        float* E = P + 8 * BM32;                            // [64] exp(s - m)
        float* V = E + BM32;                                // [4][256] pooled-vector partials of the four 16-token groups
        {
            const float* b1 = m.b_in + wave * 32 + 4 * lhalf;
            const float* w2 = m.att_w2 + wave * 32 + 4 * lhalf;
            float sc[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + 8 * q), ww = *reinterpret_cast<const float4*>(w2 + 8 * q);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    sc[mt] = fmaf(gelu_erf(acc1[mt][4 * q + 0] * WSI + bb.x), ww.x, sc[mt]);
                    sc[mt] = fmaf(gelu_erf(acc1[mt][4 * q + 1] * WSI + bb.y), ww.y, sc[mt]);
                    sc[mt] = fmaf(gelu_erf(acc1[mt][4 * q + 2] * WSI + bb.z), ww.z, sc[mt]);
                    sc[mt] = fmaf(gelu_erf(acc1[mt][4 * q + 3] * WSI + bb.w), ww.w, sc[mt]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float s2 = sc[mt] + __shfl_xor(sc[mt], 32, 64);
                if (lhalf == 0) P[wave * BM32 + mt * 32 + lrow] = s2;
            }
        }
        __syncthreads();
        const int tile = (int)blockIdx.x % m.tiles_x;
        float* pout = m.partial + ((size_t)b * m.tiles_x + tile) * POOL_PSTRIDE;
        if (wave == 0) {                                    // 64 tokens: one per lane; the waves' partials in a fixed order
            float s = (((P[lane] + P[BM32 + lane]) + (P[2 * BM32 + lane] + P[3 * BM32 + lane])) +
                       ((P[4 * BM32 + lane] + P[5 * BM32 + lane]) + (P[6 * BM32 + lane] + P[7 * BM32 + lane]))) + m.att_b2[0];
            if (lane < valid) m.scores[(size_t)b * L + t0 + lane] = s;
            else s = -INFINITY;
            const float mx = wave_max(s);                   // token t0 is always valid: mx is finite
            const float e = expf(s - mx);                   // exp(-inf) = 0 for the rows past L
            E[lane] = e;
            const float ssum = wave_sum(e);
            if (lane == 0) {
                pout[D] = mx;
                pout[D + 1] = ssum;
            }
        }
        __syncthreads();
        {   // vec: thread = (channel pair, 16-token group); the staged ln_f tile is read back (fp32 pairs, or hi + lo halfs of a pair)
            const int c2 = tid & 127, g = tid >> 7;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
            for (int i = 0; i < 16; ++i) {
                const int t = g * 16 + i;
                const float e = E[t];
                float v0, v1;
                if constexpr (AR == AR_X3) {
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                    const char* rowp = reinterpret_cast<const char*>(As + t * RS32) + c2 * 4;
                    const h2_t hi = *reinterpret_cast<const h2_t*>(rowp), lo = *reinterpret_cast<const h2_t*>(rowp + 512);
                    v0 = (float)hi[0] + (float)lo[0], v1 = (float)hi[1] + (float)lo[1];
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(As + t * RS32 + 2 * c2);
                    v0 = v.x, v1 = v.y;
                }
                a0 = fmaf(e, v0, a0);
                a1 = fmaf(e, v1, a1);
            }
            *reinterpret_cast<float2*>(V + g * D + 2 * c2) = make_float2(a0, a1);
        }
        __syncthreads();
        if (tid < D) pout[tid] = (V[tid] + V[D + tid]) + (V[2 * D + tid] + V[3 * D + tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SequenceCNNTransformer, exact fp32: one kernel per encoder layer for everything but the attention itself
//   /root/reference/chimeralm/models/components/transformer.py:64-68 (nn.TransformerEncoderLayer, post-norm, ReLU, no masks)
//       x1  = LN1(h + W_o att + b_o)
//       h'  = LN2(x1 + W_2 relu(W_1 x1 + b_1) + b_2)
//       qkv = W_qkv' h' + b_qkv'                       (the NEXT layer's in_proj; FIRST: only this, on the tile of h as it is)
// Same 64-token tile, weight packing and MFMA loop as tail32_kernel; activations are token-major in HBM ([M, 256] rows).
struct Enc32Args {
    const float* att;         // [M, 256]
    float* h;                 // [M, 256] residual stream (post-norm: normalised), updated in place
    const f32x4 *w_o, *w1, *w2, *w_qkv;
    const float *b_o, *b1, *b2, *b_qkv, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* qkv;               // [M, 768]
    size_t M;
    float eps;
};

template <int AR = AR_F32>
__device__ __forceinline__ void stage_rows32(const float* __restrict__ src, size_t row0, size_t M, float* T, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wave + 8 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < M) v = *reinterpret_cast<const f32x4*>(src + (row0 + r) * D + lane * 4);
        tile_store4<AR>(T, r, lane * 4, v);
    }
}

template <bool FIRST, int AR = AR_F32>
__global__ __launch_bounds__(512) void enc32_kernel(Enc32Args m) {
    constexpr float WS = WSCALE<AR>, WSI = WUNSCALE<AR>;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* As = smem32;
    float* Hs = As + BM32 * RS32;
    float* P1 = smem32 + 2 * BM32 * RS32;
    float* P2 = P1 + 8 * BM32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lrow = lane & 31, lhalf = lane >> 5;
    const size_t row0 = (size_t)blockIdx.x * BM32;
    const int valid = m.M - row0 < (size_t)BM32 ? (int)(m.M - row0) : BM32;
    f32x4 ws[2][KS_SET];
    f32x16 acc1[2], acc2[2];
    if constexpr (FIRST) {
        load_wset(wset_ptr(m.w_qkv, 0, D / 8, 0, wave, lane), ws[0]);
        stage_rows32<AR>(m.h, row0, m.M, As, tid);
        __syncthreads();
    } else {
        load_wset(wset_ptr(m.w_o, 0, D / 8, 0, wave, lane), ws[0]);
        stage_rows32<AR>(m.att, row0, m.M, As, tid);
        {   // acc2 = h + b_o
            const float* bo = m.b_o + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const size_t r = row0 + mt * 32 + lrow, rc = r < m.M ? r : m.M - 1;
                const float* row = m.h + rc * D + wave * 32 + 4 * lhalf;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 hv = *reinterpret_cast<const float4*>(row + 8 * q);
                    const float4 bb = *reinterpret_cast<const float4*>(bo + 8 * q);
                    acc2[mt][4 * q + 0] = (hv.x + bb.x) * WS;
                    acc2[mt][4 * q + 1] = (hv.y + bb.y) * WS;
                    acc2[mt][4 * q + 2] = (hv.z + bb.z) * WS;
                    acc2[mt][4 * q + 3] = (hv.w + bb.w) * WS;
                }
            }
        }
        __syncthreads();
        product256<false, AR>(As, m.w_o, 0, D / 8, 0, wset_ptr(m.w1, 0, D / 8, 0, wave, lane), wave, lane, ws, acc2);
        if constexpr (AR == AR_X3) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] *= WSI;
        }
        ln_to_tile<true, AR>(acc2, P1, P2, m.ln1_g, m.ln1_b, m.eps, As, valid, wave, lrow, lhalf);   // As = x1, acc2 = x1
        if constexpr (AR == AR_X3) {                         // (the fc2 products land on x1 in the weights' scale)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] *= WS;
        }
#pragma unroll 1
        for (int j = 0; j < DI / 256; ++j) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
            product256<false, AR>(As, m.w1, j, D / 8, 0, wset_ptr(m.w2, 0, DI / 8, j * 32, wave, lane), wave, lane, ws, acc1);
            {
                const float* b1 = m.b1 + j * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = *reinterpret_cast<const float4*>(b1 + 8 * q);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        f32x4 g = {fmaxf(acc1[mt][4 * q + 0] * WSI + bb.x, 0.f), fmaxf(acc1[mt][4 * q + 1] * WSI + bb.y, 0.f),
                                   fmaxf(acc1[mt][4 * q + 2] * WSI + bb.z, 0.f), fmaxf(acc1[mt][4 * q + 3] * WSI + bb.w, 0.f)};
                        tile_store4<AR>(Hs, mt * 32 + lrow, wave * 32 + 8 * q + 4 * lhalf, g);
                    }
                }
            }
            __syncthreads();
            const f32x4* nxt = j + 1 < DI / 256 ? wset_ptr(m.w1, j + 1, D / 8, 0, wave, lane)
                                                : wset_ptr(m.w_qkv ? m.w_qkv : m.w1, 0, D / 8, 0, wave, lane);
            product256<false, AR>(Hs, m.w2, 0, DI / 8, j * 32, nxt, wave, lane, ws, acc2);
            __syncthreads();
        }
        {   // + b_2, LayerNorm-2 -> As and the accumulators; h' leaves as 16 bytes per lane and feature quad
            const float* b2p = m.b2 + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(b2p + 8 * q);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc2[mt][4 * q + 0] = acc2[mt][4 * q + 0] * WSI + bb.x;
                    acc2[mt][4 * q + 1] = acc2[mt][4 * q + 1] * WSI + bb.y;
                    acc2[mt][4 * q + 2] = acc2[mt][4 * q + 2] * WSI + bb.z;
                    acc2[mt][4 * q + 3] = acc2[mt][4 * q + 3] * WSI + bb.w;
                }
            }
        }
        ln_to_tile<true, AR>(acc2, P1, P2, m.ln2_g, m.ln2_b, m.eps, As, valid, wave, lrow, lhalf);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (mt * 32 + lrow < valid)
                    *reinterpret_cast<float4*>(m.h + (row0 + mt * 32 + lrow) * D + wave * 32 + 4 * lhalf + 8 * q) =
                        make_float4(acc2[mt][4 * q + 0], acc2[mt][4 * q + 1], acc2[mt][4 * q + 2], acc2[mt][4 * q + 3]);
        if (!m.w_qkv) return;                                  // (uniform: after the last layer)
    }
    // qkv = W_qkv As + b_qkv, rows = features: a lane writes 4 consecutive features of its token row
#pragma unroll 1
    for (int nb = 0; nb < D3 / 256; ++nb) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
        product256<false, AR>(As, m.w_qkv, nb, D / 8, 0, wset_ptr(m.w_qkv, nb + 1 < D3 / 256 ? nb + 1 : 0, D / 8, 0, wave, lane), wave, lane,
                              ws, acc1);
        const float* bq = m.b_qkv + nb * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bq + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (mt * 32 + lrow < valid)
                    *reinterpret_cast<float4*>(m.qkv + (row0 + mt * 32 + lrow) * D3 + nb * 256 + wave * 32 + 4 * lhalf + 8 * q) =
                        make_float4(acc1[mt][4 * q + 0] * WSI + bb.x, acc1[mt][4 * q + 1] * WSI + bb.y, acc1[mt][4 * q + 2] * WSI + bb.z,
                                    acc1[mt][4 * q + 3] * WSI + bb.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SequenceCNNTransformer, exact fp32: Conv1d(256 -> 256, k = 3, padding = 1) + ReLU + MaxPool1d(2) of the CNN stem
//   /root/reference/chimeralm/models/components/transformer.py:47-58
// as three 256-deep products over ONE staged tile: rows t0 - 1 .. t0 + 64 of the read (zero outside it) lie in LDS once, and tap dk
// reads them shifted by dk rows; weights [3][256][256] (tap-major, each tap in launch_pack_f32t's order).  The pooling pairs are
// adjacent lanes of the accumulator layout (lane = position): one lane exchange, then the even lanes store 16 bytes per feature quad.
struct Conv32Args {
    const float* x;           // [B, Lin, 256]
    const f32x4* w;           // [3] x packed [256, 256]
    const float* bias;
    float* out;               // [B, Lin / 2, 256]
    int B, Lin, tiles_x;
};

template <int AR = AR_F32>
__global__ __launch_bounds__(512) void conv32_kernel(Conv32Args m) {
    constexpr float WSI = WUNSCALE<AR>;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* Xs = smem32;                                     // [66][RS32]: row r = position t0 - 1 + r
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lrow = lane & 31, lhalf = lane >> 5;
    const int b = (int)blockIdx.x / m.tiles_x, t0 = ((int)blockIdx.x % m.tiles_x) * BM32, Lin = m.Lin, Lout = Lin / 2;
    f32x4 ws[2][KS_SET];
    f32x16 acc[2];
    load_wset(wset_ptr(m.w, 0, D / 8, 0, wave, lane), ws[0]);
    {
        const float* src = m.x + (size_t)b * Lin * D + lane * 4;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int r = wave + 8 * i, t = t0 - 1 + r;
            if (r < BM32 + 2) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t >= 0 && t < Lin) v = *reinterpret_cast<const f32x4*>(src + (size_t)t * D);
                tile_store4<AR>(Xs, r, lane * 4, v);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    __syncthreads();
    static_for<0, 3>([&](auto dkc) {
        constexpr int dk = decltype(dkc)::value;
        const f32x4* wdk = m.w + (size_t)dk * (D * D / 4);
        const f32x4* nxt = wset_ptr(dk < 2 ? m.w + (size_t)(dk + 1) * (D * D / 4) : m.w, 0, D / 8, 0, wave, lane);
        product256<false, AR>(Xs + dk * RS32, wdk, 0, D / 8, 0, nxt, wave, lane, ws, acc);
    });
    const float* bp = m.bias + wave * 32 + 4 * lhalf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(bp + 8 * q);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float v[4] = {fmaxf(acc[mt][4 * q + 0] * WSI + bb.x, 0.f), fmaxf(acc[mt][4 * q + 1] * WSI + bb.y, 0.f),
                          fmaxf(acc[mt][4 * q + 2] * WSI + bb.z, 0.f), fmaxf(acc[mt][4 * q + 3] * WSI + bb.w, 0.f)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));       // positions 2 p, 2 p + 1: adjacent lanes
            const int pos = (t0 + mt * 32 + lrow) >> 1;
            if (!(lrow & 1) && pos < Lout)
                *reinterpret_cast<float4*>(m.out + ((size_t)b * Lout + pos) * D + wave * 32 + 4 * lhalf + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// W [N][K] row-major fp32 -> [N / 256][8 waves][K / 8 k-steps][64 lanes] float4: lane (lrow, lhalf) of wave w holds
// W[nb * 256 + w * 32 + lrow][8 s + 4 lhalf + 0..3]
__global__ __launch_bounds__(256) void pack_f32t_kernel(const float* __restrict__ w, f32x4* __restrict__ out, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)N * K / 4;
    if (i >= total) return;
    const int lane = (int)(i & 63), lrow = lane & 31, lhalf = lane >> 5;
    const size_t rest = i >> 6;
    const int ksteps = K / 8, s = (int)(rest % ksteps), nw = (int)(rest / ksteps);       // nw = nb * 8 + wave
    const float* src = w + (size_t)(nw * 32 + lrow) * K + 8 * s + 4 * lhalf;
    out[i] = f32x4{src[0], src[1], src[2], src[3]};
}

// AR_X3: W [N][K] -> [N / 256][8 waves][K / 16 k-steps][hi | lo][64 lanes] x 8 halfs: lane (lrow, lhalf) of wave w holds
// fp16(1024 W[nb * 256 + w * 32 + lrow][16 s + 8 lhalf + 0..7]) and, in the next fragment, the fp16 of what that rounding left
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ w, f32x4* __restrict__ out, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)N * K / 4;     // fragments: N K / 8 hi + N K / 8 lo
    if (i >= total) return;
    const int lane = (int)(i & 63), lrow = lane & 31, lhalf = lane >> 5;
    const size_t rest = i >> 6;
    const int frags = K / 8, f = (int)(rest % frags), nw = (int)(rest / frags), s = f >> 1, plane = f & 1;
    const float* src = w + (size_t)(nw * 32 + lrow) * K + 16 * s + 8 * lhalf;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[j] * X3_WS;
        const _Float16 hi = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);       // (|w| > 64: saturating, like split4)
        o[j] = plane ? (_Float16)__builtin_amdgcn_fmed3f(x - (float)hi, -65504.f, 65504.f) : hi;
    }
    out[i] = __builtin_bit_cast(f32x4, o);
}

void launch_pack_x3(const float* w, void* out, int n, int k, hipStream_t st) {
    const size_t total = (size_t)n * k / 4;
    hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<f32x4*>(out), n, k);
}

void launch_pack_f32t(const float* w, void* out, int n, int k, hipStream_t st) {
    const size_t total = (size_t)n * k / 4;
    hipLaunchKernelGGL(pack_f32t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<f32x4*>(out), n, k);
}

// one instantiation per kernel: the dynamic-LDS attribute is set once per kernel and device
template <auto Kern, typename Args>
static void launch_lds(dim3 grid, dim3 block, size_t lds, hipStream_t st, const Args& m) {
    CLM_SET_LDS(Kern, lds);
    hipLaunchKernelGGL(Kern, grid, block, lds, st, m);
}
// x3: the products as three fp16 MFMAs on hi + lo halfs (weights from launch_pack_x3) instead of the fp32 MFMA.
// score != null (the last block; w_in_next must be null): ln_f + pooling scores + per-64-token-tile pooling partials follow on the tile.
void launch_tail32(const float* y, float* h, const void* w_out, const void* w_fc1, const void* w_fc2, const void* w_in_next,
                   const float* b_out, const float* b_fc1, const float* b_fc2, const float* b_in_next, const float* ln2_g,
                   const float* ln2_b, const float* n_g, const float* n_b, float* z, int B, int L, int Lp, float eps, hipStream_t st,
                   bool x3, const int* p0, const Tail32Score* score, const unsigned char* ids8, const float* emb) {
    Tail32Args m{y, h, reinterpret_cast<const f32x4*>(w_out), reinterpret_cast<const f32x4*>(w_fc1), reinterpret_cast<const f32x4*>(w_fc2),
                 reinterpret_cast<const f32x4*>(score ? score->w1 : w_in_next), b_out, b_fc1, b_fc2, score ? score->b1 : b_in_next, ln2_g, ln2_b,
                 score ? score->lnf_g : n_g, score ? score->lnf_b : n_b, z, B, L, Lp,
                 (L + BM32 - 1) / BM32, eps, p0, score ? score->w2 : nullptr, score ? score->b2 : nullptr, score ? score->scores : nullptr,
                 score ? score->partial : nullptr, ids8, emb};
    const size_t lds = (size_t)(2 * BM32 * RS32 + 2 * 8 * BM32) * sizeof(float);
    static_assert(8 * BM32 + BM32 + 4 * D <= BM32 * RS32, "score stage: P / E / V fit the Hs region");
    const dim3 grid((unsigned)(m.tiles_x * B));
    const dim3 block(512);
    if (x3) {
        if (score) launch_lds<tail32_kernel<T32_SCORE, AR_X3>>(grid, block, lds, st, m);
        else if (w_in_next) launch_lds<tail32_kernel<T32_INPROJ, AR_X3>>(grid, block, lds, st, m);
        else launch_lds<tail32_kernel<T32_NONE, AR_X3>>(grid, block, lds, st, m);
    } else {
        if (score) launch_lds<tail32_kernel<T32_SCORE, AR_F32>>(grid, block, lds, st, m);
        else if (w_in_next) launch_lds<tail32_kernel<T32_INPROJ, AR_F32>>(grid, block, lds, st, m);
        else launch_lds<tail32_kernel<T32_NONE, AR_F32>>(grid, block, lds, st, m);
    }
}

// FIRST (att == null): qkv of the tile of h as it is (the first layer's in_proj); otherwise the whole layer after its attention,
// and -- w_qkv != null -- the next layer's in_proj.  Weights in launch_pack_f32t's order.
void launch_enc32(const float* att, float* h, const void* w_o, const void* w1, const void* w2, const void* w_qkv, const float* b_o,
                  const float* b1, const float* b2, const float* b_qkv, const float* ln1_g, const float* ln1_b, const float* ln2_g,
                  const float* ln2_b, float* qkv, size_t M, float eps, hipStream_t st, bool x3) {
    Enc32Args m{att, h, reinterpret_cast<const f32x4*>(w_o), reinterpret_cast<const f32x4*>(w1), reinterpret_cast<const f32x4*>(w2),
                reinterpret_cast<const f32x4*>(w_qkv), b_o, b1, b2, b_qkv, ln1_g, ln1_b, ln2_g, ln2_b, qkv, M, eps};
    const size_t lds = (size_t)(2 * BM32 * RS32 + 2 * 8 * BM32) * sizeof(float);
    const dim3 grid((unsigned)((M + BM32 - 1) / BM32)), block(512);
    if (!att) {
        if (x3) launch_lds<enc32_kernel<true, AR_X3>>(grid, block, lds, st, m);
        else launch_lds<enc32_kernel<true, AR_F32>>(grid, block, lds, st, m);
    } else {
        if (x3) launch_lds<enc32_kernel<false, AR_X3>>(grid, block, lds, st, m);
        else launch_lds<enc32_kernel<false, AR_F32>>(grid, block, lds, st, m);
    }
}

// x [B, Lin, 256] -> relu(conv1d_k3(x) + bias) pooled by 2 -> out [B, Lin / 2, 256]; w: three taps, each packed by launch_pack_f32t
void launch_conv32(const float* x, const void* w, const float* bias, float* out, int B, int Lin, hipStream_t st, bool x3) {
    Conv32Args m{x, reinterpret_cast<const f32x4*>(w), bias, out, B, Lin, (Lin + BM32 - 1) / BM32};
    const size_t lds = (size_t)(BM32 + 2) * RS32 * sizeof(float);
    const dim3 grid((unsigned)(m.tiles_x * B)), block(512);
    if (x3) launch_lds<conv32_kernel<AR_X3>>(grid, block, lds, st, m);
    else launch_lds<conv32_kernel<AR_F32>>(grid, block, lds, st, m);
}

}  // namespace clm
