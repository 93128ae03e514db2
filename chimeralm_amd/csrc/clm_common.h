// clm_common.h -- shared device helpers and the kernel launch prototypes of the ChimeraLM MI355X engine.
// gfx950 only: wave64, MFMA 32x32, 160 KiB LDS.  No CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <type_traits>

#include "clm_lab.h"

namespace clm {

// model constants fixed by the reference (chimeralm/models/lm.py:19-31)
constexpr int D = 256;          // d_model
constexpr int D3 = 768;         // in_proj width (x0 | x1 | v)
constexpr int DI = 1024;        // MLP inner width
constexpr int XCDS = 8;          // MI355X: 8 XCDs, each with its own L2; workgroup id % 8 picks the XCD (round-robin dispatch)
constexpr int NLAYER = 4;
constexpr int VOCAB = 16;
constexpr int FORDER = 64;      // implicit filter MLP width
constexpr int EMB = 5;          // positional embedding width
constexpr int HH = 512;         // head hidden
constexpr int NCLS = 2;

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using u16x8 = unsigned short __attribute__((ext_vector_type(8)));
using u16x4 = unsigned short __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- guarantees static register indexing
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- storage element types of the 16-bit activations -------------------------------------------------
struct bf16_t {
    unsigned short bits;
};
struct f16_t {
    unsigned short bits;
};

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16_t v) { return __uint_as_float(uint32_t(v.bits) << 16); }
__device__ __forceinline__ float to_float(f16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v.bits, 2);
    return float(h);
}
template <typename T>
__device__ __forceinline__ T from_float(float v);
template <>
__device__ __forceinline__ float from_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_float<bf16_t>(float v) {
    __bf16 b = __bf16(v);  // round to nearest even, NaN-preserving (v_cvt_pk_bf16_f32)
    bf16_t r;
    __builtin_memcpy(&r.bits, &b, 2);
    return r;
}
template <>
__device__ __forceinline__ f16_t from_float<f16_t>(float v) {
    _Float16 h = _Float16(v);
    f16_t r;
    __builtin_memcpy(&r.bits, &h, 2);
    return r;
}

// Wave-wide sum with DPP adds inside each 16-lane row (no LDS traffic, a few cycles of latency each) and four
// v_readlane for the row totals.  The previous __shfl_xor version lowers to six dependent ds_bpermute_b32 round
// trips (~120 cycles each); in the LayerNorm staging that chain was 26 % of the fused MLP kernel's time.
// Result is wave-uniform; the summation order is fixed (deterministic).
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// gelu(x, approximate="tanh") = x * sigmoid(2*sqrt(2/pi)*(x + 0.044715 x^3))      (HyenaMlp activation)
//                              = x / (1 + 2^(x * (A + B x^2))),  A = -2 sqrt(2/pi) log2(e),  B = 0.044715 A
// Saturates correctly for |x| -> inf (2^-inf = 0 -> x ; 2^+inf = inf -> rcp = 0 -> -0); v_exp_f32 / v_rcp_f32 are 1 ulp.
constexpr float GELU_A = -2.3022081985f, GELU_B = -0.10294324f;
__device__ __forceinline__ float gelu_tanh(float x) {
    float e = __builtin_amdgcn_exp2f(x * (GELU_A + GELU_B * (x * x)));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// two at a time: the element-wise part maps onto v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32
using f32x2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    if constexpr (lab::NOGELU) return x * 0.5f;
    f32x2 p = x * (GELU_A + GELU_B * (x * x));
    f32x2 d = {1.0f + __builtin_amdgcn_exp2f(p.x), 1.0f + __builtin_amdgcn_exp2f(p.y)};
    f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * r;
}
// exact gelu (erf form): nn.GELU() of the head (hyena.py:38,47,162)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

// ---- round 4: lo bytes of 16-bit activations (fp16c) -----------------------------------------------------------------------------
// x = fp16(x) + lo, |lo| <= 2^-11 |x|: one e5m2 byte carries lo to 3 significant bits, i.e. x to ~15 bits instead of fp16's 11.
// lo8 = e5m2(lo * LO2_SCALE); LO2_SCALE = 2^10 / 0.9155 -- the factor 1 / 0.9155 undoes the truncation bias of the weight bytes the
// lo bytes meet in the MFMA (gemm_common.h mfma_lo2); readers that want the VALUE multiply by LO2_INV and never see it.  The scaled
// lo cannot exceed 16384 * 1.1 < 57344 (e5m2's maximum) for any finite fp16(x): no saturation logic.
constexpr float LO8_TRUNC_GAIN_V = 1.0f / 0.9155f;
constexpr float LO2_SCALE = 1024.f * LO8_TRUNC_GAIN_V;
constexpr float LO2_INV = 1.0f / LO2_SCALE;
constexpr int LO2_E8M0 = 127 - 10;
// (x - fp16(x)) of four values -> four e5m2 bytes (round to nearest even), hi halfs returned through `h`
// (written so that hipcc emits, per pair, ONE v_cvt_pk_f16_f32 for the halfs and two mixed-precision FMAs on them -- the half is
//  an operand of v_fma_mix_f32, not converted back first: (x - h) S = fma(h, -S, x S); 12 instructions per four values where
//  the plain form took 20, and every LayerNorm tile, z row and y row of fp16c pays them per element)
__device__ __forceinline__ unsigned lo8_pack4(float x0, float x1, float x2, float x3, u16x4& h) {
    if constexpr (lab::NOLOPACK) {      // timing-only: the halfs alone
        h = u16x4{from_float<f16_t>(x0).bits, from_float<f16_t>(x1).bits, from_float<f16_t>(x2).bits, from_float<f16_t>(x3).bits};
        return 0u;
    }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_convertvector(f2{x0, x1}, h2), b = __builtin_convertvector(f2{x2, x3}, h2);
    // (elements to scalars FIRST: __builtin_bit_cast straight from a vector element reads element 0 with hipcc 7.2 -- HISTORY.md
    //  section 4.7; the first version of this function wrote half 0 into every odd slot, which tools/micro/mfma_lo2.cpp caught)
    const _Float16 h0 = a[0], h1 = a[1], h2_ = b[0], h3 = b[1];
    h = u16x4{__builtin_bit_cast(unsigned short, h0), __builtin_bit_cast(unsigned short, h1), __builtin_bit_cast(unsigned short, h2_),
              __builtin_bit_cast(unsigned short, h3)};
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(fmaf((float)h0, -LO2_SCALE, x0 * LO2_SCALE), fmaf((float)h1, -LO2_SCALE, x1 * LO2_SCALE), w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(fmaf((float)h2_, -LO2_SCALE, x2 * LO2_SCALE), fmaf((float)h3, -LO2_SCALE, x3 * LO2_SCALE), w, true);
    return (unsigned)w;
}
// four e5m2 lo bytes -> the four fp32 corrections (x - fp16(x), up to the lo's own rounding)
__device__ __forceinline__ void lo8_unpack4(unsigned w, float (&d)[4]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 a = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
    d[0] = a.x * LO2_INV, d[1] = a.y * LO2_INV, d[2] = b.x * LO2_INV, d[3] = b.y * LO2_INV;   // (folds into the caller's FMAs)
}

// developer switches: is `name` in the comma-separated environment variable CLM_DEBUG?  (clm_api.hip; A/B runs and tests only)
bool debug_flag(const char* name);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a C-ABI client may hold handles on several GPUs in
// one process, so every launch site keeps one bit per device instead of a once-per-process flag (a handle on a second
// device launched its > 64-KiB-LDS kernels without the attribute).  CLM_SET_LDS(kernel pointer, bytes) before the launch.
inline void ensure_lds_attr(const void* kern, size_t lds, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
    else std::fprintf(stderr, "chimeralm_hip: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed on device %d: %s\n", lds, dev, hipGetErrorString(e));
}
#define CLM_SET_LDS(kern, lds)                                                             \
    do {                                                                                   \
        static std::atomic<unsigned long long> clm_lds_done_{0};                           \
        ::clm::ensure_lds_attr(reinterpret_cast<const void*>(kern), (lds), clm_lds_done_); \
    } while (0)

// ---- kernel launchers (definitions in the .hip files); all are asynchronous on `st` --------------------
// PREC_F16C ("fp16c"): fp16 activations x weights held as hi + lo (in_proj, out_proj, score layer; the MLP weights are plain fp16
// since round 3: MLP_PREC, gemm_common.h), hi = fp16(w), lo = e4m3((w - hi) * 2^17): per 64-deep group
// four fp16 MFMAs with hi and ONE block-scaled K = 64 fp8 MFMA with lo (the activation fragments' upper bytes, i.e. the halfs
// truncated to e5m2, gathered in registers) into one fp32 accumulator.  The rounding of the packed weights is the one error of the fp16 mode that is coherent across tokens
// (every token sees the same perturbed matrix, so the attention pooling cannot average it out): tests/error_model.py
// attributes 1.0e-3 of the fp16 mode's 1.3e-3 logit error to it.  What is left is the fp16 rounding of the activation operands
// (measured 1.2e-4 .. 9.6e-4 in the logits); clm_selfcheck measures it on the loaded weights.
enum Prec { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_F16C = 3 };

struct LayerW {            // device pointers, fp32 unless noted
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const void *w_in, *w_out, *w_fc1, *w_fc2;   // packed MFMA-fragment order, compute dtype
    const float *b_in, *b_out, *b_fc1, *b_fc2;
    const float *short_w, *short_b;             // [768][3], [768]
    const float *filt_bias;                     // [256]  (D skip term)
    const void *t_in = nullptr, *t_out = nullptr, *t_fc1 = nullptr, *t_fc2 = nullptr;   // exact fp32: the fused tail's packing (tail32.hip)
};

// embedding gather: ids [B, L] (dtype code CLM_DT_*) -> h fp32 [B, L, 256]
void launch_embed(const void* ids, int ids_dtype, int64_t row_stride, const float* table, float* h,
                  unsigned char* ids8 /*[B][Lp] clamped ids, may be null*/, int B, int L, int Lp, hipStream_t st,
                  int* bad_ids = nullptr /*device-visible flag set when an id is outside [0, 16)*/);

// exact fp32, fused (tail32.hip): out_proj + LN2 + MLP + both residuals on h in place, then -- w_in_next != null -- the next
// block's LayerNorm-1 + in_proj into z (rows x0 | x1 | v, fp32), or -- score != null, the last block -- ln_f + the pooling scores and
// one online-softmax pooling partial per 64-token tile (T32_TILE; merged by launch_head_tiles).  Weights in launch_pack_f32t's order.
constexpr int T32_TILE = 64;
struct Tail32Score {
    const void* w1;                       // attention.0.weight, packed like the tail's other weights (f32t, or x3 halfs)
    const float *b1, *w2, *b2, *lnf_g, *lnf_b;
    float* scores;                        // [B, L]
    float* partial;                       // [B, ceil(L / T32_TILE), POOL_PSTRIDE]
};
void launch_tail32(const float* y, float* h, const void* w_out, const void* w_fc1, const void* w_fc2, const void* w_in_next,
                   const float* b_out, const float* b_fc1, const float* b_fc2, const float* b_in_next, const float* ln2_g,
                   const float* ln2_b, const float* n_g, const float* n_b, float* z, int B, int L, int Lp, float eps, hipStream_t st,
                   bool x3 = false /*three fp16 MFMAs on hi + lo halfs per product, weights from launch_pack_x3*/,
                   const int* p0 = nullptr /*[B]: tiles inside each read's [PAD] prefix, skipped (pad_prefix.hip)*/,
                   const Tail32Score* score = nullptr,
                   const unsigned char* ids8 = nullptr /*block 0: residual rows = embedding rows by token id ([B, Lp])*/, const float* emb = nullptr);
void launch_pack_x3(const float* w /*[n][k]*/, void* out /*n * k * 4 bytes*/, int n, int k, hipStream_t st);
void launch_pack_f32t(const float* w /*[n][k]*/, void* out /*n * k floats*/, int n, int k, hipStream_t st);
// SequenceCNNTransformer, exact fp32 (tail32.hip conv32_kernel): Conv1d(k = 3, padding = 1) + ReLU + MaxPool1d(2); w = three taps
// [dk][co][ci], each packed by launch_pack_f32t
void launch_conv32(const float* x, const void* w, const float* bias, float* out, int B, int Lin, hipStream_t st, bool x3 = false);
// SequenceCNNTransformer, exact fp32 (tail32.hip enc32_kernel): att == null: qkv of the rows of h as they are; otherwise one encoder
// layer after its attention (out_proj + LN1 + FFN + LN2 on h in place) and, w_qkv != null, the next layer's in_proj into qkv
void launch_enc32(const float* att, float* h, const void* w_o, const void* w1, const void* w2, const void* w_qkv, const float* b_o,
                  const float* b1, const float* b2, const float* b_qkv, const float* ln1_g, const float* ln1_b, const float* ln2_g,
                  const float* ln2_b, float* qkv, size_t M, float eps, hipStream_t st, bool x3 = false);

// GEMM family (gemm.hip).  `prec` selects compute dtype; T16 activations are bf16/f16 (or fp32 for PREC_F32).
// z  = in_proj(LN1(h))      -> channel-major [B, 768, Lp]
void launch_inproj(int prec, const float* h, const float* g, const float* b, const void* w, const float* bias, void* z,
                   int B, int L, int Lp, float eps, hipStream_t st);
// h += out_proj(y^T)        y channel-major [B, 256, Lp]
void launch_outproj(int prec, const void* y, const void* w, const float* bias, float* h, int B, int L, int Lp,
                    hipStream_t st);
// u  = gelu_tanh(fc1(LN2(h)))  token-major [B, L, 1024]
void launch_fc1(int prec, const float* h, const float* g, const float* b, const void* w, const float* bias, void* u,
                int B, int L, float eps, hipStream_t st);
// h += fc2(u)
void launch_fc2(int prec, const void* u, const void* w, const float* bias, float* h, int B, int L, hipStream_t st);
// scores[b,t] = w2 . gelu_erf(W1 LNf(h) + b1) + b2
void launch_score(int prec, const float* h, const float* g, const float* b, const void* w1, const float* b1,
                  const float* w2, const float* b2, float* scores, int B, int L, float eps, hipStream_t st);
// tuned 16-bit kernels (gemm16.hip); prec must be PREC_BF16 or PREC_F16
void launch_inproj16(int prec, const float* h, const float* g, const float* b, const void* w, const float* bias,
                     void* z, int B, int L, int Lp, float eps, hipStream_t st);
void launch_outproj16(int prec, const void* y, const void* w, const float* bias, float* h, int B, int L, int Lp,
                      hipStream_t st);
// h += fc2(gelu_tanh(fc1(LN2(h)))) in one kernel (hidden activations stay on chip)
void launch_mlp16(int prec, float* h, const float* g, const float* b, const void* w1, const float* b1, const void* w2,
                  const float* b2, int B, int L, float eps, hipStream_t st);
// second half of a block in one kernel: h = r + fc2(gelu(fc1(LN2(r)))), r = h + out_proj(y^T)
struct ScorePoolArgs {
    const float* h;           // [B, L, 256]
    const float *ln_g, *ln_b;
    const void* w1;           // packed attention.0.weight
    const float *b1, *w2, *b2;
    float* scores;            // [B, L]
    float* partial;           // [B, ntiles, POOL_PSTRIDE]: vec[256], m, S
    int B, L, ntiles;
    float eps;
};

struct TailArgs {
    const void* y;            // [B, 256, Lp] channel-major, 16-bit
    float* h;                 // residual stream [B, L, 256]
    const void *w_out, *w1, *w2;
    const float *b_out, *ln_g, *ln_b, *b1, *b2;
    int B, L, Lp;
    float eps;
    int Lmain;                   // tokens covered by tiles: L, or L - 1 when the lone last token is peeled off (lone_token.hip)
    const unsigned char* ids8;   // block 0 only (else null): the incoming residual row of token t is emb[ids8[b][t]],
    const float* emb;            // read from the 16-row table instead of h (the embedding kernel then never writes h)
    // NEXT_INPROJ: LayerNorm-1 + in_proj of the FOLLOWING block on the tile just produced (z written for its convolution)
    const void* n_w;
    const float *n_bias, *n_g, *n_b;
    void* n_z;
    // NEXT_SCORE (last block): ln_f + attention scores + pooling partials, see score_pool_tile
    ScorePoolArgs sp;
    // GATED hand-over (NEXT_INPROJ with zg != 0, gemm16.hip inproj_blocks_gated): the in_proj stage applies the next block's 3-tap
    // short filter and the x1 * v gate itself and writes TWO rows per channel into n_z -- row c = x0f, row 256 + c = g = x1f * vf
    // (rows 512.. unused) -- instead of x0 | x1 | v: a third less z traffic, and the convolution's phase A becomes a load.
    int zg;
    const float4* n_fir;      // [256][3] per channel c and row group q (x0, x1, v): {w0, w1, w2, cb}, cb = short_b + in_proj_b * (w0 + w1 + w2)
    float2* edge_bnd;         // [gridDim.x][2][768] raw in_proj rows (no bias): [w][0] = tokens 126, 127 of the tile before workgroup w's
                              // first tile, [w][1] = tokens 0, 1 of that first tile (launch_gated_patch recomputes those two tokens)
    float2* edge_read;        // [B][768] or null: raw rows of the last two tiled tokens of every read (the peeled lone token's history)
    // Round 4, PREC_F16C: the 16-bit tensors either side of the convolution carry one lo byte per element (lo8_pack4): y to ~15 bits
    // into out_proj (its lo plane is staged into a second LDS tile and enters through mfma_lo2), x0f / g to ~15 bits into the
    // convolution.  tests/error_model.py: the fp16 rounding of y and z was 20-60 % of the mode's logit error variance.
    const unsigned char* ylo; // [B][256][Lp] lo bytes of y, written by the convolution (null: y is plain fp16)
    int zlo;                  // zg only: the gated in_proj stage also writes the lo bytes of x0f | g into rows 512.. of n_z ([2][256][Lp])
    int mlp_lo;               // PREC_F16C: w1 / w2 are packed as hi + lo too (tail16_kernel MLPC) instead of plain fp16
    // Round 5 (pad_prefix.hip): the tiles this launch computes, as a device list -- tiles[0] = their number, tiles[1 + j] = entry j
    // (tile_entry / tile_b / tile_tx / tile_cont below), reads in order, tiles of a read ascending and contiguous from its first
    // computed tile on.  Tiles wholly inside a read's [PAD] prefix are not in it: their rows come from the all-[PAD] table.
    const int* tiles;
};
// entry of the tile list: read b (< 4096), 128-token tile tx of that read (< 2^19), cont = the entry before it is tile tx - 1 of the
// same read (then the short filter's two-token history travels from one to the other as before; a read's first computed tile
// starts without history -- right for tile 0, and for a later tile its first two tokens lie inside the prefix the table overwrites)
__host__ __device__ inline int tile_entry(int b, int tx, bool cont) { return b | (tx << 12) | (cont ? (int)0x80000000u : 0); }
__host__ __device__ inline int tile_b(int e) { return e & 0xfff; }
__host__ __device__ inline int tile_tx(int e) { return (e >> 12) & 0x7ffff; }
__host__ __device__ inline bool tile_cont(int e) { return e < 0; }
constexpr int TILE_LIST_MAX_READS = 4096;
constexpr int PAD_ID = 4;         // [PAD] of the reference's tokenizer (chimeralm/data/tokenizer.py:152-159 pads on the left with it)
// p0[b] = 128-token tiles wholly inside read b's leading run of [PAD] (0 for every read when `enabled` is 0), and the tile list of
// the reads' remaining tiles -- each read from one tile BEFORE its first non-prefix tile on (that tile hands the gated in_proj stage
// its history; what it writes lies inside the prefix and is overwritten from the table)
void launch_pad_tiles(const unsigned char* ids8, int B, int Lp, int Lmain, int enabled, int* p0, int* tiles, hipStream_t st);
// rows [0, 128 p0[b]) of every read's z block (nrow16 element rows of `es` bytes + nlo byte rows behind 2 D element rows, as the
// gated hand-over lays them out) <- the same rows of the all-[PAD] table (one read, row pitch LpT)
void launch_prefix_fill_z(const int* p0, void* z, const void* table, int B, int Lp, int LpT, int Lmain, int es, int nrow16, int nlo, hipStream_t st,
                          int seg_skip_S = 0 /*> 1: the convolution that reads z skips prefix segments (SegPrefix, S segments): the rows
                                              of the segments it will not read are not copied either*/,
                          const int* partner = nullptr /*[B] the other read of each read's pair (-1: none); null: b ^ 1*/);
// pairs of the segmented convolution by descending [PAD] prefix (stable): perm[r] = the read of rank r, partner[b] = the other read of
// b's pair or -1
void launch_pair_order(const int* p0, int B, int* perm, int* partner, hipStream_t st);
// the same for the last block's products: pooling scores [B][L] and per-tile pooling partials [B][ntiles][POOL_PSTRIDE]
void launch_prefix_fill_pool(const int* p0, float* scores, float* partial, const float* t_scores, const float* t_partial, int B, int L,
                             int ntiles, int Lmain, hipStream_t st, int per128 = 1 /*partials per 128 tokens: 2 for tail32's 64-token tiles*/);
// ... and for the unfused exact path, whose pooling kernels read the residual stream itself: rows [0, 128 p0[b]) of h [B][L][256]
void launch_prefix_fill_h(const int* p0, float* h, const float* t_h, int B, int L, int Lmain, hipStream_t st);
// tiles of the tail kernel are taken in CONTIGUOUS ranges per workgroup (the short filter's two-token history then comes from the
// workgroup's own previous tile): range length for `total` tiles on `grid` workgroups
__host__ __device__ inline int tail_range_len(int total, int grid) { return (total + grid - 1) / grid; }
int tail16_grid(int total_tiles);   // workgroups launch_tail16 starts for that many tiles
// the first two tokens of every workgroup range that starts inside a read: recomputed from edge_bnd (gemm16.hip)
void launch_gated_patch(int prec, const TailArgs& m, hipStream_t st);
// n_fir of one layer from its short filter and in_proj bias
void launch_fir_table(const float* short_w /*[768][3]*/, const float* short_b, const float* in_bias, float4* fir /*[256][3]*/, hipStream_t st);
constexpr int NEXT_NONE = 0, NEXT_INPROJ = 1, NEXT_SCORE = 2;
// The last token of a read of 128 k + 1 tokens, through the second half of a block (and what follows) as fp32 matrix-vector
// products: see lone_token.hip.  All weights are the fp32 originals [out][in].
struct LoneTokenArgs {
    const void* y;               // [B, 256, Lp] channel-major, 16-bit: the convolution's output (this token's column is read)
    float* h;                    // residual stream [B, L, 256]: row L-1 read (unless ids8) and, if !last, written
    const unsigned char* ids8;   // block 0 of the id path: the incoming residual is emb[ids8[b][L-1]]
    const float* emb;
    const float *w_out, *b_out, *ln2_g, *ln2_b, *w_fc1, *b_fc1, *w_fc2, *b_fc2;
    int last;                    // 0: n_* = LayerNorm-1 + in_proj of the next block, z column written; 1: n_g / n_b = ln_f, score + partial
    const float *n_g, *n_b, *n_w, *n_bias;
    const float4* n_fir;         // gated hand-over (see TailArgs): n_z gets x0f / g at this token, history from edge_read
    const float2* edge_read;
    void* n_z;                   // [B, 768, Lp] 16-bit
    const float *att_w1, *att_b1, *att_w2, *att_b2;
    float *scores, *partial;     // [B, L]; [B, ntiles, POOL_PSTRIDE]
    float* ws;                   // fp32 scratch, lone_token_ws_floats(B) floats: vectors handed from stage to stage
    int B, L, Lp, ntiles;
    float eps;
    const unsigned char* ylo;    // round 4, PREC_F16C (see TailArgs): lo bytes of y (null: none); zlo: write the lo bytes of x0f | g too
    int zlo;
};
size_t lone_token_ws_floats(int B);
void launch_lone_token(int prec, const LoneTokenArgs& a, hipStream_t st);
// out_proj + LN2 + MLP (+ what follows on the same tile: NEXT_*), 16-bit modes (gemm16.hip)
void launch_tail16(int prec, const TailArgs& m, int next, hipStream_t st);
void tail16_dump_stamps();   // developer build only (CLM_DEBUG=stamp)
void conv_dump_stamps();
size_t packed_weight_bytes(int prec, int n, int k);
// pack W [n][k] fp32 (device) into MFMA fragment order of the compute dtype
void launch_pack_weight(int prec, const float* w, void* out, int n, int k, hipStream_t st);

// long convolution (hyena_conv.hip)
constexpr int SEG_LEN = 8192;                     // tokens per segment of the long-read path (half a 16384 transform)
int conv_logn_for(int L);                         // log2 of the single-shot FFT size for L tokens; <0 if L > 8193
int conv_segments_for(int L);                     // 1 = single shot; >1 = number of 8192-token segments
bool conv_lone_tail(int L);                       // long read of S*8192 + 1 tokens: the last output is a dot product (krev)
// krev [256][stride] = the layer's filter reversed in time, channel-major (launch_hyena_conv_seg, lone-tail lengths)
void launch_filter_reversed(const float* k /*[L][256]*/, const float* dskip, float* krev, int L, int stride, hipStream_t st);
void launch_filter(const float* z /*[maxlen][5]*/, const float* t /*[maxlen]*/, const float* w0, const float* b0,
                   const float* freq, const float* w2, const float* b2, const float* w4, const float* b4,
                   const float* w6, const float* deltas, float* k_out /*[L][256]*/, int L, hipStream_t st);
// spectrum of one layer's filter: kf [256][N] float2 = FFT_N(k[:, c]) / N (double precision inside);
// scratch: 256*N double2
// prev_off >= 0: taps [prev_off, prev_off + seg_len) additionally in the upper half of the block (long-read partitions)
void launch_filter_spectrum(const float* k /*[L][256]*/, const float* dskip /*[256], folded into tap 0*/, float2* kf,
                            double2* scratch, int L, int logn, int seg_off, int seg_len, int prev_off, hipStream_t st);
void launch_twiddles(float2* tw, int logn, hipStream_t st);   // tw[m] = exp(-2 pi i m / N), m < N/2
// y = ((causal_conv(v*x1, k) + D*(v*x1)) * x0)   with (x0,x1,v) = short_filter(z)    [B,256,Lp]; D lives in kf
// ids8/ztab non-null (16-bit modes, block 0): x0|x1|v come from the 16-row table ztab[id][768] via the token ids
// ids8 [B][Lp] instead of z (in_proj of block 0 is not launched)
// flags (A/B switches of the engine, CLM_DEBUG=conv_oneshot / conv_no_xcd at clm_create):
constexpr int CONV_ONESHOT = 1;   // 16384-point class, 16-bit: one workgroup per unit (hyena_conv_kernel) instead of the persistent kernel
constexpr int CONV_NO_XCD = 2;    // units in plain order instead of all read pairs of a channel on one XCD
constexpr int CONV_GATED = 4;     // z holds x0f (row c) and g = x1f * vf (row 256 + c), filtered and gated by the fused tail kernel
                                  // (TailArgs::zg); ignored for block 0's id-table path
void launch_hyena_conv(int prec, const void* z, void* y, const float2* kf, const float2* tw, const float* ktime,
                       const float* short_w, const float* short_b, int B, int L, int Lp, int logn,
                       const unsigned char* ids8, const float* ztab, hipStream_t st, int flags = 0,
                       const float2* kf_packed = nullptr /*16384-point class: kf through launch_spectrum_lanepack(kf, ., 1, 0);
                                                           without it the one-workgroup-per-unit kernel runs*/,
                       unsigned char* ylo = nullptr /*PREC_F16C, round 4: [B][256][Lp] lo bytes of y (lo8_pack4); with it the gated
                                                      rows of z are read as hi + lo too (rows 512.. of z: [2][256][Lp] bytes)*/);
// partition spectrum j (natural order, src [256][16384]) -> slice j of the lane-packed [256][KS][16][512] quads of the
// segmented kernel (hyena_conv_seg_kernel)
void launch_spectrum_lanepack(const float2* src, float2* dst, int KS, int j, hipStream_t st);
void launch_ztab(const float* emb, const float* g, const float* bta, const float* w, const float* bias, float* ztab,
                 float eps, hipStream_t st);

// Round 5, [PAD]-prefix reuse in the segmented convolution (pad_prefix.hip): segments wholly inside the [PAD] prefix of BOTH reads of
// a pair are not transformed -- the partition products read their spectra in the all-[PAD] table (pair form: launch_spectra_pair_form).
constexpr int SEG_DOT_THREADS = 512;                // threads of hyena_conv_seg_kernel = partial dot products per (channel, segment)
struct SegPrefix {
    const int* p0 = nullptr;          // [B] 128-token tiles wholly inside each read's [PAD] prefix; null: every segment is computed
    const float2* tab = nullptr;      // [256][tab_segs][N] the table's segment spectra in pair form (needed where p0 is given)
    int tab_segs = 0;
    const int* perm = nullptr;        // [B] or null: read perm[2 k] and perm[2 k + 1] form pair k (launch_pair_order: by descending prefix)
    const float* dots_in = nullptr;   // [256][dots_segs][SEG_DOT_THREADS]: the table's running per-thread sums of the last token's dot
                                      // product after each segment (reads of S * 8192 + 1 tokens; needed where p0 is given for them)
    float* dots_out = nullptr;        // the forward that fills a table: where those sums go (one read)
    int dots_segs = 0;
};
// long reads (L > 8193): partitioned convolution over S segments; kf [256][KS][N] (partition j built with prev_off = (j-1)*SEG_LEN),
// gscratch [pairs][256][S][N]
void launch_hyena_conv_seg(int prec, const void* z, void* y, const float2* kf, int KS /*partitions stored per channel >= S*/,
                           const float2* tw, const float* short_w,
                           const float* short_b, float2* gscratch, int B, int L, int Lp, int S,
                           const float* krev /*null unless conv_lone_tail(L)*/, int krev_stride,
                           const unsigned char* ids8 /*16-bit modes, block 0: as launch_hyena_conv*/, const float* ztab,
                           hipStream_t st, int flags = 0, unsigned char* ylo = nullptr /*as launch_hyena_conv*/,
                           const SegPrefix& pfx = SegPrefix{});
// table [256][S_T][N] (the scratch of the table's one read: G) -> (1 + i) G in place, the spectrum of a PAIR of such reads
void launch_spectra_pair_form(float2* table, int S_T, hipStream_t st);

// head (head.hip)
void launch_softmax_stats(const float* scores, float* stats /*[B][2] = max, sum*/, int B, int L, hipStream_t st);
constexpr int POOL_SPLIT = 16;
void launch_pool(const float* h, const float* g, const float* b, const float* scores, const float* stats,
                 float* partial /*[B][POOL_SPLIT][4][256]*/, int B, int L, float eps, hipStream_t st);
struct HeadW {  // transposed [in][out] fp32
    const float *w0t, *b0, *w3t, *b3, *w60t, *b60, *w63t, *b63, *wot, *bo;
};
void launch_head_mlp(const float* partial, const HeadW& hw, float* pooled_out, float* logits, int B, hipStream_t st);
// 16-bit modes: score + per-tile online-softmax pooling partials in one pass over h (gemm16.hip), merged by the head kernel
constexpr int POOL_PSTRIDE = 264;   // floats per tile partial: vec[256], max, sum, pad
void launch_score_pool16(int prec, const float* h, const float* g, const float* bta, const void* w1, const float* b1,
                         const float* w2, const float* b2, float* scores, float* partial /*[B][ntiles][POOL_PSTRIDE]*/,
                         int B, int L, float eps, hipStream_t st);
void launch_head_tiles(const float* partial, int ntiles, const HeadW& hw, float* pooled_out, float* logits, int B,
                       hipStream_t st);
void launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st);

}  // namespace clm
