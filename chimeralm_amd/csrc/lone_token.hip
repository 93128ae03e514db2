// lone_token.hip -- the last token of reads whose length is a multiple of 128 plus one (8193 = 64 x 128 + 1, the [SEP] of every
// 8k-bp read; also 4097, 16385, 32769): a 128-token tile of its own in the tail kernel, i.e. a whole extra round of the
// persistent workgroups for one token per read (2,080 tiles on 256 CUs = 8.125 rounds -> 9 for a 32-read shard: +10 % of the
// tail kernel; 17 instead of 16.25 -> 16 rounds for a 64-read chunk).
//
// The Hyena block is causal (short filter, long convolution) and everything else in it is per token, so the LAST token of a
// read influences no other token of the backbone: its chain  r = h + out_proj(y);  h' = r + fc2(gelu(fc1(LN2(r))));
// z' = in_proj(LN1'(h'))  (or, after the last block, ln_f -> attention.0 -> GELU -> attention.2 = its pooling score) can be
// peeled off the tile kernels.  The convolution kernel already produces y at that token (it sees every earlier token), and
// needs z at that token for the next block -- written here.  The chain runs as fp32 matrix-vector products straight from the
// fp32 weights (exact: no 16-bit rounding at all for this token -- it may carry a large share of the attention pooling),
// between the tail kernel and the next convolution: ~1.6 M multiply-adds per read, weights served by L2.
// Reference arithmetic: HyenaDNA block / HyenaMlp and
// BinarySequenceClassifier attention scores (/root/reference/chimeralm/models/components/hyena.py:50-53,117-119).
#include "clm_common.h"

namespace clm {

namespace {

constexpr int LT_THREADS = 512, LT_WAVES = LT_THREADS / 64;

// out[n] = act(bias[n] + W[n, :] . x) (+ add[n]) for n < N, K = 64 * KV * 4: each wave takes rows n = wave, wave + 8, ...; a lane
// holds its 4 * KV consecutive-by-quad inputs in registers (x[j] = quad j: elements 4 * (lane + 64 j) .. + 3)
template <int KV, int ACT /*0 none, 1 gelu_tanh, 2 gelu_erf*/>
__device__ __forceinline__ void matvec(const float* __restrict__ W, const float* __restrict__ bias, const float4 (&x)[KV], int N,
                                       const float* add, float* out, int wave, int lane) {
    // RB rows per trip: all their loads are issued before the first reduction (a row at a time the loop is one L2 round trip
    // per row: 290 trips per wave, ~0.25 ms per launch)
    constexpr int K = 256 * KV, RB = KV == 1 ? 8 : 4;
    for (int n0 = wave * RB; n0 < N; n0 += LT_WAVES * RB) {      // N is a multiple of LT_WAVES * RB
        float4 w[RB][KV];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < KV; ++j) w[i][j] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + i) * K + 4 * (lane + 64 * j));
        float s[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < KV; ++j) acc += (w[i][j].x * x[j].x + w[i][j].y * x[j].y) + (w[i][j].z * x[j].z + w[i][j].w * x[j].w);
            s[i] = wave_sum(acc);
        }
        if (lane < RB) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i) v = lane == i ? s[i] : v;   // wave_sum results are wave-uniform: lane i takes row n0 + i
            const int n = n0 + lane;
            v += bias ? bias[n] : 0.f;
            if (ACT == 1) v = gelu_tanh(v);
            if (ACT == 2) v = gelu_erf(v);
            out[n] = add ? v + add[n] : v;
        }
    }
}

// LayerNorm of the 256 values in `src` (LDS), result as this lane's quad (elements 4 lane .. 4 lane + 3); every wave computes it
__device__ __forceinline__ float4 layer_norm_quad(const float* src, const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                  int lane) {
    const float4 v = *reinterpret_cast<const float4*>(src + 4 * lane);
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / D);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 g4 = *reinterpret_cast<const float4*>(g + 4 * lane), b4 = *reinterpret_cast<const float4*>(b + 4 * lane);
    return make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
}

// The chain as FOUR (last block: five) small launches, each spreading the rows of one matrix over gridDim.x workgroups per read
// (64 rows each): a workgroup's L2 -> CU bandwidth is what bounds a matrix-vector product (one workgroup per read doing the whole
// chain measured 65 us per launch -- more than the 75 us the peeled tile saves at 32 reads; split over 4-16 workgroups per
// stage the chain takes ~25 us incl. launch gaps).  Vectors between stages live in a small fp32 scratch (ws: r | u | h' per read).
enum { LS_OUT = 0, LS_FC1 = 1, LS_FC2 = 2, LS_NEXT = 3, LS_ATT = 4, LS_SCORE = 5, LS_NEXT_GATED = 6 };
constexpr int LT_ROWS = 64;                                       // rows per workgroup: 8 waves x 8 rows (fc2: two trips of 4)
constexpr int WS_R = 0, WS_U = D, WS_H = D + DI, WS_STRIDE = D + DI + D;

template <typename T, int STAGE>
__global__ __launch_bounds__(LT_THREADS) void lone_stage_kernel(LoneTokenArgs a) {
    __shared__ __attribute__((aligned(16))) float vin[DI];
    const int b = blockIdx.y, n0 = blockIdx.x * LT_ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = a.L - 1;
    float* ws = a.ws + (size_t)b * WS_STRIDE;
    if constexpr (STAGE == LS_OUT) {            // r = h + out_proj(y)
        if (tid < D) {
            float yv = to_float(reinterpret_cast<const T*>(a.y)[((size_t)b * D + tid) * a.Lp + t]);
            if (a.ylo) yv += __builtin_amdgcn_cvt_f32_bf8((int)a.ylo[((size_t)b * D + tid) * a.Lp + t], 0) * LO2_INV;   // (fp16c: hi + lo)
            vin[tid] = yv;
        }
        __syncthreads();
        const float4 x[1] = {*reinterpret_cast<const float4*>(vin + 4 * lane)};
        // incoming residual row: block 0 of the id path reads the embedding table, else the residual stream
        const float* hrow = a.ids8 ? a.emb + (size_t)a.ids8[(size_t)b * a.Lp + t] * D : a.h + ((size_t)b * a.L + t) * D;
        matvec<1, 0>(a.w_out + (size_t)n0 * D, a.b_out + n0, x, LT_ROWS, hrow + n0, ws + WS_R + n0, wave, lane);
    } else if constexpr (STAGE == LS_FC1) {     // u = gelu_tanh(fc1(LN2(r)))
        const float4 x[1] = {layer_norm_quad(ws + WS_R, a.ln2_g, a.ln2_b, a.eps, lane)};
        matvec<1, 1>(a.w_fc1 + (size_t)n0 * D, a.b_fc1 + n0, x, LT_ROWS, nullptr, ws + WS_U + n0, wave, lane);
    } else if constexpr (STAGE == LS_FC2) {     // h' = r + fc2(u)
        float4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const float4*>(ws + WS_U + 4 * (lane + 64 * j));
        matvec<4, 0>(a.w_fc2 + (size_t)n0 * DI, a.b_fc2 + n0, x, LT_ROWS, ws + WS_R + n0, vin, wave, lane);
        __syncthreads();
        if (tid < LT_ROWS) {
            ws[WS_H + n0 + tid] = vin[tid];
            if (!a.last) a.h[((size_t)b * a.L + t) * D + n0 + tid] = vin[tid];
        }
    } else if constexpr (STAGE == LS_NEXT) {    // z' = in_proj(LN1'(h')) of the next block, at this token's column
        const float4 x[1] = {layer_norm_quad(ws + WS_H, a.n_g, a.n_b, a.eps, lane)};
        matvec<1, 0>(a.n_w + (size_t)n0 * D, a.n_bias + n0, x, LT_ROWS, nullptr, vin, wave, lane);
        __syncthreads();
        if (tid < LT_ROWS) reinterpret_cast<T*>(a.n_z)[((size_t)b * D3 + n0 + tid) * a.Lp + t] = from_float<T>(vin[tid]);
    } else if constexpr (STAGE == LS_NEXT_GATED) {   // the gated hand-over (gemm16.hip inproj_blocks_gated) at this token: channels
        // n0 .. n0 + 63, rows x0 | x1 | v raw (no bias), the short filter's history = the raw rows of tokens t - 2, t - 1 the last
        // tile of the read left in edge_read; x0f -> row c, g = x1f * vf -> row 256 + c
        const float4 x[1] = {layer_norm_quad(ws + WS_H, a.n_g, a.n_b, a.eps, lane)};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            matvec<1, 0>(a.n_w + (size_t)(q * D + n0) * D, nullptr, x, LT_ROWS, nullptr, vin + q * LT_ROWS, wave, lane);
        __syncthreads();
        if (tid < LT_ROWS) {
            const int c = n0 + tid;
            float zf[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 f = a.n_fir[c * 3 + q];
                const float2 hist = a.edge_read[(size_t)b * D3 + q * D + c];
                zf[q] = fmaf(f.z, vin[q * LT_ROWS + tid], fmaf(f.y, hist.y, fmaf(f.x, hist.x, f.w)));
            }
            T* z = reinterpret_cast<T*>(a.n_z) + (size_t)b * D3 * a.Lp + t;
            const float gv = zf[1] * zf[2];
            z[(size_t)c * a.Lp] = from_float<T>(zf[0]);
            z[(size_t)(D + c) * a.Lp] = from_float<T>(gv);
            if (a.zlo) {                                    // fp16c: the lo bytes of x0f | g, rows 512.. of z as [2][256][Lp] bytes
                unsigned char* zl = reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(a.n_z) + ((size_t)b * D3 + 2 * D) * a.Lp) + t;
                u16x4 hb;
                const unsigned l = lo8_pack4(zf[0], gv, 0.f, 0.f, hb);
                zl[(size_t)c * a.Lp] = (unsigned char)(l & 0xffu);
                zl[(size_t)(D + c) * a.Lp] = (unsigned char)((l >> 8) & 0xffu);
            }
        }
    } else if constexpr (STAGE == LS_ATT) {     // ln_f; attention.0 + GELU(erf) -> ws.u[0..255]; ln_f row = this token's pooling vector
        const float4 x[1] = {layer_norm_quad(ws + WS_H, a.n_g, a.n_b, a.eps, lane)};
        if (blockIdx.x == 0 && wave == 0)
            *reinterpret_cast<float4*>(a.partial + ((size_t)b * a.ntiles + (a.ntiles - 1)) * POOL_PSTRIDE + 4 * lane) = x[0];
        matvec<1, 2>(a.att_w1 + (size_t)n0 * D, a.att_b1 + n0, x, LT_ROWS, nullptr, ws + WS_U + n0, wave, lane);
    } else {                                     // pooling score; a tile of one token: max = score, sum = exp(0) = 1
        if (wave == 0) {
            const float4 av = *reinterpret_cast<const float4*>(ws + WS_U + 4 * lane), w2 = *reinterpret_cast<const float4*>(a.att_w2 + 4 * lane);
            const float s = wave_sum((av.x * w2.x + av.y * w2.y) + (av.z * w2.z + av.w * w2.w)) + a.att_b2[0];
            if (lane == 0) {
                float* part = a.partial + ((size_t)b * a.ntiles + (a.ntiles - 1)) * POOL_PSTRIDE;
                a.scores[(size_t)b * a.L + t] = s;
                part[D] = s;
                part[D + 1] = 1.0f;
            }
        }
    }
}

template <typename T>
static void launch_lone_t(const LoneTokenArgs& a, hipStream_t st) {
    const dim3 blk(LT_THREADS);
    hipLaunchKernelGGL((lone_stage_kernel<T, LS_OUT>), dim3(D / LT_ROWS, a.B), blk, 0, st, a);
    hipLaunchKernelGGL((lone_stage_kernel<T, LS_FC1>), dim3(DI / LT_ROWS, a.B), blk, 0, st, a);
    hipLaunchKernelGGL((lone_stage_kernel<T, LS_FC2>), dim3(D / LT_ROWS, a.B), blk, 0, st, a);
    if (!a.last && a.n_fir) {
        hipLaunchKernelGGL((lone_stage_kernel<T, LS_NEXT_GATED>), dim3(D / LT_ROWS, a.B), blk, 0, st, a);
    } else if (!a.last) {
        hipLaunchKernelGGL((lone_stage_kernel<T, LS_NEXT>), dim3(D3 / LT_ROWS, a.B), blk, 0, st, a);
    } else {
        hipLaunchKernelGGL((lone_stage_kernel<T, LS_ATT>), dim3(D / LT_ROWS, a.B), blk, 0, st, a);
        hipLaunchKernelGGL((lone_stage_kernel<T, LS_SCORE>), dim3(1, a.B), dim3(64), 0, st, a);
    }
}

}  // namespace

size_t lone_token_ws_floats(int B) { return (size_t)B * WS_STRIDE; }

void launch_lone_token(int prec, const LoneTokenArgs& a, hipStream_t st) {
    if (prec == PREC_BF16) launch_lone_t<bf16_t>(a, st);
    else launch_lone_t<f16_t>(a, st);
}

}  // namespace clm
