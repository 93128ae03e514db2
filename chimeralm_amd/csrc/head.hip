// head.hip -- HBM-bound ends of the path: embedding gather in front, attention pooling + classifier MLP behind.
//
// Reference arithmetic:
//   backbone.embeddings.word_embeddings  nn.Embedding(16, 256)                      SURVEY.md section 8(a) row 5
//   BinarySequenceClassifier.forward     /root/reference/chimeralm/models/components/hyena.py:117-146
//     a = softmax(scores, dim=1) over ALL L positions (pads included, mask is always None: hyena.py:256)
//     pooled = sum_L a * ln_f(h);  classifier (hyena.py:56-71), ResidualBlock (:160-180), output_layer (:74)
#include "chimeralm_hip.h"
#include "clm_common.h"

namespace clm {

// ---------------------------------------------------------------------------------------- embedding
template <typename IdT>
__global__ __launch_bounds__(256) void embed_kernel(const IdT* __restrict__ ids, int64_t row_stride,
                                                    const float* __restrict__ table, float* __restrict__ h,
                                                    unsigned char* __restrict__ ids8, int B, int L, int Lp,
                                                    int* __restrict__ bad_ids) {
    // one wave per token row: 64 lanes x float4 = 1 KiB
    const int lane = threadIdx.x & 63;
    const size_t tok = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (size_t)B * L) return;
    const int b = int(tok / L), t = int(tok % L);
    int id = (int)ids[(size_t)b * row_stride + t];
    // the reference's nn.Embedding(16, 256) raises IndexError for ids outside [0, 16); a kernel cannot raise: the id is
    // clamped (no wild read) and the handle's flag makes the NEXT API call fail with the message (clm_api.hip)
    if ((id < 0 || id >= VOCAB) && lane == 0 && bad_ids) *bad_ids = 1;
    id = id < 0 ? 0 : (id >= VOCAB ? VOCAB - 1 : id);
    if (ids8 && lane == 0) ids8[(size_t)b * Lp + t] = (unsigned char)id;   // compact copy for the block-0 conv
    if (h) {
        const float4 v = *reinterpret_cast<const float4*>(table + (size_t)id * D + lane * 4);
        *reinterpret_cast<float4*>(h + tok * D + lane * 4) = v;
    }
}

// ids only (16-bit modes, block 0 reads the embedding through the id tables): one thread per token
template <typename IdT>
__global__ __launch_bounds__(256) void ids8_kernel(const IdT* __restrict__ ids, int64_t row_stride,
                                                   unsigned char* __restrict__ ids8, int B, int L, int Lp,
                                                   int* __restrict__ bad_ids) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * Lp) return;
    const int b = int(i / Lp), t = int(i % Lp);
    int id = t < L ? (int)ids[(size_t)b * row_stride + t] : 0;
    if ((id < 0 || id >= VOCAB) && bad_ids) *bad_ids = 1;      // see embed_kernel
    ids8[i] = (unsigned char)(id < 0 ? 0 : (id >= VOCAB ? VOCAB - 1 : id));
}

void launch_embed(const void* ids, int ids_dtype, int64_t row_stride, const float* table, float* h,
                  unsigned char* ids8, int B, int L, int Lp, hipStream_t st, int* bad_ids) {
    if (!h) {
        dim3 g1((unsigned)(((size_t)B * Lp + 255) / 256)), b1(256);
        if (ids_dtype == CLM_DT_I64)
            hipLaunchKernelGGL(ids8_kernel<int64_t>, g1, b1, 0, st, (const int64_t*)ids, row_stride, ids8, B, L, Lp, bad_ids);
        else if (ids_dtype == CLM_DT_I32)
            hipLaunchKernelGGL(ids8_kernel<int32_t>, g1, b1, 0, st, (const int32_t*)ids, row_stride, ids8, B, L, Lp, bad_ids);
        else
            hipLaunchKernelGGL(ids8_kernel<uint8_t>, g1, b1, 0, st, (const uint8_t*)ids, row_stride, ids8, B, L, Lp, bad_ids);
        return;
    }
    dim3 grid((unsigned)(((size_t)B * L + 3) / 4)), block(256);
    if (ids_dtype == CLM_DT_I64)
        hipLaunchKernelGGL(embed_kernel<int64_t>, grid, block, 0, st, (const int64_t*)ids, row_stride, table, h, ids8, B, L, Lp, bad_ids);
    else if (ids_dtype == CLM_DT_I32)
        hipLaunchKernelGGL(embed_kernel<int32_t>, grid, block, 0, st, (const int32_t*)ids, row_stride, table, h, ids8, B, L, Lp, bad_ids);
    else
        hipLaunchKernelGGL(embed_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)ids, row_stride, table, h, ids8, B, L, Lp, bad_ids);
}

// ---------------------------------------------------------------------------------------- softmax statistics
// stats[b] = (max_t s[b,t], sum_t exp(s[b,t] - max)); one workgroup per read, fixed reduction order.
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ scores,
                                                            float* __restrict__ stats, int L) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = scores + (size_t)b * L;
    float m = -INFINITY;
    for (int t = tid; t < L; t += 256) m = fmaxf(m, s[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < L; t += 256) sum += expf(s[t] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        stats[2 * b] = m;
        stats[2 * b + 1] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

void launch_softmax_stats(const float* scores, float* stats, int B, int L, hipStream_t st) {
    hipLaunchKernelGGL(softmax_stats_kernel, dim3(B), dim3(256), 0, st, scores, stats, L);
}

// ---------------------------------------------------------------------------------------- attention pooling
// partial[b][split][wave][256] = sum over this wave's tokens of softmax weight * ln_f(h[b,t,:]).
// One wave per token (lane = 4 channels), LayerNorm recomputed in fp32 from the residual stream.
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                   const float* __restrict__ bta, const float* __restrict__ scores,
                                                   const float* __restrict__ stats, float* __restrict__ partial,
                                                   int L, float eps) {
    const int b = blockIdx.y, split = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float m = stats[2 * b], inv = 1.0f / stats[2 * b + 1];
    const float4 g4 = *reinterpret_cast<const float4*>(g + lane * 4);
    const float4 b4 = *reinterpret_cast<const float4*>(bta + lane * 4);
    const int per = (L + POOL_SPLIT - 1) / POOL_SPLIT;
    const int tbeg = split * per, tend = min(L, tbeg + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int t = tbeg + wave; t < tend; t += 4) {
        const float wgt = expf(scores[(size_t)b * L + t] - m) * inv;
        const float4 x = *reinterpret_cast<const float4*>(h + ((size_t)b * L + t) * D + lane * 4);
        const float mean = wave_sum((x.x + x.y) + (x.z + x.w)) * (1.0f / D);
        const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
        const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
        const float rstd = 1.0f / sqrtf(var + eps);
        a0 += wgt * (d0 * rstd * g4.x + b4.x);
        a1 += wgt * (d1 * rstd * g4.y + b4.y);
        a2 += wgt * (d2 * rstd * g4.z + b4.z);
        a3 += wgt * (d3 * rstd * g4.w + b4.w);
    }
    float* out = partial + (((size_t)b * POOL_SPLIT + split) * 4 + wave) * D + lane * 4;
    *reinterpret_cast<float4*>(out) = make_float4(a0, a1, a2, a3);
}

void launch_pool(const float* h, const float* g, const float* b, const float* scores, const float* stats,
                 float* partial, int B, int L, float eps, hipStream_t st) {
    hipLaunchKernelGGL(pool_kernel, dim3(POOL_SPLIT, B), dim3(256), 0, st, h, g, b, scores, stats, partial, L, eps);
}

// ---------------------------------------------------------------------------------------- classifier MLP
// One workgroup per read.  Weights are pre-transposed to [in][out] so thread o streams column o coalesced and
// sums over the inputs in a fixed order.
template <int IN, int OUT, bool GELU>
__device__ __forceinline__ void dense_layer(const float* __restrict__ wt, const float* __restrict__ bias,
                                            const float* xin, float* xout, const float* resid) {
    for (int o = threadIdx.x; o < OUT; o += 256) {
        float acc = bias[o];
        for (int i = 0; i < IN; ++i) acc = fmaf(wt[(size_t)i * OUT + o], xin[i], acc);
        if (GELU) acc = gelu_erf(acc);
        if (resid) acc += resid[o];
        xout[o] = acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void head_mlp_kernel(const float* __restrict__ partial, HeadW hw,
                                                       float* __restrict__ pooled_out, float* __restrict__ logits) {
    __shared__ float x0[D], x1[HH], x2[HH], x3[HH];
    const int b = blockIdx.x, tid = threadIdx.x;
    {   // fixed-order combine of the pooling partials: splits outer, waves inner
        float acc = 0.f;
        const float* p = partial + (size_t)b * POOL_SPLIT * 4 * D + tid;
        for (int s = 0; s < POOL_SPLIT * 4; ++s) acc += p[(size_t)s * D];
        x0[tid] = acc;
        pooled_out[(size_t)b * D + tid] = acc;
    }
    __syncthreads();
    dense_layer<D, HH, true>(hw.w0t, hw.b0, x0, x1, nullptr);      // classifier.0 + GELU
    dense_layer<HH, HH, true>(hw.w3t, hw.b3, x1, x2, nullptr);     // classifier.3 + GELU
    dense_layer<HH, HH, true>(hw.w60t, hw.b60, x2, x3, nullptr);   // ResidualBlock.layers.0 + GELU
    dense_layer<HH, HH, false>(hw.w63t, hw.b63, x3, x1, x2);       // ResidualBlock.layers.3 + residual
    if (tid < NCLS) {
        float acc = hw.bo[tid];
        for (int i = 0; i < HH; ++i) acc = fmaf(hw.wot[(size_t)i * NCLS + tid], x1[i], acc);
        logits[(size_t)b * NCLS + tid] = acc;
    }
}

void launch_head_mlp(const float* partial, const HeadW& hw, float* pooled_out, float* logits, int B, hipStream_t st) {
    hipLaunchKernelGGL(head_mlp_kernel, dim3(B), dim3(256), 0, st, partial, hw, pooled_out, logits);
}

// ---------------------------------------------------------------------------------------- 16-bit modes: tiles -> logits
// Merge of the per-tile online-softmax partials written by score_pool16_kernel, then the classifier for HR reads per
// workgroup.  The classifier is a chain of four small matrix-vector products whose cost is the latency of streaming
// the weights from L2, so: every weight element is fetched once per HR reads, each dot product is split in two halves
// (1024 threads = 512 outputs x 2), and the weight stream runs 16 elements ahead of the FMAs in a register ping-pong.
constexpr int HR = 4, HT = 1024, MAXTILES = (32770 + 63) / 64;     // (the exact path's partials are per 64-token tile: tail32.hip)

template <int IN, bool GELU>
__device__ __forceinline__ void dense_rows(const float* __restrict__ wt, const float* __restrict__ bias,
                                           const float (*xin)[HH], float (*xout)[HH], const float (*resid)[HH],
                                           float (*part)[HH]) {
    constexpr int HALF = IN / 2, PF = 16;
    static_assert(HALF % (2 * PF) == 0, "two prefetch sets per loop trip");
    const int o = threadIdx.x & (HH - 1), kh = threadIdx.x >> 9;
    const float* w = wt + (size_t)kh * HALF * HH + o;
    float acc[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) acc[r] = 0.f;
    float wa[PF], wb[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) wa[j] = w[(size_t)j * HH];
#pragma unroll 1
    for (int i0 = 0; i0 < HALF; i0 += 2 * PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) wb[j] = w[(size_t)(i0 + PF + j) * HH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < PF; j += 4)
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(&xin[r][kh * HALF + i0 + j]);
                acc[r] = fmaf(wa[j + 3], x.w, fmaf(wa[j + 2], x.z, fmaf(wa[j + 1], x.y, fmaf(wa[j], x.x, acc[r]))));
            }
        __builtin_amdgcn_sched_barrier(0);
        const int nx = i0 + 2 * PF < HALF ? i0 + 2 * PF : 0;     // wrap-around keeps the prefetch unconditional
#pragma unroll
        for (int j = 0; j < PF; ++j) wa[j] = w[(size_t)(nx + j) * HH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < PF; j += 4)
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(&xin[r][kh * HALF + i0 + PF + j]);
                acc[r] = fmaf(wb[j + 3], x.w, fmaf(wb[j + 2], x.z, fmaf(wb[j + 1], x.y, fmaf(wb[j], x.x, acc[r]))));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < HR; ++r) part[r][o] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
        const float bo = bias[o];
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            float v = (acc[r] + part[r][o]) + bo;
            if (GELU) v = gelu_erf(v);
            if (resid) v += resid[r][o];
            xout[r][o] = v;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(HT) void head_tiles_kernel(const float* __restrict__ partial, int ntiles, HeadW hw,
                                                        float* __restrict__ pooled_out, float* __restrict__ logits,
                                                        int B) {
    __shared__ __attribute__((aligned(16))) float x0[HR][HH], x1[HR][HH], x2[HR][HH], x3[HR][HH], part[HR][HH];
    __shared__ float wl[HR][MAXTILES + 3], sl[HR][MAXTILES + 3], red[HR][4];
    const int tid = threadIdx.x, b0 = blockIdx.x * HR;
    {   // 1024 threads = HR reads x 256 channels
        const int r = tid >> 8, c = tid & 255, b = b0 + r, bc = b < B ? b : B - 1, wv = c >> 6, lane = c & 63;
        const float* p = partial + (size_t)bc * ntiles * POOL_PSTRIDE;
        float mx = -INFINITY;
        for (int i = c; i < ntiles; i += 256) {
            const float2 ms = *reinterpret_cast<const float2*>(p + (size_t)i * POOL_PSTRIDE + D);
            wl[r][i] = ms.x;
            sl[r][i] = ms.y;
            mx = fmaxf(mx, ms.x);
        }
        mx = wave_max(mx);
        if (lane == 0) red[r][wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[r][0], red[r][1]), fmaxf(red[r][2], red[r][3]));
        for (int i = c; i < ntiles; i += 256) wl[r][i] = expf(wl[r][i] - mx);
        __syncthreads();
        float ssum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int i = 0; i < ntiles; ++i) ssum = fmaf(sl[r][i], wl[r][i], ssum);      // fixed order: tile 0 .. ntiles-1
        int i = 0;
        for (; i + 4 <= ntiles; i += 4) {                                             // four loads in flight, fixed order
            const float v0 = p[(size_t)i * POOL_PSTRIDE + c], v1 = p[(size_t)(i + 1) * POOL_PSTRIDE + c],
                        v2 = p[(size_t)(i + 2) * POOL_PSTRIDE + c], v3 = p[(size_t)(i + 3) * POOL_PSTRIDE + c];
            a0 = fmaf(v0, wl[r][i], a0);
            a1 = fmaf(v1, wl[r][i + 1], a1);
            a2 = fmaf(v2, wl[r][i + 2], a2);
            a3 = fmaf(v3, wl[r][i + 3], a3);
        }
        for (; i < ntiles; ++i) a0 = fmaf(p[(size_t)i * POOL_PSTRIDE + c], wl[r][i], a0);
        const float pooled = ((a0 + a1) + (a2 + a3)) / ssum;
        if (b < B) pooled_out[(size_t)b * D + c] = pooled;
        x0[r][c] = pooled;
    }
    __syncthreads();
    dense_rows<D, true>(hw.w0t, hw.b0, x0, x1, nullptr, part);       // classifier.0 + GELU
    dense_rows<HH, true>(hw.w3t, hw.b3, x1, x2, nullptr, part);      // classifier.3 + GELU
    dense_rows<HH, true>(hw.w60t, hw.b60, x2, x3, nullptr, part);    // ResidualBlock.layers.0 + GELU
    dense_rows<HH, false>(hw.w63t, hw.b63, x3, x1, x2, part);        // ResidualBlock.layers.3 + residual
    if (tid < HR * NCLS) {
        const int r = tid / NCLS, k = tid % NCLS;
        if (b0 + r < B) {
            float acc = hw.bo[k];
            for (int i = 0; i < HH; ++i) acc = fmaf(hw.wot[(size_t)i * NCLS + k], x1[r][i], acc);
            logits[(size_t)(b0 + r) * NCLS + k] = acc;
        }
    }
}

void launch_head_tiles(const float* partial, int ntiles, const HeadW& hw, float* pooled_out, float* logits, int B,
                       hipStream_t st) {
    hipLaunchKernelGGL(head_tiles_kernel, dim3((B + HR - 1) / HR), dim3(HT), 0, st, partial, ntiles, hw, pooled_out,
                       logits, B);
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols) {
        int r = int(i / cols), c = int(i % cols);
        out[(size_t)c * rows + r] = in[i];
    }
}
void launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st) {
    size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, rows, cols);
}

}  // namespace clm
