// tf_fp32.hip -- SequenceCNNTransformer forward in the REFERENCE'S OWN PRECISION (fp32), the parity mode of the encoder net.
//
// Reference: /root/reference/chimeralm/models/components/transformer.py:28-104 (modules and forward; no masks anywhere).
// The 16-bit path (tf_model.hip, attention.hip) is the throughput path; its fp16 MFMA inputs leave 0.6-2.2e-2 on logits of
// magnitude 3-9, far from north_star's 1e-3.  This path computes every product exactly as fp32 x fp32 with fp32 accumulation:
//   * dense layers and the k = 3 convolutions: v_mfma_f32_32x32x2_f32 (the convolution as a K = 768 GEMM whose A rows are
//     gathered on the fly: x[t-1] | x[t] | x[t+1], zero outside the read -- Conv1d(padding=1));
//   * attention: one thread per query row, K / V tiles of 64 keys staged in LDS (read as broadcasts), online softmax,
//     fp32 throughout (8 heads of 32; scores scaled by 1/sqrt(32) like nn.MultiheadAttention);
//   * LayerNorm (post-norm: LN(x + sublayer(x))), positional encoding, ReLU / MaxPool1d(2) as plain fp32 kernels.
// It is a correctness reference on the GPU, not tuned: stages are separate kernels and activations live in HBM as fp32.
#include <string>
#include <utility>

#include "clm_common.h"

namespace clm {
namespace tf32 {

constexpr int TVOC = 12;

// x[b, t, :] = emb[id]   (ids8 already clamped to [0, 16); rows >= 12 do not exist in nn.Embedding(12, 256): clamp to 11)
__global__ __launch_bounds__(256) void embed_kernel(const unsigned char* __restrict__ ids8, int ids_stride,
                                                    const float* __restrict__ emb, float* __restrict__ x, int B, int L) {
    const size_t tok = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (size_t)B * L) return;
    const int b = int(tok / L), t = int(tok % L), lane = threadIdx.x & 63;
    const int id = ids8[(size_t)b * ids_stride + t];
    const float4 v = *reinterpret_cast<const float4*>(emb + (size_t)(id < TVOC ? id : TVOC - 1) * D + lane * 4);
    *reinterpret_cast<float4*>(x + tok * D + lane * 4) = v;
}

// C[m, n] = act(sum_k A(m, k) W(n, k) + bias[n]) (+ R[m, n]).  Tile 64 x 64 per workgroup, 4 waves of one 32 x 32 MFMA tile.
//   CONV3 = false: A(m, k) = A[m * lda + k],  W(n, k) = W[n * K + k]
//   CONV3 = true : rows are (read, position t) with Lrow positions per read, K = 3 * 256:
//                  A(m, dk * 256 + ci) = x[b, t + dk - 1, ci] (0 outside),  W(n, dk * 256 + ci) = w[n][ci][dk]  (Conv1d weight)
template <bool RELU, bool CONV3>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ R,
                                                   float* __restrict__ C, int ldc, size_t M, int N, int K, int Lrow) {
    constexpr int KC = 32, LS = KC + 1;
    __shared__ float As[64 * LS], Ws[64 * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mt = wave >> 1, nt = wave & 1;
    const size_t m0 = (size_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * 256, row = e >> 5, kk = e & 31, k = k0 + kk;
            const size_t m = m0 + row;
            float a = 0.f;
            if (m < M) {
                if (CONV3) {
                    const int dk = k >> 8, ci = k & 255;
                    const long t = (long)(m % (size_t)Lrow) + dk - 1;
                    if (t >= 0 && t < Lrow) a = A[(m + dk - 1) * (size_t)lda + ci];
                } else {
                    a = A[m * (size_t)lda + k];
                }
            }
            As[row * LS + kk] = a;
            const int n = n0 + row;
            float w = 0.f;
            if (n < N) w = CONV3 ? W[((size_t)n * D + (k & 255)) * 3 + (k >> 8)] : W[(size_t)n * K + k];
            Ws[row * LS + kk] = w;
        }
        __syncthreads();
        const float* ap = As + (mt * 32 + (lane & 31)) * LS + (lane >> 5);
        const float* wp = Ws + (nt * 32 + (lane & 31)) * LS + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], wp[2 * ks], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D map: column = lane & 31 (output feature), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (token)
    const int n = n0 + nt * 32 + (lane & 31);
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) {
                float v = acc[r] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                if (R) v += R[m * (size_t)ldc + n];
                C[m * (size_t)ldc + n] = v;
            }
        }
    }
}

// MaxPool1d(2, 2) over positions (the ReLU before it was applied by the GEMM): out[b, p, :] = max(in[b, 2p, :], in[b, 2p + 1, :])
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Lin, int Lout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;               // one float4 of an output row
    if (i >= (size_t)B * Lout * (D / 4)) return;
    const size_t row = i / (D / 4);
    const int c4 = int(i % (D / 4)), b = int(row / Lout), p = int(row % Lout);
    const float4 a = *reinterpret_cast<const float4*>(in + ((size_t)b * Lin + 2 * p) * D + c4 * 4);
    const float4 c = *reinterpret_cast<const float4*>(in + ((size_t)b * Lin + 2 * p + 1) * D + c4 * 4);
    *reinterpret_cast<float4*>(out + row * D + c4 * 4) = make_float4(fmaxf(a.x, c.x), fmaxf(a.y, c.y), fmaxf(a.z, c.z), fmaxf(a.w, c.w));
}

// out[m, :] = LayerNorm(in[m, :] (+ pe[m % Lpos, :])) * g + b; one wave per row, two-pass variance
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ in, const float* __restrict__ pe, const float* __restrict__ g,
                                                 const float* __restrict__ bta, float* __restrict__ out, size_t M, int Lpos, float eps) {
    const size_t m = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float4 x = *reinterpret_cast<const float4*>(in + m * D + lane * 4);
    if (pe) {
        const float4 p = *reinterpret_cast<const float4*>(pe + (m % (size_t)Lpos) * D + lane * 4);
        x = make_float4(x.x + p.x, x.y + p.y, x.z + p.z, x.w + p.w);
    }
    const float mean = wave_sum((x.x + x.y) + (x.z + x.w)) * (1.0f / D);
    const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
    const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 g4 = *reinterpret_cast<const float4*>(g + lane * 4), b4 = *reinterpret_cast<const float4*>(bta + lane * 4);
    *reinterpret_cast<float4*>(out + m * D + lane * 4) =
        make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
}

// softmax(q k^T / sqrt(32)) v per (read, head): thread = query position; keys in LDS tiles of 64
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L3) {
    __shared__ float Ks[64 * 32], Vs[64 * 32];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int qpos = blockIdx.x * 256 + tid;
    const bool active = qpos < L3;
    const float* base = qkv + (size_t)b * L3 * 768;
    float q[32], o[32];
    const float scale = 0.17677669529663687f;                              // 1 / sqrt(32)
    {
        const float* qp = base + (size_t)(active ? qpos : 0) * 768 + head * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + d);
            q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
        }
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    float mx = -INFINITY, sum = 0.f;
    for (int k0 = 0; k0 < L3; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                      // 64 keys x 8 float4 of K and of V
            const int e = tid + i * 256, j = e >> 3, d4 = e & 7;
            const int kp = k0 + j < L3 ? k0 + j : L3 - 1;
            const float* rowp = base + (size_t)kp * 768 + head * 32 + d4 * 4;
            *reinterpret_cast<float4*>(Ks + j * 32 + d4 * 4) = *reinterpret_cast<const float4*>(rowp + 256);
            *reinterpret_cast<float4*>(Vs + j * 32 + d4 * 4) = *reinterpret_cast<const float4*>(rowp + 512);
        }
        __syncthreads();
        const int nk = L3 - k0 < 64 ? L3 - k0 : 64;
        for (int j = 0; j < nk; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(q[d], Ks[j * 32 + d], s);
            const float nm = fmaxf(mx, s), corr = expf(mx - nm), p = expf(s - nm);   // exp(-inf) = 0 on the first key
            sum = sum * corr + p;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] = fmaf(p, Vs[j * 32 + d], o[d] * corr);
            mx = nm;
        }
    }
    if (active) {
        const float inv = 1.0f / sum;
        float* op = out + ((size_t)b * L3 + qpos) * D + head * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4)
            *reinterpret_cast<float4*>(op + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    }
}

// Round 4: the same attention on the fp32 MFMA (v_mfma_f32_32x32x2_f32), built like the 16-bit kernel (attention.hip): scores
// TRANSPOSED, S^T = K Q^T (rows = keys, column = the lane's query), so the softmax statistics are register reductions plus one
// exchange between the half-waves; P^T never leaves the registers -- accumulator register t of a 32-key block IS the B operand of
// MFMA step t of O^T = V^T P^T, with the A operand read from the V row of the key that register holds (the reduction index may be
// paired freely: lanes 0-31 feed key (t & 3) + 8 (t >> 2), lanes 32-63 that key + 4).  Per 64-key tile a wave issues 64 MFMAs of 64
// cycles next to ~900 cycles of softmax VALU: MFMA-bound, where the scalar kernel above spends 2 x 32 FMAs per (query, key).
// 128 queries per workgroup (4 waves), K / V tiles of 64 keys double-buffered: 38 KiB of LDS, four workgroups per CU.
constexpr int A32_QT = 128, A32_KT = 64, A32_KRS = 36, A32_VRS = 40;   // row strides (floats): 16 rows of a ds_read_b128 group on 16 bank
                                                                      // groups (K); rows 4 apart on banks + 32 (V, ds_read_b32)
__global__ __launch_bounds__(256, 4) void attention32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L) {
    using f32x4 = float __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float Ks[2][A32_KT * A32_KRS];
    __shared__ __attribute__((aligned(16))) float Vs[2][A32_KT * A32_VRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
    // workgroup -> (query tile, head, read): the query tiles of one (read, head) share an XCD's L2 (as attention.hip)
    const int ntq = (L + A32_QT - 1) / A32_QT;
    const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const int bh = (slot / ntq) * 8 + xcd, q0 = (slot % ntq) * A32_QT, h = bh & 7, b = bh >> 3;
    const float* base = qkv + (size_t)b * L * 768 + h * 32;            // row t: q at +0, k at +256, v at +512
    const float c = 1.4426950408889634f * 0.17677669529663687f;        // log2(e) / sqrt(32)
    f32x4 qf[4];                                                       // Q^T as B operand: d = 8 s + 4 hf + 0..3
    {
        const int q = q0 + wave * 32 + n;
        const float* qp = base + (size_t)(q < L ? q : L - 1) * 768 + 4 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f32x4*>(qp + 8 * s);
    }
    f32x4 kreg[2], vreg[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, j = e >> 3, d4 = e & 7;
            const int key = k0 + j < L ? k0 + j : L - 1;                // clamped; the score mask removes the clones
            const float* p = base + (size_t)key * 768 + 4 * d4;
            kreg[i] = *reinterpret_cast<const f32x4*>(p + 256);
            vreg[i] = *reinterpret_cast<const f32x4*>(p + 512);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, j = e >> 3, d4 = e & 7;
            *reinterpret_cast<f32x4*>(&Ks[buf][j * A32_KRS + 4 * d4]) = kreg[i];
            *reinterpret_cast<f32x4*>(&Vs[buf][j * A32_VRS + 4 * d4]) = vreg[i];
        }
    };
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int ntiles = (L + A32_KT - 1) / A32_KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * A32_KT;
        if (t + 1 < ntiles) load_tile(k0 + A32_KT);
        f32x16 s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
            const float* kp = &Ks[buf][(blk * 32 + n) * A32_KRS + 4 * hf];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * st);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[st][j], s[blk], 0, 0, 0);
            }
        }
        if (k0 + A32_KT > L) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= L) s[blk][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);                              // finite: key k0 is always valid
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);   // 0 on the first tile
        const float mc = m_new * c;
        float psum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[blk][r] = __builtin_amdgcn_exp2f(fmaf(s[blk][r], c, -mc));
                psum += s[blk][r];
            }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        // O^T += V^T P^T: step t2 of a block pairs register t2 of P^T with the V row of the key it holds
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const float* vp = &Vs[buf][(blk * 32 + 4 * hf) * A32_VRS + n];
#pragma unroll
            for (int t2 = 0; t2 < 16; ++t2)
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[((t2 & 3) + 8 * (t2 >> 2)) * A32_VRS], s[blk][t2], o, 0, 0, 0);
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1);                       // the other buffer was last read in trip t - 1
        __syncthreads();
    }
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    const int q = q0 + wave * 32 + n;
    if (q < L) {
        float* op = out + ((size_t)b * L + q) * D + h * 32 + 4 * hf;    // lane (query n, half hf) holds d = (r & 3) + 8 (r >> 2) + 4 hf
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(op + 8 * gq) = make_float4(o[4 * gq + 0] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
    }
}

// fp16x3: the same attention with every operand as two halfs (hi = fp16(x), lo = fp16(x - hi)) and three fp16 MFMAs per product,
// built like the 16-bit kernel (attention.hip): S^T = K Q^T with a lane owning one query; P^T stays in registers and, split into
// halfs, is the B operand of O^T = V^T P^T up to the fixed permutation of the key index that is applied to V's rows when the tile
// is staged; V^T fragments by transposing LDS reads.  q / k / v arrive as fp32 and are split on their way into registers / LDS.
constexpr int AX_QT = 128, AX_KT = 64, AX_KRS = 40, AX_VRS = 32;       // halfs: K rows 80 B, V rows 64 B (attention.hip's strides)
__device__ __forceinline__ int ax_v_row(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }
__global__ __launch_bounds__(256, 3) void attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L) {
    using f32x4 = float __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f8 __attribute__((ext_vector_type(8)));
    using v4i16 = short __attribute__((ext_vector_type(4)));
    typedef v4i16 __attribute__((address_space(3))) * lds_v4;
    __shared__ __attribute__((aligned(16))) _Float16 Ks[2][2][AX_KT * AX_KRS];       // [buffer][hi | lo]
    __shared__ __attribute__((aligned(16))) _Float16 Vs[2][2][AX_KT * AX_VRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
    const int ntq = (L + AX_QT - 1) / AX_QT;
    const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const int bh = (slot / ntq) * 8 + xcd, q0 = (slot % ntq) * AX_QT, h = bh & 7, b = bh >> 3;
    const float* base = qkv + (size_t)b * L * 768 + h * 32;
    const float c = 1.4426950408889634f * 0.17677669529663687f;        // log2(e) / sqrt(32)
    auto clamp8 = [](f8 v) {                                           // (beyond fp16's range: saturate, never inf -- tail32.hip split4)
        f8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
        return r;
    };
    auto split8 = [&](f8 v, h8& hi, h8& lo) {
        hi = __builtin_convertvector(clamp8(v), h8);
        lo = __builtin_convertvector(clamp8(v - __builtin_convertvector(hi, f8)), h8);
    };
    h8 qh[2], ql[2];                                                   // Q^T as B operand: d = 16 s + 8 hf + 0..7
    {
        const int q = q0 + wave * 32 + n;
        const float* qp = base + (size_t)(q < L ? q : L - 1) * 768 + 8 * hf;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s2), bq = *reinterpret_cast<const f32x4*>(qp + 16 * s2 + 4);
            split8(f8{a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]}, qh[s2], ql[s2]);
        }
    }
    // staging: thread -> (key row, 4-float piece) x 2 of the K and of the V tile
    f32x4 kreg[2], vreg[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, j = e >> 3, d4 = e & 7;
            const int key = k0 + j < L ? k0 + j : L - 1;
            const float* p = base + (size_t)key * 768 + 4 * d4;
            kreg[i] = *reinterpret_cast<const f32x4*>(p + 256);
            vreg[i] = *reinterpret_cast<const f32x4*>(p + 512);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, j = e >> 3, d4 = e & 7;
            auto clamp4 = [](f32x4 v) {
                f32x4 r;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) r[e2] = __builtin_amdgcn_fmed3f(v[e2], -65504.f, 65504.f);
                return r;
            };
            const h4 kh = __builtin_convertvector(clamp4(kreg[i]), h4), vh = __builtin_convertvector(clamp4(vreg[i]), h4);
            const h4 kl = __builtin_convertvector(clamp4(kreg[i] - __builtin_convertvector(kh, f32x4)), h4);
            const h4 vl = __builtin_convertvector(clamp4(vreg[i] - __builtin_convertvector(vh, f32x4)), h4);
            *reinterpret_cast<h4*>(&Ks[buf][0][j * AX_KRS + 4 * d4]) = kh;
            *reinterpret_cast<h4*>(&Ks[buf][1][j * AX_KRS + 4 * d4]) = kl;
            *reinterpret_cast<h4*>(&Vs[buf][0][ax_v_row(j) * AX_VRS + 4 * d4]) = vh;
            *reinterpret_cast<h4*>(&Vs[buf][1][ax_v_row(j) * AX_VRS + 4 * d4]) = vl;
        }
    };
    auto mm = [](h8 a, h8 bq, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc, 0, 0, 0); };
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int ntiles = (L + AX_KT - 1) / AX_KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * AX_KT;
        if (t + 1 < ntiles) load_tile(k0 + AX_KT);
        f32x16 s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 kh = *reinterpret_cast<const h8*>(&Ks[buf][0][(blk * 32 + n) * AX_KRS + 16 * ks + 8 * hf]);
                const h8 kl = *reinterpret_cast<const h8*>(&Ks[buf][1][(blk * 32 + n) * AX_KRS + 16 * ks + 8 * hf]);
                s[blk] = mm(kh, qh[ks], s[blk]);
                s[blk] = mm(kl, qh[ks], s[blk]);
                s[blk] = mm(kh, ql[ks], s[blk]);
            }
        }
        if (k0 + AX_KT > L) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= L) s[blk][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
        const float mc = m_new * c;
        float psum = 0.f;
        h8 ph[2][2], pl[2][2];                                         // P^T as B operand: [block][k-step of 16 keys], hi and lo halfs
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(fmaf(s[blk][8 * ks + j], c, -mc));
                    psum += p[j];
                }
                split8(p, ph[blk][ks], pl[blk][ks]);
            }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        {
            const int li = lane & 15, g1 = (lane >> 4) & 1, q4 = li >> 2, p4 = li & 3;
            const int voff = (8 * hf + q4) * AX_VRS + 16 * g1 + 4 * p4;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 vf[2];
#pragma unroll
                    for (int pln = 0; pln < 2; ++pln) {
                        const _Float16* p0 = &Vs[buf][pln][voff + (blk * 32 + ks * 16) * AX_VRS];
                        const v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
                        const v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0 + 4 * AX_VRS));
                        typedef short s8 __attribute__((ext_vector_type(8)));
                        const s8 both = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                        vf[pln] = __builtin_bit_cast(h8, both);
                    }
                    o = mm(vf[0], ph[blk][ks], o);
                    o = mm(vf[1], ph[blk][ks], o);
                    o = mm(vf[0], pl[blk][ks], o);
                }
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    const int q = q0 + wave * 32 + n;
    if (q < L) {
        float* op = out + ((size_t)b * L + q) * D + h * 32 + 4 * hf;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(op + 8 * gq) = make_float4(o[4 * gq + 0] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
    }
}

template <bool RELU, bool CONV3>
static void gemm(const float* A, int lda, const float* W, const float* bias, const float* R, float* C, int ldc, size_t M, int N,
                 int K, int Lrow, hipStream_t st) {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
    hipLaunchKernelGGL((gemm_kernel<RELU, CONV3>), grid, dim3(256), 0, st, A, lda, W, bias, R, C, ldc, M, N, K, Lrow);
}

}  // namespace tf32

// Workspace (floats): x [B*L*256] | y [B*L*256] | qkv [M*768] | att [M*256] | u [M*1024] | t [M*256]
size_t tf32_workspace_floats(int B, int L) {
    const size_t M = (size_t)B * (L / 8);
    return (size_t)2 * B * L * D + M * (768 + 256 + 1024 + 256);
}

// `get(key)`: fp32 device pointer of a reference state-dict tensor.  h [M][256] receives the encoder output (the residual stream
// the pooling head reads), exactly where the 16-bit path leaves it.
int tf32_forward(const unsigned char* ids8, int ids_stride, int B, int L, int n_layers, float* ws, float* h,
                 const float* (*get)(void*, const std::string&), void* ctx, hipStream_t st, bool unfused, bool x3) {
    using namespace tf32;
    const int L1 = L / 2, L2 = L1 / 2, L3 = L2 / 2;
    const size_t M = (size_t)B * L3;
    float* x = ws;
    float* y = x + (size_t)B * L * D;
    float* qkv = y + (size_t)B * L * D;
    float* att = qkv + M * 768;
    float* u = att + M * D;
    float* t = u + M * 1024;
    auto Wt = [&](const std::string& k) { return get(ctx, k); };
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)(((size_t)B * L + 3) / 4)), dim3(256), 0, st, ids8, ids_stride, Wt("embedding.weight"),
                       x, B, L);
    // 3 x [Conv1d(k = 3, padding = 1) -> ReLU -> MaxPool1d(2)]; a trailing odd position is dropped by the pooling, as in torch
    int Lin = L;
    for (int i : {0, 3, 6}) {
        const std::string n = "cnn." + std::to_string(i);
        const int Lout = Lin / 2;
        if (!unfused) {      // round 4: convolution + ReLU + pooling in one kernel on the fp32 MFMA (tail32.hip conv32_kernel)
            launch_conv32(x, get(ctx, "t32." + n), Wt(n + ".bias"), y, B, Lin, st, x3);
            std::swap(x, y);
        } else {
            gemm<true, true>(x, D, Wt(n + ".weight"), Wt(n + ".bias"), nullptr, y, D, (size_t)B * Lin, D, 3 * D, Lin, st);
            hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)(((size_t)B * Lout * (D / 4) + 255) / 256)), dim3(256), 0, st, y, x, B, Lin, Lout);
        }
        Lin = Lout;
    }
    // + positional encoding, LayerNorm -> residual stream
    hipLaunchKernelGGL(ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, Wt("pos_encoder.pe"), Wt("norm.weight"),
                       Wt("norm.bias"), h, M, L3, 1e-5f);
    // Round 4: the dense layers of an encoder layer run fused on the fp32 MFMA (tail32.hip enc32_kernel: out_proj + LN1 + FFN + LN2 +
    // the next layer's in_proj on 64-token tiles; weights in its packing under "t32.<layer>.<in|out|ff1|ff2>").  CLM_DEBUG=unfused_fp32:
    // the seven separate launches per layer of round 2 (the tests cross-check the two).
    auto T32 = [&](int i, const char* what) { return get(ctx, "t32." + std::to_string(i) + "." + what); };
    auto LP = [&](int i) { return "transformer_encoder.layers." + std::to_string(i) + "."; };
    if (!unfused && n_layers > 0)
        launch_enc32(nullptr, h, nullptr, nullptr, nullptr, T32(0, "in"), nullptr, nullptr, nullptr, Wt(LP(0) + "self_attn.in_proj_bias"),
                     nullptr, nullptr, nullptr, nullptr, qkv, M, 1e-5f, st, x3);
    for (int i = 0; i < n_layers; ++i) {
        const std::string p = LP(i);
        if (!unfused) {
            if (x3) hipLaunchKernelGGL(attention_x3_kernel, dim3((unsigned)(((L3 + AX_QT - 1) / AX_QT) * 8 * B)), dim3(256), 0, st, qkv, att, L3);
            else hipLaunchKernelGGL(attention32_kernel, dim3((unsigned)(((L3 + A32_QT - 1) / A32_QT) * 8 * B)), dim3(256), 0, st, qkv, att, L3);
            const bool more = i + 1 < n_layers;
            launch_enc32(att, h, T32(i, "out"), T32(i, "ff1"), T32(i, "ff2"), more ? T32(i + 1, "in") : nullptr,
                         Wt(p + "self_attn.out_proj.bias"), Wt(p + "linear1.bias"), Wt(p + "linear2.bias"),
                         more ? Wt(LP(i + 1) + "self_attn.in_proj_bias") : nullptr, Wt(p + "norm1.weight"), Wt(p + "norm1.bias"),
                         Wt(p + "norm2.weight"), Wt(p + "norm2.bias"), qkv, M, 1e-5f, st, x3);
            continue;
        }
        gemm<false, false>(h, D, Wt(p + "self_attn.in_proj_weight"), Wt(p + "self_attn.in_proj_bias"), nullptr, qkv, 768, M, 768, D, 0, st);
        hipLaunchKernelGGL(attention_kernel, dim3((unsigned)((L3 + 255) / 256), 8, (unsigned)B), dim3(256), 0, st, qkv, att, L3);
        gemm<false, false>(att, D, Wt(p + "self_attn.out_proj.weight"), Wt(p + "self_attn.out_proj.bias"), h, t, D, M, D, D, 0, st);
        hipLaunchKernelGGL(ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, t, (const float*)nullptr, Wt(p + "norm1.weight"),
                           Wt(p + "norm1.bias"), h, M, 1, 1e-5f);
        gemm<true, false>(h, D, Wt(p + "linear1.weight"), Wt(p + "linear1.bias"), nullptr, u, 1024, M, 1024, D, 0, st);
        gemm<false, false>(u, 1024, Wt(p + "linear2.weight"), Wt(p + "linear2.bias"), h, t, D, M, D, 1024, 0, st);
        hipLaunchKernelGGL(ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, t, (const float*)nullptr, Wt(p + "norm2.weight"),
                           Wt(p + "norm2.bias"), h, M, 1, 1e-5f);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace clm
