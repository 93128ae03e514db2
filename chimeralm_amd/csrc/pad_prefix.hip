// pad_prefix.hip -- round 5: the [PAD] prefix of a left-padded batch is computed ONCE per weight load.
//
// The reference pads every batch on the left to its longest read and masks nothing: the pad tokens run through the backbone and
// are attended and pooled like any other (/root/reference/chimeralm/data/tokenizer.py:152-159, models/components/hyena.py:244-256).
// The backbone is causal (tests/test_gpu_parity.py::test_the_residual_stream_is_causal) and has no positional term on the tokens,
// so the hidden state of position t INSIDE a read's pad prefix is the same in every read of every batch: a function of (weights,
// t, layer).  On the reference's own test BAM (median read 3.9k bases, batches of 12 padded to up to 32,769 tokens) 69 % of all
// 128-token tiles lie wholly inside such a prefix.  So, per weight load and arithmetic, one all-[PAD] read is run through the
// engine and what the rest of the path reads of it is kept (clm_api.hip PadTable): per block the z rows the next convolution reads,
// and for the last block the pooling scores and per-tile pooling partials (exact path: the final residual rows).  In a batch:
//   1. pad_count_kernel    p0[b] = 128-token tiles wholly inside read b's leading run of [PAD]
//   2. tile_list_kernel    the list of tiles the persistent tail kernels compute (gemm16.hip: contiguous ranges of the LIST per
//                          workgroup, so the work stays balanced whatever the prefixes are); each read enters one tile before
//                          its first non-prefix tile -- that tile gives the gated in_proj stage its two-token history
//   3. prefix_fill_*       after each block's tail kernel: rows [0, 128 p0[b]) of the read's z block (scores, partials, h) <- table
// The convolution is unchanged: it reads whole rows, prefix included.  A batch without pads pays two tiny launches per chunk and
// four empty fill launches.
#include "clm_common.h"

namespace clm {

__global__ __launch_bounds__(256) void pad_count_kernel(const unsigned char* __restrict__ ids8, int B, int Lp, int Lmain, int enabled,
                                                        int* __restrict__ p0) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63, b = (int)blockIdx.x * 4 + wave;
    if (b >= B) return;                                    // (whole wave)
    int P = 0;
    if (enabled) {
        const unsigned char* row = ids8 + (size_t)b * Lp;
        constexpr unsigned PADW = PAD_ID * 0x01010101u;
        for (int t = 0; t < Lmain; t += 1024) {            // one KiB per wave and step, 16 bytes per lane (Lp is a multiple of 64)
            const int off = t + lane * 16;
            uint4 v = make_uint4(~PADW, ~PADW, ~PADW, ~PADW);          // beyond the row: "not a pad"
            if (off < Lp) v = *reinterpret_cast<const uint4*>(row + off);
            const unsigned w[4] = {v.x ^ PADW, v.y ^ PADW, v.z ^ PADW, v.w ^ PADW};
            int n = 16;                                    // leading [PAD] bytes of this lane's 16
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (w[k]) n = 4 * k + ((__ffs((int)w[k]) - 1) >> 3);
            const unsigned long long m = __ballot(n < 16);
            if (m) {
                const int fl = __ffsll((long long)m) - 1;
                P = t + fl * 16 + __shfl(n, fl, 64);
                break;
            }
            P = t + 1024;
        }
        if (P > Lmain) P = Lmain;
    }
    if (lane == 0) p0[b] = P / 128;
}

__global__ __launch_bounds__(256) void tile_list_kernel(const int* __restrict__ p0, int B, int tiles_x, int* __restrict__ tiles) {
    __shared__ int s[256];
    __shared__ int base;
    const int tid = (int)threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + tid;
        int start = 0, n = 0;
        if (b < B) {
            const int p = p0[b] < tiles_x ? p0[b] : tiles_x;
            start = p > 0 ? p - 1 : 0;
            n = tiles_x - start;
        }
        s[tid] = n;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                // inclusive scan
            const int v = tid >= d ? s[tid - d] : 0;
            __syncthreads();
            s[tid] += v;
            __syncthreads();
        }
        const int off = base + s[tid] - n;
        for (int k = 0; k < n; ++k) tiles[1 + off + k] = tile_entry(b, start + k, k > 0);
        __syncthreads();
        if (tid == 255) base += s[255];
        __syncthreads();
    }
    if (tid == 0) tiles[0] = base;
}

void launch_pad_tiles(const unsigned char* ids8, int B, int Lp, int Lmain, int enabled, int* p0, int* tiles, hipStream_t st) {
    const int tiles_x = (Lmain + 127) / 128;
    hipLaunchKernelGGL(pad_count_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, ids8, B, Lp, Lmain, enabled, p0);
    hipLaunchKernelGGL(tile_list_kernel, dim3(1), dim3(256), 0, st, p0, B, tiles_x, tiles);
}

// one block = rows blockIdx.x, blockIdx.x + 32, ... of read blockIdx.y; 16-byte pieces (rows start 64-byte aligned: Lp, LpT are
// multiples of 64; the filled length is a multiple of 128 tokens or ends at Lmain, whose ragged end is copied byte-wise)
__global__ __launch_bounds__(256) void prefix_fill_z_kernel(const int* __restrict__ p0, unsigned char* __restrict__ z,
                                                            const unsigned char* __restrict__ table, int Lp, int LpT, int Lmain, int es,
                                                            int nrow16, int nlo, int B, int seg_skip_S, const int* __restrict__ partner) {
    const int b = (int)blockIdx.y, p = p0[b];
    if (p == 0) return;
    const int ntok = 128 * p < Lmain ? 128 * p : Lmain;
    // the segmented convolution that reads these rows starts at segment m_start of the PAIR (hyena_conv.hip SegPrefix, same formula):
    // tokens before it are never read
    int tok0 = 0;
    const int other = partner ? partner[b] : ((b ^ 1) < B ? (b ^ 1) : -1);
    if (seg_skip_S > 1 && other >= 0) {
        const int pb = p0[other], pm = p < pb ? p : pb;
        int m_start = pm > 0 ? ((pm - 1) * 128) / SEG_LEN : 0;
        if (m_start > seg_skip_S - 1) m_start = seg_skip_S - 1;
        tok0 = m_start * SEG_LEN;
    }
    if (tok0 >= ntok) return;
    unsigned char* zb = z + (size_t)b * D3 * Lp * es;
    for (int r = (int)blockIdx.x; r < nrow16 + nlo; r += (int)gridDim.x) {
        const bool lo = r >= nrow16;                       // lo planes: byte rows behind 2 D element rows (clm_common.h TailArgs::zlo)
        const size_t bytes = (size_t)(ntok - tok0) * (lo ? 1 : es), skip = (size_t)tok0 * (lo ? 1 : es);   // (tok0: a multiple of 8,192)
        const unsigned char* src = (lo ? table + (size_t)2 * D * LpT * es + (size_t)(r - nrow16) * LpT : table + (size_t)r * LpT * es) + skip;
        unsigned char* dst = (lo ? zb + (size_t)2 * D * Lp * es + (size_t)(r - nrow16) * Lp : zb + (size_t)r * Lp * es) + skip;
        const size_t n16 = bytes / 16;
        for (size_t i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = n16 * 16 + threadIdx.x; i < bytes; i += 256) dst[i] = src[i];
    }
}

void launch_prefix_fill_z(const int* p0, void* z, const void* table, int B, int Lp, int LpT, int Lmain, int es, int nrow16, int nlo,
                          hipStream_t st, int seg_skip_S, const int* partner) {
    hipLaunchKernelGGL(prefix_fill_z_kernel, dim3(32, (unsigned)B), dim3(256), 0, st, p0, reinterpret_cast<unsigned char*>(z),
                       reinterpret_cast<const unsigned char*>(table), Lp, LpT, Lmain, es, nrow16, nlo, B, seg_skip_S, partner);
}

// Pairs of the segmented convolution by descending prefix length (stable: equal prefixes keep batch order, so a batch without pads
// keeps its pairs): rank by counting -- B is a chunk's reads (12 ... 256, at most 4,096), one workgroup.
__global__ __launch_bounds__(256) void pair_order_kernel(const int* __restrict__ p0, int B, int* __restrict__ perm, int* __restrict__ partner) {
    for (int b = (int)threadIdx.x; b < B; b += 256) {
        const int p = p0[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int q = p0[j];
            rank += (q > p || (q == p && j < b)) ? 1 : 0;
        }
        perm[rank] = b;
    }
    __threadfence_block();                                 // (one workgroup; perm is in global memory: the writes before the barrier)
    __syncthreads();
    for (int r = (int)threadIdx.x; r < B; r += 256) partner[perm[r]] = (r ^ 1) < B ? perm[r ^ 1] : -1;
}
void launch_pair_order(const int* p0, int B, int* perm, int* partner, hipStream_t st) {
    hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(256), 0, st, p0, B, perm, partner);
}

__global__ __launch_bounds__(256) void prefix_fill_pool_kernel(const int* __restrict__ p0, float* __restrict__ scores, float* __restrict__ partial,
                                                               const float* __restrict__ t_scores, const float* __restrict__ t_partial, int L,
                                                               int ntiles, int Lmain, int per128) {
    const int b = (int)blockIdx.y, p = p0[b];
    if (p == 0) return;
    // (per128: partials per 128-token tile of p0's count -- 1 in the 16-bit path, 2 for the exact path's 64-token tiles)
    const int ntok = 128 * p < Lmain ? 128 * p : Lmain, nt = p * per128 < ntiles ? p * per128 : ntiles;
    const int i0 = (int)blockIdx.x * 256 + (int)threadIdx.x, stride = (int)gridDim.x * 256;
    for (int i = i0; i < ntok; i += stride) scores[(size_t)b * L + i] = t_scores[i];
    // whole tiles only: a tile that ends at Lmain but is not all prefix was computed by the tail kernel (p counts WHOLE tiles)
    for (int i = i0; i < nt * POOL_PSTRIDE; i += stride) partial[(size_t)b * ntiles * POOL_PSTRIDE + i] = t_partial[i];
}

void launch_prefix_fill_pool(const int* p0, float* scores, float* partial, const float* t_scores, const float* t_partial, int B, int L,
                             int ntiles, int Lmain, hipStream_t st, int per128) {
    hipLaunchKernelGGL(prefix_fill_pool_kernel, dim3(16, (unsigned)B), dim3(256), 0, st, p0, scores, partial, t_scores, t_partial, L, ntiles,
                       Lmain, per128);
}

__global__ __launch_bounds__(256) void prefix_fill_h_kernel(const int* __restrict__ p0, float* __restrict__ h, const float* __restrict__ t_h, int L,
                                                            int Lmain) {
    const int b = (int)blockIdx.y, p = p0[b];
    if (p == 0) return;
    const size_t n4 = (size_t)(128 * p < Lmain ? 128 * p : Lmain) * D / 4;
    float4* dst = reinterpret_cast<float4*>(h + (size_t)b * L * D);
    const float4* src = reinterpret_cast<const float4*>(t_h);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

void launch_prefix_fill_h(const int* p0, float* h, const float* t_h, int B, int L, int Lmain, hipStream_t st) {
    hipLaunchKernelGGL(prefix_fill_h_kernel, dim3(64, (unsigned)B), dim3(256), 0, st, p0, h, t_h, L, Lmain);
}

}  // namespace clm
