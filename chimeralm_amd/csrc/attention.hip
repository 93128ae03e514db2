// attention.hip -- multi-head self-attention forward for the SequenceCNNTransformer encoder.
//
// Reference arithmetic: nn.TransformerEncoderLayer's self-attention as the reference constructs it
//   /root/reference/chimeralm/models/components/transformer.py:64-68,98   (d_model 256, nhead 8 -> head dim 32, batch_first,
//   no attention mask and no key-padding mask: every position attends to every position of its read)
//       a[b, i, h, :] = sum_j softmax_j(q[b,i,h,:] . k[b,j,h,:] / sqrt(32)) v[b, j, h, :]
// with q | k | v the three 256-wide thirds of the in_proj output (in_proj_weight [768, 256] is laid out q, k, v).
//
// Shape of the problem on MI355X: head dim 32 makes the softmax, not the matrix products, the bound -- per 32 x 32 block of
// scores a wave issues 4 MFMAs (128 cycles of pipe) and ~450 cycles of VALU (exp2 at quarter rate).  So the kernel is built
// around a cheap softmax:
//   * scores are computed TRANSPOSED, S^T = K Q^T (rows = keys, columns = queries): in the 32x32 accumulator layout a lane then
//     owns ONE query (column l & 31) and 16 of its 32 keys in registers, so the row maximum / row sum of the softmax are
//     register reductions plus one exchange between the two half-waves (lane ^ 32) -- no 32-lane shuffle trees;
//   * exp2(s*c - m*c) with c = log2(e)/sqrt(32): one FMA + one v_exp_f32 per score;
//   * P^T never leaves the registers: the accumulator registers of S^T, converted to 16 bit, ARE the B operand of
//     O^T = V^T P^T up to a fixed permutation of the key index inside each group of 16 -- which, being the reduction index, may
//     be permuted freely as long as V's rows are permuted the same way: V rows are simply stored in that order when the tile
//     is staged into LDS (bits 2 and 3 of the key index swapped);
//   * V^T fragments come from the [key][d] tile with transposing LDS reads (ds_read_b64_tr_b16), K fragments with plain
//     16-byte reads; K rows are padded to 80 bytes and V rows kept at 64 bytes so that both patterns are bank-conflict free
//     (MI355X_MICROARCH.md "LDS": 4 x 16-lane groups for b128, 2 x 32 for the transposing read);
//   * 256-thread workgroups (4 waves x 32 queries), 64-key tiles double-buffered in 18 KiB of LDS, < 128 VGPRs: several
//     workgroups per CU hide the exp latency of each other.
// Online softmax (running maximum m, running sum l, accumulator rescaled when m grows) over the key tiles; keys beyond L are
// masked to -inf in the last tile; fp32 statistics and accumulation, 16-bit MFMA inputs.
#include "chimeralm_hip.h"
#include "gemm_common.h"

namespace clm {

namespace {

using v4i16 = short __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) * lds_v4i16_ptr;

constexpr int HD = 32;            // head dim
constexpr int NH = D / HD;        // 8 heads
constexpr int QT = 128;           // queries per workgroup (4 waves x 32)
constexpr int KT = 64;            // keys per staged tile
constexpr int KRS = 40;           // K tile row stride in elements: 80 bytes
constexpr int VRS = 32;           // V tile row stride in elements: 64 bytes

// exchange between the two half-waves (lane ^ 32): one ds_bpermute per 64-key tile
__device__ __forceinline__ float half_max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32, 64); }

// row of the V tile that holds key `k` (k < KT): bits 2 and 3 swapped inside each group of 16
__device__ __forceinline__ int v_row(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

}  // namespace

// HILO (round 3, the transformer's fp16c mode): the output leaves as TWO 16-bit planes, out[0] = fp16(64 a) and out[1] =
// fp16(64 a - out[0]), `plane` elements apart.  The attention output is close to the same vector at every position of a read
// (an average of v over all keys), so its rounding to 16 bits is the one activation rounding of this net that does NOT average
// out in the pooling: measured on the CPU (tests/tf_error_probe.py) it alone moves the logits by 2-4e-3 where every other
// operand's rounding stays below 1e-3.  The factor 64 (exact) keeps the lo plane out of fp16's subnormal range; out_proj
// multiplies both planes by the same weights and scales its accumulators by 1/64 (tf_model.hip, enc_ffn16_kernel).
constexpr float ATT_HILO_SCALE = 64.0f;

template <int PREC, bool HILO = false>
__global__ __launch_bounds__(256, 4) void attention_fwd_kernel(const typename CT<PREC>::elem* __restrict__ qkv,
                                                            typename CT<PREC>::elem* __restrict__ out, int L, size_t plane) {
    using elem = typename CT<PREC>::elem;
    __shared__ __attribute__((aligned(16))) elem Ks[2][KT * KRS];
    __shared__ __attribute__((aligned(16))) elem Vs[2][KT * VRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
    // Workgroup -> (query tile, head, read).  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with
    // its own L2: the query tiles of one (read, head) -- which all stream the same K / V -- are given ids that are congruent
    // modulo 8 so that they share one XCD's L2 (with the natural order every tile pulled its own copy of K / V through the fabric:
    // rocprofv3 FETCH_SIZE showed 2.2 GB per launch for 0.4 GB of qkv).
    const int ntq = (L + QT - 1) / QT;
    const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const int bh = (slot / ntq) * 8 + xcd;                          // B * 8 (read, head) pairs: always a multiple of 8
    const int q0 = (slot % ntq) * QT, h = bh & 7, b = bh >> 3;
    const elem* base = qkv + (size_t)b * L * D3 + h * HD;            // row t: base + t * 768 ; q at +0, k at +256, v at +512
    const float c = 1.4426950408889634f * 0.17677669529663687f;      // log2(e) / sqrt(32)

    // Q^T as B operand, both k-steps: lane (query n, half hf) holds q[d = 16 s + 8 hf + 0..7]
    u16x8 qf[2];
    {
        const int q = q0 + wave * 32 + n;
        const elem* qp = base + (size_t)(q < L ? q : L - 1) * D3;
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 16 * s + 8 * hf);
    }
    // staging map: thread -> (key row, 16-byte piece) of the K and of the V tile: 64 keys x 4 pieces = 256 threads
    const int sk = tid >> 2, sp = tid & 3;
    auto load_tile = [&](int k0, uint4& kreg, uint4& vreg) {
        const int key = k0 + sk < L ? k0 + sk : L - 1;                // clamped; the score mask removes the clones
        const elem* p = base + (size_t)key * D3 + 8 * sp;
        kreg = *reinterpret_cast<const uint4*>(p + D);
        vreg = *reinterpret_cast<const uint4*>(p + 2 * D);
    };
    auto store_tile = [&](int buf, const uint4& kreg, const uint4& vreg) {
        *reinterpret_cast<uint4*>(&Ks[buf][sk * KRS + 8 * sp]) = kreg;
        *reinterpret_cast<uint4*>(&Vs[buf][v_row(sk) * VRS + 8 * sp]) = vreg;
    };

    f32x16 o;                      // O^T: rows d, column = this lane's query
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;  // running maximum (shared by both halves) and this half's share of the running sum

    const int ntiles = (L + KT - 1) / KT;
    uint4 kreg, vreg;
    load_tile(0, kreg, vreg);
    store_tile(0, kreg, vreg);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * KT;
        if (t + 1 < ntiles) load_tile(k0 + KT, kreg, vreg);           // in flight under the block below
        // ---- S^T = K Q^T for the 64 keys of the tile: two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u16x8 kf = *reinterpret_cast<const u16x8*>(&Ks[buf][(blk * 32 + n) * KRS + 16 * ks + 8 * hf]);
                s[blk] = mfma<PREC>(kf, qf[ks], s[blk]);
            }
        }
        if (k0 + KT > L) {                                             // last tile: keys beyond the read
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= L) s[blk][r] = -INFINITY;
        }
        // ---- online softmax: this lane's query, 32 of the 64 keys here, the other 32 in the partner half-wave
        float mx = s[0][0];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = half_max(mx);
        const float m_new = fmaxf(m, mx);                              // finite: key k0 is always valid
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);   // 0 on the first tile (m = -inf)
        const float mc = m_new * c;
        float psum = 0.f;
        u16x8 pf[2][2];                                                // P^T as B operand: [block][k-step of 16 keys]
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(fmaf(s[blk][8 * ks + j], c, -mc));
                    psum += p[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[blk][ks][j] = from_float<elem>(p[j]).bits;
            }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        // ---- O^T += V^T P^T: V^T fragments by transposing reads of the [key row][d] tile
        {
            const int li = lane & 15, g1 = (lane >> 4) & 1, q4 = li >> 2, p4 = li & 3;
            const elem* vb = &Vs[buf][(8 * hf + q4) * VRS + 16 * g1 + 4 * p4];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const elem* p0 = vb + (blk * 32 + ks * 16) * VRS;
                    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0));
                    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0 + 4 * VRS));
                    const u16x8 vf = {(unsigned short)lo[0], (unsigned short)lo[1], (unsigned short)lo[2], (unsigned short)lo[3],
                                      (unsigned short)hi[0], (unsigned short)hi[1], (unsigned short)hi[2], (unsigned short)hi[3]};
                    o = mfma<PREC>(vf, pf[blk][ks], o);
                }
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1, kreg, vreg);           // the other buffer was last read in trip t - 1
        __syncthreads();
    }
    // ---- normalise and store: lane (query n, half hf) holds d = (r & 3) + 8 (r >> 2) + 4 hf
    const float inv = 1.0f / half_sum(l);
    const int q = q0 + wave * 32 + n;
    if (q < L) {
        elem* op = out + ((size_t)b * L + q) * D + h * HD + 4 * hf;
        const float sc = HILO ? inv * ATT_HILO_SCALE : inv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const elem e0 = from_float<elem>(o[4 * g + 0] * sc), e1 = from_float<elem>(o[4 * g + 1] * sc),
                       e2 = from_float<elem>(o[4 * g + 2] * sc), e3 = from_float<elem>(o[4 * g + 3] * sc);
            u16x4 pk = {e0.bits, e1.bits, e2.bits, e3.bits};
            *reinterpret_cast<u16x4*>(op + 8 * g) = pk;
            if constexpr (HILO) {
                u16x4 pl = {from_float<elem>(o[4 * g + 0] * sc - to_float(e0)).bits, from_float<elem>(o[4 * g + 1] * sc - to_float(e1)).bits,
                            from_float<elem>(o[4 * g + 2] * sc - to_float(e2)).bits, from_float<elem>(o[4 * g + 3] * sc - to_float(e3)).bits};
                *reinterpret_cast<u16x4*>(op + plane + 8 * g) = pl;
            }
        }
    }
}

void launch_attention_fwd(int prec, const void* qkv, void* out, int B, int L, hipStream_t st, bool hilo) {
    static_assert(NH == 8, "the workgroup -> XCD mapping assumes 8 heads");
    dim3 grid((unsigned)(((L + QT - 1) / QT) * NH * B)), block(256);
    const size_t plane = (size_t)B * L * D;
    if (prec == PREC_BF16)
        hipLaunchKernelGGL(attention_fwd_kernel<PREC_BF16>, grid, block, 0, st, (const bf16_t*)qkv, (bf16_t*)out, L, plane);
    else if (hilo)
        hipLaunchKernelGGL((attention_fwd_kernel<PREC_F16, true>), grid, block, 0, st, (const f16_t*)qkv, (f16_t*)out, L, plane);
    else
        hipLaunchKernelGGL(attention_fwd_kernel<PREC_F16>, grid, block, 0, st, (const f16_t*)qkv, (f16_t*)out, L, plane);
}

}  // namespace clm

// ---- C ABI (include/chimeralm_hip.h): stand-alone entry for tests and for the encoder path under construction
extern "C" int clm_attention_fwd(const void* qkv, void* out, int B, int L, int precision, void* stream) {
    if (!qkv || !out || B < 1 || L < 1 || (precision != CLM_PREC_F16 && precision != CLM_PREC_BF16)) return CLM_E_INVALID;
    clm::launch_attention_fwd(precision == CLM_PREC_BF16 ? clm::PREC_BF16 : clm::PREC_F16, qkv, out, B, L,
                              reinterpret_cast<hipStream_t>(stream), false);
    return hipGetLastError() == hipSuccess ? CLM_OK : CLM_E_HIP;
}
