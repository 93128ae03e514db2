// gemm.hip -- the dense projections of the ChimeraLM predict path on CDNA4 matrix cores.
//
// Replaces (reference arithmetic, executed there by aten GEMMs on CPU/CUDA):
//   HyenaDNA block  LN1 -> mixer.in_proj (256->768)      SURVEY.md section 8(a) rows 6-7(i)
//                   mixer.out_proj (256->256) + residual                    row 7(vii), row 6
//                   LN2 -> mlp.fc1 (256->1024) -> gelu(tanh) -> mlp.fc2 (1024->256) + residual   row 9
//   head            ln_f -> attention.0 (256->256) -> GELU -> attention.2 (256->1)
//                   /root/reference/chimeralm/models/components/hyena.py:50-53,119
//
// One workgroup (4 waves) owns a tile of BM tokens of ONE read and the whole output width: the activation
// tile is staged once into LDS in the compute dtype (LayerNorm fused into the staging: "activation
// stationary"), weights stream from L2 in pre-packed MFMA fragment order (1 KiB contiguous per wave load).
// Waves split the output columns (64 each per 256-wide block), so no weight fragment is fetched twice per
// workgroup.  Accumulation is always fp32 (v_mfma_f32_32x32x16_{bf16,f16} / v_mfma_f32_32x32x2_f32).
//
// MFMA 32x32 register maps (cdna_hip_programming.md section 3):
//   A/B operand, 16-bit: lane l holds row/col (l&31), k = 8*(l>>5) + j, j = 0..7
//   A/B operand, f32   : lane l holds row/col (l&31), k = (l>>5)
//   C/D                : col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
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

// ---------------------------------------------------------------------------------------- weight packing
// 16-bit: out[((nt*(K/16) + ks)*64 + lane)*8 + j] = W[nt*32 + (lane&31)][ks*16 + 8*(lane>>5) + j]
// f32   : out[ (nt*(K/2)  + ks)*64 + lane       ] = W[nt*32 + (lane&31)][ks*2  +    (lane>>5)    ]
template <int PREC>
__global__ void pack_weight_kernel(const float* __restrict__ w, typename CT<PREC>::elem* __restrict__ out, int n,
                                   int k) {
    using C = CT<PREC>;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * k) return;
    constexpr int KP = (PREC == PREC_F32) ? 1 : 8;
    int j = int(i % KP);
    size_t q = i / KP;
    int lane = int(q % 64);
    q /= 64;
    int ksteps = k / C::MFMA_K;
    int ks = int(q % ksteps);
    int nt = int(q / ksteps);
    int row = nt * 32 + (lane & 31);
    int col = ks * C::MFMA_K + KP * (lane >> 5) + j;
    out[i] = from_float<typename C::elem>(w[(size_t)row * k + col]);
}

size_t packed_weight_bytes(int prec, int n, int k) { return (size_t)n * k * (prec == PREC_F32 ? 4 : 2); }

void launch_pack_weight(int prec, const float* w, void* out, int n, int k, hipStream_t st) {
    size_t total = (size_t)n * k;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (prec == PREC_F32)
        hipLaunchKernelGGL(pack_weight_kernel<PREC_F32>, grid, block, 0, st, w, (float*)out, n, k);
    else if (prec == PREC_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<PREC_BF16>, grid, block, 0, st, w, (bf16_t*)out, n, k);
    else
        hipLaunchKernelGGL(pack_weight_kernel<PREC_F16>, grid, block, 0, st, w, (f16_t*)out, n, k);
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

template <int PREC, int ASRC, int EPI, int K, int N>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    using C = CT<PREC>;
    using elem = typename C::elem;
    using frag = typename C::frag;
    constexpr int BM = C::BM, RS = C::RS, MT = BM / 32, KC = 256;
    constexpr int NBLK = N / 256, KCH = K / KC, KSTEPS = KC / C::MFMA_K, KSTEPS_ALL = K / C::MFMA_K;
    static_assert(NBLK == 1 || KCH == 1, "either the output or the reduction is blocked, not both");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * BM, L = a.L, Lp = a.Lp;
    const int lrow = lane & 31, lhalf = lane >> 5;

    for (int nb = 0; nb < NBLK; ++nb) {
        f32x16 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        for (int kc = 0; kc < KCH; ++kc) {
            if (KCH > 1 || nb == 0) {
                __syncthreads();
                // ------------------------------------------------ stage A[BM][KC] into LDS (compute dtype)
                if (ASRC == A_LN) {
                    const float4 g4 = *reinterpret_cast<const float4*>(a.ln_g + lane * 4);
                    const float4 b4 = *reinterpret_cast<const float4*>(a.ln_b + lane * 4);
                    for (int r = wave; r < BM; r += 4) {
                        int t = t0 + r;
                        float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
                        if (t < L) {  // wave-uniform
                            float4 x = *reinterpret_cast<const float4*>(a.h_in + ((size_t)b * L + t) * D + lane * 4);
                            float mean = wave_sum((x.x + x.y) + (x.z + x.w)) * (1.0f / D);
                            float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                            float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / D);
                            float rstd = 1.0f / sqrtf(var + a.eps);
                            y0 = d0 * rstd * g4.x + b4.x;
                            y1 = d1 * rstd * g4.y + b4.y;
                            y2 = d2 * rstd * g4.z + b4.z;
                            y3 = d3 * rstd * g4.w + b4.w;
                        }
                        store4<elem>(As + r * RS + lane * 4, y0, y1, y2, y3);
                    }
                } else if (ASRC == A_TM) {
                    const elem* src = reinterpret_cast<const elem*>(a.a_in);
                    if (PREC == PREC_F32) {
                        for (int r = wave; r < BM; r += 4) {
                            int t = t0 + r;
                            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (t < L)
                                x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) +
                                                                     ((size_t)b * L + t) * K + kc * KC + lane * 4);
                            float* d = reinterpret_cast<float*>(As) + r * RS + lane * 4;
                            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
                        }
                    } else {
                        for (int r = tid >> 5; r < BM; r += 8) {
                            int t = t0 + r, c8 = (tid & 31) * 8;
                            uint4 x = make_uint4(0, 0, 0, 0);
                            if (t < L)
                                x = *reinterpret_cast<const uint4*>(src + ((size_t)b * L + t) * K + kc * KC + c8);
                            *reinterpret_cast<uint4*>(As + r * RS + c8) = x;
                        }
                    }
                } else {  // A_CM: y [B, 256, Lp] channel-major -> LDS [token][channel]   (v1: scalar LDS scatter)
                    const elem* src = reinterpret_cast<const elem*>(a.a_in);
                    constexpr int TPV = (PREC == PREC_F32) ? 4 : 8;   // tokens per 16-byte vector
                    constexpr int VPR = BM / TPV;                      // vectors per channel row (16)
                    static_assert(VPR == 16, "tile is 256 B per channel");
                    for (int c = tid >> 4; c < KC; c += 16) {
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
                __syncthreads();
            }
            // ---------------------------------------------------- MFMA over this K chunk
            const frag* wp = reinterpret_cast<const frag*>(a.w);
            const int nt0 = nb * 8 + wave * 2;
            const frag* wp0 = wp + ((size_t)(nt0 + 0) * KSTEPS_ALL + kc * KSTEPS) * 64 + lane;
            const frag* wp1 = wp + ((size_t)(nt0 + 1) * KSTEPS_ALL + kc * KSTEPS) * 64 + lane;
#pragma unroll 4
            for (int ks = 0; ks < KSTEPS; ++ks) {
                frag b0 = wp0[(size_t)ks * 64];
                frag b1 = wp1[(size_t)ks * 64];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    frag af;
                    if (PREC == PREC_F32)
                        af = *reinterpret_cast<const frag*>(As + (mt * 32 + lrow) * RS + ks * 2 + lhalf);
                    else
                        af = *reinterpret_cast<const frag*>(As + (mt * 32 + lrow) * RS + ks * 16 + lhalf * 8);
                    if (EPI == E_CM) {
                        acc[mt][0] = mfma<PREC>(b0, af, acc[mt][0]);
                        acc[mt][1] = mfma<PREC>(b1, af, acc[mt][1]);
                    } else {
                        acc[mt][0] = mfma<PREC>(af, b0, acc[mt][0]);
                        acc[mt][1] = mfma<PREC>(af, b1, acc[mt][1]);
                    }
                }
            }
        }

        // -------------------------------------------------------- epilogue of this 256-wide output block
        const int nbase = nb * 256 + wave * 64;
        if (EPI == E_CM) {
            elem* z = reinterpret_cast<elem*>(a.out);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int n = nbase + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    float bias = a.bias[n];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int t = t0 + mt * 32 + lrow;
                        if (t < L) z[((size_t)b * N + n) * Lp + t] = from_float<elem>(acc[mt][nt][r] + bias);
                    }
                }
        } else if (EPI == E_GELU_TM) {
            elem* u = reinterpret_cast<elem*>(a.out);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int n = nbase + nt * 32 + lrow;
                float bias = a.bias[n];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int t = t0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        if (t < L) u[((size_t)b * L + t) * N + n] = from_float<elem>(gelu_tanh(acc[mt][nt][r] + bias));
                    }
            }
        } else if (EPI == E_RESID) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int n = nbase + nt * 32 + lrow;
                float bias = a.bias[n];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int t = t0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        if (t < L) {
                            float* p = a.h_out + ((size_t)b * L + t) * D + n;
                            *p = *p + (acc[mt][nt][r] + bias);
                        }
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
}

template <int PREC, int ASRC, int EPI, int K, int N>
static void launch_gemm(const GemmArgs& a, hipStream_t st) {
    using C = CT<PREC>;
    constexpr size_t lds = (size_t)C::BM * C::RS * sizeof(typename C::elem);
    auto kern = gemm_kernel<PREC, ASRC, EPI, K, N>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    dim3 grid((a.L + C::BM - 1) / C::BM, a.B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
}

#define CLM_DISPATCH_PREC(prec, ASRC, EPI, K, N, args, st)                         \
    do {                                                                           \
        if ((prec) == PREC_F32) launch_gemm<PREC_F32, ASRC, EPI, K, N>(args, st);  \
        else if ((prec) == PREC_BF16) launch_gemm<PREC_BF16, ASRC, EPI, K, N>(args, st); \
        else launch_gemm<PREC_F16, ASRC, EPI, K, N>(args, st);                     \
    } while (0)

void launch_inproj(int prec, const float* h, const float* g, const float* b, const void* w, const float* bias, void* z,
                   int B, int L, int Lp, float eps, hipStream_t st) {
    GemmArgs a{};
    a.h_in = h; a.ln_g = g; a.ln_b = b; a.w = w; a.bias = bias; a.out = z; a.B = B; a.L = L; a.Lp = Lp; a.eps = eps;
    CLM_DISPATCH_PREC(prec, A_LN, E_CM, D, D3, a, st);
}

void launch_outproj(int prec, const void* y, const void* w, const float* bias, float* h, int B, int L, int Lp,
                    hipStream_t st) {
    GemmArgs a{};
    a.a_in = y; a.w = w; a.bias = bias; a.h_out = h; a.B = B; a.L = L; a.Lp = Lp;
    CLM_DISPATCH_PREC(prec, A_CM, E_RESID, D, D, a, st);
}

void launch_fc1(int prec, const float* h, const float* g, const float* b, const void* w, const float* bias, void* u,
                int B, int L, float eps, hipStream_t st) {
    GemmArgs a{};
    a.h_in = h; a.ln_g = g; a.ln_b = b; a.w = w; a.bias = bias; a.out = u; a.B = B; a.L = L; a.Lp = 0; a.eps = eps;
    CLM_DISPATCH_PREC(prec, A_LN, E_GELU_TM, D, DI, a, st);
}

void launch_fc2(int prec, const void* u, const void* w, const float* bias, float* h, int B, int L, hipStream_t st) {
    GemmArgs a{};
    a.a_in = u; a.w = w; a.bias = bias; a.h_out = h; a.B = B; a.L = L; a.Lp = 0;
    CLM_DISPATCH_PREC(prec, A_TM, E_RESID, DI, D, a, st);
}

void launch_score(int prec, const float* h, const float* g, const float* b, const void* w1, const float* b1,
                  const float* w2, const float* b2, float* scores, int B, int L, float eps, hipStream_t st) {
    GemmArgs a{};
    a.h_in = h; a.ln_g = g; a.ln_b = b; a.w = w1; a.bias = b1; a.w2 = w2; a.b2 = b2; a.scores = scores;
    a.B = B; a.L = L; a.Lp = 0; a.eps = eps;
    CLM_DISPATCH_PREC(prec, A_LN, E_SCORE, D, D, a, st);
}

}  // namespace clm
