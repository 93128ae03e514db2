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
#include "gemm_common.h"

namespace clm {

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

// compensated mode: every fragment twice, hi = fp16(w) then lo = fp16(w - hi):
//   out[(((nt*(K/16) + ks)*2 + hl)*64 + lane)*8 + j]
// (|lo| <= 2^-12 |w| is an fp16 subnormal for |w| < 0.25; v_mfma_f32_32x32x16_f16 keeps subnormal inputs --
// tools/micro/mfma_denorm.cpp -- and the 2^-24 subnormal spacing still leaves the pair within 2^-19 of |w| = 0.06.)
// Compensated mode, stream of one (32-column tile, 64-deep group of four k-steps) = 8 fragment slots of 64 lanes x 16 bytes:
//   slots 0..3   hi = fp16(w) of k-steps 0..3 (the fp16 MFMA fragments as in pack_weight_kernel)
//   slots 4, 5   lo = e4m3((w - hi) * 2^17 * LO8_TRUNC_GAIN): the lane's 32 bytes of the K = 64 scaled MFMA, byte 8 s + j = k-step s,
//                element j -- the order in which the activation fragments of the four k-steps convert in registers
//   slots 6, 7   unused (never loaded: the stream keeps its 8-slot stride)
// |w - hi| <= 2^-11 |w|: with 2^17 the e4m3 range (448) holds every |w| < 8, larger ones saturate (their lo term is then short,
// never wrong in sign or NaN).
__global__ void pack_weight_split_kernel(const float* __restrict__ w, f16_t* __restrict__ out, int n, int k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * k) return;
    const int j = int(i % 8);
    size_t q = i / 8;
    const int lane = int(q % 64);
    q /= 64;
    const int ksteps = k / 16, ks = int(q % ksteps), nt = int(q / ksteps);
    const float v = w[(size_t)(nt * 32 + (lane & 31)) * k + ks * 16 + 8 * (lane >> 5) + j];
    const f16_t hi = from_float<f16_t>(v);
    // (x LO8_TRUNC_GAIN: the activations reach the lo product TRUNCATED to e5m2, 0.9155 of their value on average -- gemm_common.h)
    const float lo = fminf(fmaxf((v - to_float(hi)) * (LO8_SCALE * LO8_TRUNC_GAIN), -448.f), 448.f);
    const int group = ks / 4, s = ks % 4;
    const size_t set = ((size_t)nt * (ksteps / 4) + group) * (8 * 64 * 8);            // in 16-bit units
    out[set + ((size_t)s * 64 + lane) * 8 + j] = hi;
    const int two = __builtin_amdgcn_cvt_pk_fp8_f32(lo, 0.f, 0, false);                // e4m3, round to nearest even
    reinterpret_cast<unsigned char*>(out + set)[(size_t)(4 + (s >> 1)) * 1024 + lane * 16 + (s & 1) * 8 + j] = (unsigned char)(two & 0xff);
}

size_t packed_weight_bytes(int prec, int n, int k) {
    return (size_t)n * k * (prec == PREC_F32 ? 4 : prec == PREC_F16C ? 4 : 2);
}

void launch_pack_weight(int prec, const float* w, void* out, int n, int k, hipStream_t st) {
    size_t total = (size_t)n * k;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (prec == PREC_F32)
        hipLaunchKernelGGL(pack_weight_kernel<PREC_F32>, grid, block, 0, st, w, (float*)out, n, k);
    else if (prec == PREC_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<PREC_BF16>, grid, block, 0, st, w, (bf16_t*)out, n, k);
    else if (prec == PREC_F16C)
        hipLaunchKernelGGL(pack_weight_split_kernel, grid, block, 0, st, w, (f16_t*)out, n, k);
    else
        hipLaunchKernelGGL(pack_weight_kernel<PREC_F16>, grid, block, 0, st, w, (f16_t*)out, n, k);
}

// A "set" = the weight fragments one wave needs for one (256-wide output block, 256-deep reduction chunk):
// 2 column tiles x 16 k-steps of 16 bytes per lane (128 VGPRs for the 16-bit types).  The 16-bit kernels keep TWO
// sets in registers: set s+1 streams in from L2 while the 128 MFMAs of set s run (weights never touch LDS, and the
// very first set is in flight during the LayerNorm staging).  __builtin_amdgcn_sched_barrier pins that order; left
// alone, hipcc sinks every weight load to just before its first use and each group of four MFMAs then eats a full
// L2 round trip (measured: 106 TFLOP/s on fc1 at one wave per SIMD).
template <int PREC, int ASRC, int EPI, int K, int N>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    using C = CT<PREC>;
    using elem = typename C::elem;
    using frag = typename C::frag;
    constexpr int BM = C::BM, RS = C::RS, MT = BM / 32, KC = 256;
    constexpr int NBLK = N / 256, KCH = K / KC, KSTEPS = KC / C::MFMA_K, KSTEPS_ALL = K / C::MFMA_K;
    constexpr int NSETS = NBLK * KCH;
    static_assert(NBLK == 1 || KCH == 1, "either the output or the reduction is blocked, not both");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * BM, L = a.L;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const bool full_tile = t0 + BM <= L;   // workgroup-uniform
    const frag* wp = reinterpret_cast<const frag*>(a.w);
    f32x16 acc[MT][2];

    if constexpr (PREC == PREC_F32) {
        for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            for (int kc = 0; kc < KCH; ++kc) {
                if (KCH > 1 || nb == 0) {
                    __syncthreads();
                    stage_a_tile<PREC, ASRC, K>(a, As, b, t0, kc);
                    __syncthreads();
                }
                const int nt0 = nb * 8 + wave * 2;
                const frag* wp0 = wp + ((size_t)(nt0 + 0) * KSTEPS_ALL + kc * KSTEPS) * 64 + lane;
                const frag* wp1 = wp + ((size_t)(nt0 + 1) * KSTEPS_ALL + kc * KSTEPS) * 64 + lane;
#pragma unroll 8
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    frag b0 = wp0[(size_t)ks * 64];
                    frag b1 = wp1[(size_t)ks * 64];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        frag af = *reinterpret_cast<const frag*>(As + (mt * 32 + lrow) * RS + ks * 2 + lhalf);
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
            if (full_tile) epilogue<PREC, EPI, N, false>(a, acc, b, t0, nb, smem);
            else epilogue<PREC, EPI, N, true>(a, acc, b, t0, nb, smem);
        }
    } else {
        // set = SETK k-steps of both column tiles: 2 x SETK x 4 VGPRs; two sets are resident (ping-pong).
        // Blocks (output block x reduction chunk) run in a rolled loop; the SPC sets of a block are unrolled so
        // that every register-array index is a compile-time constant (SPC is even: buffer = set index & 1).
        constexpr int SPC = KSTEPS / SETK, NBLOCKS = NBLK * KCH;
        static_assert(SPC == 2, "ping-pong assumes two sets per block");
        frag bs[2][2][SETK];
        load_set<PREC, K>(wp, 0, 0, 0, wave, lane, bs[0]);     // in flight during the first A-tile staging
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int blk = 0; blk < NBLOCKS; ++blk) {
            const int nb = blk / KCH, kc = blk % KCH;
            if (kc == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
            if (KCH > 1 || nb == 0) {
                __syncthreads();
                stage_a_tile<PREC, ASRC, K>(a, As, b, t0, kc);
                __syncthreads();
            }
            load_set<PREC, K>(wp, nb, kc, 1, wave, lane, bs[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute_set<PREC, EPI>(As, 0, lrow, lhalf, bs[0], acc);
            __builtin_amdgcn_sched_barrier(0);
            {   // UNCONDITIONAL prefetch (wraps to block 0 after the last block): a conditional one makes hipcc's
                // waitcnt pass merge the "no loads pending" path and wait for the new loads inside this very set
                const int nx = (blk + 1 < NBLOCKS) ? blk + 1 : 0;
                load_set<PREC, K>(wp, nx / KCH, nx % KCH, 0, wave, lane, bs[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute_set<PREC, EPI>(As, 1, lrow, lhalf, bs[1], acc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KCH == 1) {   // one reduction chunk per output block: finish the block here
                if (full_tile) epilogue<PREC, EPI, N, false>(a, acc, b, t0, nb, smem);
                else epilogue<PREC, EPI, N, true>(a, acc, b, t0, nb, smem);
            }
        }
        if constexpr (KCH > 1) {        // blocked reduction (fc2): single output block, epilogue after the loop
            if (full_tile) epilogue<PREC, EPI, N, false>(a, acc, b, t0, 0, smem);
            else epilogue<PREC, EPI, N, true>(a, acc, b, t0, 0, smem);
        }
    }
}

template <int PREC, int ASRC, int EPI, int K, int N>
static void launch_gemm(const GemmArgs& a, hipStream_t st) {
    using C = CT<PREC>;
    constexpr size_t lds = (size_t)C::BM * C::RS * sizeof(typename C::elem);
    auto kern = gemm_kernel<PREC, ASRC, EPI, K, N>;
    CLM_SET_LDS(kern, lds);
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
