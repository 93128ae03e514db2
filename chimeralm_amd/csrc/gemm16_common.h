// gemm16_common.h -- device helpers shared by the 16-bit MFMA kernels (gemm16.hip: Hyena blocks; tf_model.hip: the
// SequenceCNNTransformer encoder): LDS tile strides, MFMA loops over an LDS-resident activation tile with register-resident
// weight half-sets, LayerNorm straight from accumulator registers.  See gemm16.hip / gemm.hip for the design notes.
#pragma once
#include "gemm_common.h"

namespace clm {

using v4i16 = short __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) * lds_v4i16_ptr;

constexpr int RS16 = 264;     // row stride (elements) of token-major LDS tiles: 528 B, conflict-free ds_read_b128
constexpr int RSKM = 160;     // row stride of the k-major (channel-major) tile: 320 B = 256 B + 64 B, so the four
                              // rows of a transposing read land on four different 64-byte bank groups
constexpr int RSOUT = 136;    // row stride of the wave-private output staging tile: 272 B (16-byte aligned rows)
// Round 4, the lo tile of a compensated ACTIVATION operand (gemm_common.h lo8_pack4 / mfma_lo2): token-major, one e5m2 byte per
// feature, row stride 272 B = 256 + 16 (like RS16: a 16-lane group's 16-byte reads fall on 16 different bank groups).  Inside a
// row the 32 bytes a lane feeds the K = 64 MFMA of the 64-deep group g = k >> 6 for its k-half h = (k >> 3) & 1 lie together, in
// the k order of the fp16 fragments (byte 8 s + j = k-step s = (k >> 4) & 3, element j = k & 7): two 16-byte reads per row tile.
// No swizzle on top of that, on purpose: ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) over 64
// banks, and with a row stride of 68 dwords the 16 rows of such a group fall on 16 different 16-byte bank groups exactly when
// all lanes read the SAME chunk -- an XOR swizzle by the row's 16-row group (tried first, to spread the writers) made every one of
// the 160 lo reads of a tile 2-way conflicted: +5k LDS cycles per tile, SQ_LDS_BANK_CONFLICT 69 M -> 170 M per launch.  The
// LayerNorm writer's 16 ds_write_b32 per lane are 4-way conflicted instead (all 32 lanes of a group write the same in-chunk dword:
// 8 reachable banks; +768 LDS cycles per tile), which is the cheaper end (tools/dev/lds_lo_sim.py).
constexpr int RSL = 272;
__host__ __device__ constexpr int lo_pos(int k) { return (k & ~63) + 32 * ((k >> 3) & 1) + 8 * ((k >> 4) & 3) + (k & 7); }

template <int PREC>
__device__ __forceinline__ unsigned short to_bits(float v) {
    return from_float<typename CT<PREC>::elem>(v).bits;
}

// acc[mt] += A-tile(token-major LDS) x weight set `part` (k-steps [part * KPS, (part + 1) * KPS) of the 256-deep chunk).
// ROWS_N: accumulator rows are output features (lane = token).  Compensated mode: the set holds (hi, lo) pairs and every
// A fragment feeds two MFMAs -- half the LDS bytes per MFMA of the plain modes.
// LO2 (fp16c only): the activations' lo tile `Al` (RSL, lo_pos) adds a third term per row tile and 64-deep group.
// MT0, NMT (fp16c only): the row tiles [MT0, MT0 + NMT) of the 128-token tile instead of all four (the gated in_proj stage runs its
// v block in two 64-token halves: half the accumulator registers at a time).
template <int PREC, bool ROWS_N, int AHEAD = 4, bool LO2_ = false, int MT0 = 0, int NMT = 4>
__device__ __forceinline__ void compute_tm(const typename CT<PREC>::elem* As, int part, int lrow, int lhalf,
                                           const u16x8 (&src)[1][SETK], f32x16 (&acc)[4], const unsigned char* Al = nullptr) {
    static_assert(!LO2_ || PREC == PREC_F16C, "activation lo tiles exist in the compensated mode only");
    static_assert((MT0 == 0 && NMT == 4) || PREC == PREC_F16C, "row-tile ranges are implemented for the compensated mode");
    constexpr bool LO2 = LO2_ && !lab::NOLO2;
    // Explicitly software-pipelined over the 4*KPS (k-step, row tile) items: the A fragment of item i + AHEAD is requested
    // before the MFMA(s) of item i (ring of AHEAD + 1 fragments), and that order is pinned with sched_group_barrier.  Left to
    // itself hipcc serialises `ds_read -> s_waitcnt lgkmcnt(0) -> mfma` wherever registers are tight (fc1: both accumulators
    // and two half-sets live), and every MFMA then pays a full LDS latency: fc1 ran at half the rate of fc2.
    constexpr int FR = WFR<PREC>, KP = KPS<PREC>, NI = 4 * KP, R = AHEAD + 1;
    const typename CT<PREC>::elem* a0 = As + lrow * RS16 + part * KP * 16 + lhalf * 8;
#define CLM_A_OFF(x) (lab::A0 ? 0 : (x))
    if constexpr (PREC == PREC_F16C) {
        // Compensated mode.  Items run ROW-TILE major (mt = i >> 2, k-step = i & 3): a row tile's four activation fragments feed
        // four fp16 MFMAs with the hi fragments and, their upper bytes gathered as e5m2 as they pass (8 registers, two v_perm_b32
        // per fragment: gemm_common.h frag_to_e5m2t), ONE K = 64 fp8 MFMA with the lo bytes -- half the cycles of the four fp16
        // lo MFMAs it replaces, 2.9e-6 instead of 8.3e-5 rms on a 64-deep product (tools/micro/mfma_fp8_lo.cpp).
        static_assert(KP == 4 && NI == 16, "a set is one 64-deep group");
        constexpr int NIR = 4 * NMT;                          // items of this call: (row tile, k-step), row-tile major
        u16x8 afc[R];
        i32x8 a8 = {0, 0, 0, 0, 0, 0, 0, 0};          // the row tile's 32 e5m2 bytes, gathered fragment by fragment (frag_to_e5m2t)
        i32x8 w8;
        {
            const unsigned* lo32 = reinterpret_cast<const unsigned*>(&src[0][4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) w8[r] = (int)lo32[r];
        }
        // LO2: the set's four hi fragments truncated to e5m2 (their upper bytes) are the weight operand of the activations' lo term
        i32x8 w8h = {0, 0, 0, 0, 0, 0, 0, 0}, alo = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned char* al0 = LO2 ? Al + lrow * RSL + 64 * part + 32 * lhalf : nullptr;
        if constexpr (LO2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                int w0, w1;
                frag_to_e5m2t(src[0][ks], w0, w1);
                w8h[2 * ks] = w0, w8h[2 * ks + 1] = w1;
            }
        }
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) afc[i % R] = *reinterpret_cast<const u16x8*>(a0 + CLM_A_OFF((MT0 + (i >> 2)) * 32 * RS16 + (i & 3) * 16));
#pragma unroll
        for (int i = 0; i < NIR; ++i) {
            if (i + AHEAD < NIR) {
                const int n = i + AHEAD;
                afc[n % R] = *reinterpret_cast<const u16x8*>(a0 + CLM_A_OFF((MT0 + (n >> 2)) * 32 * RS16 + (n & 3) * 16));
            }
            const int mt = MT0 + (i >> 2), ks = i & 3;
            if constexpr (LO2) {
                if (ks == 0) {                            // the row tile's 32 lo bytes: needed three MFMAs from now
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    const i32x4 p0 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL), p1 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL + 16);
                    alo = i32x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                }
            }
            if (ROWS_N) acc[mt] = mfma<PREC>(src[0][ks], afc[i % R], acc[mt]);
            else acc[mt] = mfma<PREC>(afc[i % R], src[0][ks], acc[mt]);
            if constexpr (!lab::NOLO) {
                int w0, w1;
                frag_to_e5m2t(afc[i % R], w0, w1);
                a8[2 * ks] = w0, a8[2 * ks + 1] = w1;
                if (ks == 3) acc[mt] = mfma_lo8<ROWS_N>(w8, a8, acc[mt]);
            }
            if constexpr (LO2) {
                if (ks == 3) acc[mt] = mfma_lo2<ROWS_N>(w8h, alo, acc[mt]);
            }
        }
        if constexpr (LO2) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
        for (int i = 0; i < NIR; ++i) {
            if (i + AHEAD < NIR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (LO2 && (i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (!lab::NOLO) {
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            if (LO2 && (i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        return;
    }
    u16x8 af[R];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) af[i % R] = *reinterpret_cast<const u16x8*>(a0 + (i & 3) * 32 * RS16 + (i >> 2) * 16);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i + AHEAD < NI) {
            const int n = i + AHEAD;
            af[n % R] = *reinterpret_cast<const u16x8*>(a0 + (n & 3) * 32 * RS16 + (n >> 2) * 16);
        }
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            if (ROWS_N) acc[i & 3] = mfma<PREC>(src[0][(i >> 2) * FR + f], af[i % R], acc[i & 3]);
            else acc[i & 3] = mfma<PREC>(af[i % R], src[0][(i >> 2) * FR + f], acc[i & 3]);
        }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i + AHEAD < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FR, 0);
    }
}

// Same, A operand taken from the k-major tile Ys[k][token] with the transposing read: a 16-lane group reads a
// 4(k) x 16(token) block and lane i receives token i's four k values (cdna_hip_programming.md T10).
template <int PREC, bool LO2_ = false>
__device__ __forceinline__ void compute_km(const typename CT<PREC>::elem* Ys, int part, int lane,
                                           const u16x8 (&src)[1][SETK], f32x16 (&acc)[4], const unsigned char* Al = nullptr) {
    static_assert(!LO2_ || PREC == PREC_F16C, "activation lo tiles exist in the compensated mode only");
    constexpr bool LO2 = LO2_ && !lab::NOLO2;
    constexpr int FR = WFR<PREC>, KP = KPS<PREC>;
    const int li = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5, q = li >> 2, p = li & 3;
    const typename CT<PREC>::elem* base = Ys + (8 * h + q) * RSKM + 16 * g1 + 4 * p;
    if constexpr (PREC == PREC_F16C) {      // as compute_tm: row-tile major, four fp16 hi MFMAs + one fp8 lo MFMA per row tile
        i32x8 w8;
        {
            const unsigned* lo32 = reinterpret_cast<const unsigned*>(&src[0][4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) w8[r] = (int)lo32[r];
        }
        i32x8 a8 = {0, 0, 0, 0, 0, 0, 0, 0};
        // LO2 (the y tile's lo bytes, token-major like every lo tile): as compute_tm
        i32x8 w8h = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned char* al0 = LO2 ? Al + (lane & 31) * RSL + 64 * part + 32 * h : nullptr;
        if constexpr (LO2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                int w0, w1;
                frag_to_e5m2t(src[0][ks], w0, w1);
                w8h[2 * ks] = w0, w8h[2 * ks + 1] = w1;
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            i32x8 alo = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (LO2) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 q0 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL), q1 = *reinterpret_cast<const i32x4*>(al0 + mt * 32 * RSL + 16);
                alo = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const typename CT<PREC>::elem* p0 = base + ((part * KP + ks) * 16) * RSKM + mt * 32;
                v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0));
                v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0 + 4 * RSKM));
                u16x8 af = {(unsigned short)lo[0], (unsigned short)lo[1], (unsigned short)lo[2], (unsigned short)lo[3],
                            (unsigned short)hi[0], (unsigned short)hi[1], (unsigned short)hi[2], (unsigned short)hi[3]};
                acc[mt] = mfma<PREC>(src[0][ks], af, acc[mt]);
                int w0, w1;
                frag_to_e5m2t(af, w0, w1);
                a8[2 * ks] = w0, a8[2 * ks + 1] = w1;
            }
            acc[mt] = mfma_lo8<true>(w8, a8, acc[mt]);
            if constexpr (LO2) acc[mt] = mfma_lo2<true>(w8h, alo, acc[mt]);
        }
        return;
    }
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const typename CT<PREC>::elem* p0 = base + ((part * KP + ks) * 16) * RSKM + mt * 32;
            v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0));
            v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p0 + 4 * RSKM));
            u16x8 af = {(unsigned short)lo[0], (unsigned short)lo[1], (unsigned short)lo[2], (unsigned short)lo[3],
                        (unsigned short)hi[0], (unsigned short)hi[1], (unsigned short)hi[2], (unsigned short)hi[3]};
#pragma unroll
            for (int f = 0; f < FR; ++f) acc[mt] = mfma<PREC>(src[0][ks * FR + f], af, acc[mt]);   // rows = output feature, cols = token
        }
    }
}

// ---- one 256-deep reduction phase of a wave's 32 output columns over an LDS-resident tile ---------------------------
// The weight sets of the phase ping-pong through bs[0] / bs[1]: on entry bs[0] holds (or has in flight) set 0 of
// (wp, nb, kc) and, if P1_LOADED, bs[1] set 1; set p + 1 is requested before the MFMAs of set p, and the slot of the last
// set requests set 0 of the FOLLOWING phase (wnext, nnb, nkc) -- unconditionally (a conditional prefetch makes hipcc's
// waitcnt pass merge the "nothing pending" state and wait inside the set, see gemm.hip).  NPARTS is even, so every phase
// ends with its successor's set 0 in bs[0] and bs[1] free.  `hook(hook0 + i)`, i = 0, 1, runs right after the weight request
// of each half of the phase: the caller may slip further global loads in there (they queue BEHIND the weights the next
// MFMAs wait for -- vmcnt completes in order -- and have a whole MFMA phase to land).
struct NoHook {
    __device__ __forceinline__ void operator()(int) const {}
};
// PRECN: the packing of the FOLLOWING phase's weights where it differs from this phase's (tail16_kernel in fp16c: the two MLP
// products run on plain fp16 weights between compensated out_proj / in_proj products).  PRECN = PREC_RAWNEXT: `wnext` already
// points at the lane's first fragment of the following set, whatever its packing, and all SETK fragments behind it are requested
// (a set of either packing is SETK or fewer consecutive fragments 64 apart; the surplus ones of a compensated set are the next
// set's first two: harmless) -- for a slot whose successor is only known at run time, without a branch around the request.
constexpr int PREC_SAME = -1, PREC_RAWNEXT = -2;
template <int PREC, int K>
__device__ __forceinline__ const u16x8* set_base(const u16x8* wp, int nb, int kc, int part, int wave, int lane) {
    constexpr int KSTEPS = 256 / CT<PREC>::MFMA_K, KSTEPS_ALL = K / CT<PREC>::MFMA_K;
    constexpr int FR = CT<PREC>::MFMA_K == 16 ? WFR<PREC> : 1, KP = SETK / FR;
    return wp + ((size_t)(nb * 8 + wave) * KSTEPS_ALL + kc * KSTEPS + part * KP) * (FR * 64) + lane;
}
template <int PREC, int K, int KN, bool ROWS_N, bool P1_LOADED = false, typename Hook = NoHook, int PRECN = PREC_SAME, bool LO2 = false,
          int MT0 = 0, int NMT = 4>
__device__ __forceinline__ void phase_tm(const typename CT<PREC>::elem* As, const u16x8* wp, int nb, int kc,
                                         const u16x8* wnext, int nnb, int nkc, int wave, int lane,
                                         u16x8 (&bs)[2][1][SETK], f32x16 (&acc)[4], Hook hook = Hook(), int hook0 = 0,
                                         const unsigned char* Al = nullptr /*LO2: the activations' lo tile*/) {
    constexpr int NP = NPARTS<PREC>;
    const int lrow = lane & 31, lhalf = lane >> 5;
    static_for<0, NP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p + 1 < NP) {
            if constexpr (!(P1_LOADED && p == 0)) load_set<PREC, K, 1>(wp, nb, kc, p + 1, wave, lane, bs[(p + 1) & 1]);
        } else if constexpr (PRECN == PREC_RAWNEXT) {
#pragma unroll
            for (int ks = 0; ks < SETK; ++ks) bs[0][0][ks] = wnext[(size_t)ks * 64];
        } else {
            load_set<PRECN == PREC_SAME ? PREC : PRECN, KN, 1>(wnext, nnb, nkc, 0, wave, lane, bs[0]);
        }
        if constexpr (p % (NP / 2) == 0) hook(hook0 + p / (NP / 2));
        __builtin_amdgcn_sched_barrier(0);
        compute_tm<PREC, ROWS_N, lab::AHEAD, LO2, MT0, NMT>(As, p, lrow, lhalf, bs[p & 1], acc, Al);
        __builtin_amdgcn_sched_barrier(0);
    });
}
// same over the k-major tile (out_proj); set 1 is always loaded by the caller
template <int PREC, int K, int KN, int PRECN = PREC, bool LO2 = false>
__device__ __forceinline__ void phase_km(const typename CT<PREC>::elem* Ys, const u16x8* wp, int nb, int kc,
                                         const u16x8* wnext, int nnb, int nkc, int wave, int lane,
                                         u16x8 (&bs)[2][1][SETK], f32x16 (&acc)[4], const unsigned char* Al = nullptr) {
    constexpr int NP = NPARTS<PREC>;
    static_for<0, NP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p > 0) {
            if constexpr (p + 1 < NP) load_set<PREC, K, 1>(wp, nb, kc, p + 1, wave, lane, bs[(p + 1) & 1]);
            else load_set<PRECN, KN, 1>(wnext, nnb, nkc, 0, wave, lane, bs[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        compute_km<PREC, LO2>(Ys, p, lane, bs[p & 1], acc, Al);
        __builtin_amdgcn_sched_barrier(0);
    });
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
}

// LayerNorm over the 256 features of every token of a tile whose values sit in accumulator registers (rows = this wave's
// 32 features as register quads, lane = token): per-lane (sum, squared deviations) over 16 features -> LDS tables -> combined
// over the 16 partials of the 8 waves; the normalised tile goes to As in the compute dtype (rows beyond L as zeros).
// P1 / P2: [16][128] floats each, outside As.  Ends with a barrier (As complete); the internal barrier also orders
// every earlier LDS access of the workgroup before the As writes.
// KEEP: the normalised fp32 values also replace the accumulator contents (post-norm blocks: they are the next residual).
// (gamma / beta from an LDS table instead of the global loads behind the barriers below, with beta folded into the consuming
//  layer's bias where possible: measured in tail16_kernel, 1.572 vs 1.573 ms per launch -- not kept.  The statistics and the
//  affine step on register pairs (v_pk_*_f32, half the arithmetic instructions but 190 more v_mov for splats and pairs):
//  1.568 vs 1.578 ms in the in_proj variant, 1.431 vs 1.422 in the score variant -- not kept either.)
// LO (fp16c): the normalised values also leave their lo bytes in `Al` (RSL, lo_pos; gemm_common.h lo8_pack4) -- the operand tile of
// the product that follows is then carried to ~15 bits instead of fp16's 11.
template <int PREC, bool KEEP = false, bool LO = false>
__device__ __forceinline__ void ln_acc_to_tile(f32x16 (&acc2)[4], float* P1, float* P2, const float* __restrict__ g,
                                               const float* __restrict__ bta, float eps, typename CT<PREC>::elem* As,
                                               int t0, int L, int wave, int lrow, int lhalf, unsigned char* Al = nullptr) {
    static_assert(!LO || PREC == PREC_F16C, "activation lo tiles exist in the compensated mode only");
    constexpr int BM = 128;
    float mean[4], rstd[4];
    if constexpr (lab::NOLN) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) mean[mt] = 0.f, rstd[mt] = 1.f;
        __syncthreads();
    } else if constexpr (lab::LN1PASS) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc2[mt][r], q = fmaf(acc2[mt][r], acc2[mt][r], q);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (lhalf == 0) P1[wave * BM + mt * 32 + lrow] = s, P2[wave * BM + mt * 32 + lrow] = q;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += P1[w * BM + mt * 32 + lrow], q += P2[w * BM + mt * 32 + lrow];
            mean[mt] = s * (1.0f / D);
            rstd[mt] = 1.0f / sqrtf(fmaxf(q * (1.0f / D) - mean[mt] * mean[mt], 0.f) + eps);
        }
    } else {
    // the two-exchange form: mean first, then the deviations about it.  (ONE exchange -- every lane leaves the sum of its 16 values
    // and their squared deviations about its own mean, combined exactly after Chan et al. -- was built and MEASURED SLOWER in round 3:
    // 22.76 vs 22.45 ms of tail kernel per step; the statistics cost VALU + LDS instructions, not barriers.  HISTORY.md section 4.9.)
    // (the two half-waves of a wave hold the same tokens: their partial sums are added through one lane exchange before they go to
    //  the table -- 8 partials per token instead of 16, half the table reads and adds of every thread.  Round 3, timing-only build
    //  of the final kernel: the statistics are 14 % of the tail kernel, profiles/r03_timing_only.txt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc2[mt][r];
        s += __shfl_xor(s, 32, 64);
        if (lhalf == 0) P1[wave * BM + mt * 32 + lrow] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += P1[w * BM + mt * 32 + lrow];
        mean[mt] = s * (1.0f / D);
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc2[mt][r] - mean[mt];
            v += d * d;
        }
        v += __shfl_xor(v, 32, 64);
        if (lhalf == 0) P2[wave * BM + mt * 32 + lrow] = v;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += P2[w * BM + mt * 32 + lrow];
        rstd[mt] = 1.0f / sqrtf(v * (1.0f / D) + eps);
    }
    }
    const float* gp = g + wave * 32 + 4 * lhalf;
    const float* bp = bta + wave * 32 + 4 * lhalf;
    // Rows beyond the read leave as zeros.  Only a read's last tile has any (none at all where the last token is peeled: 8k-bp
    // reads), so the 64 selects per thread are kept off the common path by a uniform branch (timing-only build without them:
    // -2 .. -4 % on the tail kernel).
    auto write_tile = [&](auto masked) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g4 = *reinterpret_cast<const float4*>(gp + 8 * q);
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bool ok = !decltype(masked)::value || t0 + mt * 32 + lrow < L;
                const float y0 = ok ? (acc2[mt][4 * q + 0] - mean[mt]) * rstd[mt] * g4.x + b4.x : 0.f;
                const float y1 = ok ? (acc2[mt][4 * q + 1] - mean[mt]) * rstd[mt] * g4.y + b4.y : 0.f;
                const float y2 = ok ? (acc2[mt][4 * q + 2] - mean[mt]) * rstd[mt] * g4.z + b4.z : 0.f;
                const float y3 = ok ? (acc2[mt][4 * q + 3] - mean[mt]) * rstd[mt] * g4.w + b4.w : 0.f;
                if (KEEP) {
                    acc2[mt][4 * q + 0] = y0;
                    acc2[mt][4 * q + 1] = y1;
                    acc2[mt][4 * q + 2] = y2;
                    acc2[mt][4 * q + 3] = y3;
                }
                if constexpr (LO) {
                    u16x4 pk;
                    const unsigned lo4 = lo8_pack4(y0, y1, y2, y3, pk);
                    *reinterpret_cast<u16x4*>(As + (mt * 32 + lrow) * RS16 + wave * 32 + 8 * q + 4 * lhalf) = pk;
                    // features wave * 32 + 8 q + 4 lhalf + {0..3}: one dword of the lo row (lo_pos is contiguous over 4-aligned j)
                    *reinterpret_cast<unsigned*>(Al + (mt * 32 + lrow) * RSL + lo_pos(wave * 32 + 8 * q) + 4 * lhalf) = lo4;
                } else {
                    u16x4 pk = {to_bits<PREC>(y0), to_bits<PREC>(y1), to_bits<PREC>(y2), to_bits<PREC>(y3)};
                    *reinterpret_cast<u16x4*>(As + (mt * 32 + lrow) * RS16 + wave * 32 + 8 * q + 4 * lhalf) = pk;
                }
            }
        }
    };
    if (t0 + BM <= L) write_tile(std::false_type{});        // (uniform)
    else write_tile(std::true_type{});
    __syncthreads();
}

}  // namespace clm
