// gemm16.hip -- the tuned bf16/f16 projection kernels (fp32 accumulate) of the Hyena block on gfx950.
//
//   in_proj16   z = in_proj(LN1(h))                      reference: HyenaOperator.in_proj after HyenaBlock.norm1
//   out_proj16  h += out_proj(y^T)                                   HyenaOperator.out_proj + residual
//   mlp16       h += fc2(gelu_tanh(fc1(LN2(h))))                     HyenaBlock.norm2 + HyenaMlp + residual
//   
//
// What the first version got wrong (rocprofv3, profiles/r01_*): its epilogues issued one 2- or 4-byte global store
// per accumulator register (128 store instructions per wave and output block, ~100 cycles each under load) and
// out_proj transposed its channel-major input with 2-byte LDS scatters.  Here
//   * 16-bit outputs are transposed through a wave-private LDS tile and leave as 16-byte stores of whole rows;
//   * fp32 residual updates use the MFMA orientation whose accumulator quad is 4 consecutive features of one token:
//     one float4 read-modify-write per quad;
//   * out_proj keeps y in LDS exactly as it lies in HBM ([channel][token]) and forms MFMA operands with
//     ds_read_b64_tr_b16, the CDNA4 transposing LDS read (no transposition pass at all);
//   * fc1 -> GELU -> fc2 is ONE kernel: the 1024-wide hidden activations never leave the CU (they go from
//     accumulators through GELU into an LDS tile that is the next MFMA's A operand), which removes 4 KiB/token of
//     HBM traffic per layer and the whole fc1 store epilogue.
// Weight fragments stream from L2 through the same two-set register ping-pong as in gemm.hip.
//
// Geometry: 512-thread workgroups = 8 waves = 2 per SIMD, 128 tokens x 256 features per pass, ONE 32-column tile
// per wave (64 accumulator registers per GEMM), so a wave needs < 256 registers and the second wave on each SIMD
// runs MFMAs while the first sits in a wait, a GELU or a store phase.  (rocprofv3 --pmc on the 4-wave version:
// SQ_WAIT_ANY 47 % of wave cycles, MFMA busy 15 %.)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm16_common.h"

namespace clm {

// h[b, t, 32 features of this wave] += acc + bias.  Accumulator rows = output features (lane = token, register quad =
// 4 consecutive features).  Storing straight from that layout makes every wave instruction touch 64 different cache
// lines (one 16-byte piece per token row); measured 28 % of the fused MLP kernel.  Instead the wave transposes its
// 128 x 32 fp32 tile through LDS (16 KiB, XOR-swizzled 16-byte chunks) so that each read-modify-write instruction
// covers 8 token rows x one whole 128-byte line.  `scratch` must be free (all waves past their last LDS read).
__device__ __forceinline__ void resid_epilogue(float* h_out, const float* bias, f32x16 (&acc)[4], int b, int t0, int L,
                                               int wave, int lane, float* scratch) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    float* rs = scratch + wave * (128 * 32);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int tok = mt * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int chunk = (2 * q + lhalf) ^ (tok & 7);
            *reinterpret_cast<float4*>(rs + tok * 32 + 4 * chunk) =
                make_float4(acc[mt][4 * q + 0], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
        }
    }
    // wave-private tile: the LDS operations of one wave execute in order, no barrier needed
    const int c = lane & 7, rsub = lane >> 3;
    const float4 bb = *reinterpret_cast<const float4*>(bias + wave * 32 + 4 * c);
    float* hrow = h_out + ((size_t)b * L + t0) * D + wave * 32 + 4 * c;
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tr = i * 8 + rsub;
        v[i] = *reinterpret_cast<const float4*>(hrow + (size_t)(t0 + tr < L ? tr : 0) * D);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tr = i * 8 + rsub;
        const float4 a = *reinterpret_cast<const float4*>(rs + tr * 32 + 4 * (c ^ (tr & 7)));
        if (t0 + tr < L)
            *reinterpret_cast<float4*>(hrow + (size_t)tr * D) =
                make_float4(v[i].x + a.x + bb.x, v[i].y + a.y + bb.y, v[i].z + a.z + bb.z, v[i].w + a.w + bb.w);
    }
}

// ================================================================================================ in_proj
// in_proj over a staged LN tile: three 256-feature blocks, z written channel-major [B, 768, Lp].  On entry bs[0]
// holds (or has in flight) the first half-set of block 0; Zs = 8 wave-private staging tiles [32 features][RSOUT].
// `hook(step)`, step = 0..5, runs right after the weight request of each half-block: the caller may slip further global
// loads in there (they queue BEHIND the weights the next MFMAs wait for -- vmcnt completes in order -- and have a whole
// MFMA phase to land before the following weight request needs them out of the way).
template <int PREC, typename Hook = NoHook>
__device__ __forceinline__ void inproj_blocks(const typename CT<PREC>::elem* As, typename CT<PREC>::elem* Zs,
                                              const u16x8* wp, const float* __restrict__ bias_all, void* zout, int b,
                                              int t0, int Lp, int wave, int lane, u16x8 (&bs)[2][1][SETK],
                                              f32x16 (&acc)[4], Hook hook = Hook()) {
    using elem = typename CT<PREC>::elem;
    constexpr int K = D, NBLOCKS = D3 / 256;
    const int lrow = lane & 31, lhalf = lane >> 5;
    elem* zs = Zs + wave * 32 * RSOUT;
#pragma unroll
    for (int nb = 0; nb < NBLOCKS; ++nb) {
        zero_acc(acc);
        phase_tm<PREC, K, K, false>(As, wp, nb, 0, wp, nb + 1 < NBLOCKS ? nb + 1 : 0, 0, wave, lane, bs, acc, hook, 2 * nb);
        // epilogue: rows = tokens (register quads = 4 consecutive tokens), cols = feature (lane) -> zs[feature][token]
        const int nbase = nb * 256 + wave * 32;
        int lrow_e = lrow;                                     // opaque copy: the table address is formed here, not kept (spilled)
        asm volatile("" : "+v"(lrow_e));                       // across the MFMA phase -- see the GELU store of tail16_kernel
        const float bias = bias_all[nbase + lrow_e];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u16x4 pk = {to_bits<PREC>(acc[mt][4 * q + 0] + bias), to_bits<PREC>(acc[mt][4 * q + 1] + bias),
                            to_bits<PREC>(acc[mt][4 * q + 2] + bias), to_bits<PREC>(acc[mt][4 * q + 3] + bias)};
                *reinterpret_cast<u16x4*>(zs + lrow * RSOUT + mt * 32 + 8 * q + 4 * lhalf) = pk;
            }
        __builtin_amdgcn_sched_barrier(0);
        elem* zg = reinterpret_cast<elem*>(zout) + ((size_t)b * D3 + nbase) * Lp + t0;
        const int col8 = (lane & 15) * 8;
        const bool in_row = t0 + col8 < Lp;                    // Lp is a multiple of 64: whole vectors in or out
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 4 + (lane >> 4);
            const uint4 v = *reinterpret_cast<const uint4*>(zs + row * RSOUT + col8);
            if (in_row) *reinterpret_cast<uint4*>(zg + (size_t)row * Lp + col8) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int PREC>
__global__ __launch_bounds__(512) void in_proj16_kernel(GemmArgs a) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int BM = 128, K = D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);
    elem* Zs = As + BM * RS16;                              // 8 wave-private tiles [32 features][RSOUT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * BM;
    const frag* wp = reinterpret_cast<const frag*>(a.w);
    f32x16 acc[4];
    frag bs[2][1][SETK];

    load_set<PREC, K, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
    __builtin_amdgcn_sched_barrier(0);
    stage_a_tile<PREC, A_LN, K, 8>(a, As, b, t0, 0);
    __syncthreads();
    inproj_blocks<PREC>(As, Zs, wp, a.bias, a.out, b, t0, a.Lp, wave, lane, bs, acc);
}

// ================================================================================================ in_proj, gated hand-over
// The in_proj stage of the fused tail kernel in the form the long convolution wants to READ (SURVEY section 7
// "make the GEMM epilogues write what the conv kernels read"): instead of x0 | x1 | v (three rows per channel, filtered and gated
// by every convolution workgroup that reads them) it writes
//     row c        x0f[t] = short_filter(x0)[t]
//     row 256 + c  g[t]   = short_filter(x1)[t] * short_filter(v)[t]
// -- a third less z traffic in both kernels, and the 3-tap filter runs ONCE, on the fp32 accumulators (the old path filtered
// values already rounded to 16 bits).  Reference arithmetic: HyenaOperator's short_filter (Conv1d(768, 768, 3, padding=2,
// groups=768)[..., :L]) and `v * x1`.
//   short_filter(z)[t] = sb + w0 (z[t-2] + b) + w1 (z[t-1] + b) + w2 (z[t] + b),  zero padding before the read's first token
//                      = cb + w0 r[t-2] + w1 r[t-1] + w2 r[t]   with r = the raw accumulator (no bias), cb = sb + b (w0 + w1 + w2),
//                        and r[-1] = r[-2] = -b at the start of a read (so that r + b = 0 there).
// Accumulator layout (rows = tokens): lane (feature lrow, half lhalf) holds, per register pair index i = 4 mt + q', the four
// CONSECUTIVE tokens 32 mt + 8 q' + 4 lhalf + {0..3}; the two tokens before them belong to the partner lane (lane ^ 32): its pair
// i (lhalf = 1 needs lhalf = 0's) or i - 1 (lhalf = 0 needs lhalf = 1's) -- one ds_bpermute per value.  Pair 0 of the lhalf = 0
// lanes needs the PREVIOUS TILE's tokens 126, 127: tiles are taken in contiguous ranges per workgroup (tail_range_len), the
// raw values are kept in an LDS stash from tile to tile; the first tile of a range that starts inside a read is computed with
// r = -b there and its tokens 0, 1 are recomputed by gated_patch_kernel from the raw values both sides leave in edge_bnd.
// Block order x1, v, x0: x1f waits in the (dead) fc2 accumulators for v; the y prefetch of the hooks stays in the last block.
constexpr int ZG_ORDER[3] = {1, 2, 0};
constexpr int ZG_HALO_FLOATS = 8 * 3 * 32 * 2;          // [wave][q][lrow][2]: raw tokens 126, 127 of the workgroup's previous tile

struct GatedTile {                                       // uniform per tile
    bool fresh;          // no history in the stash: first token of a read, or first tile of this workgroup's range
    bool head_bnd;       // first tile of the range, inside a read: leave tokens 0, 1 raw in edge_bnd[w][1]
    bool tail_bnd;       // last tile of the range and the next tile continues the read: leave tokens 126, 127 in edge_bnd[w + 1][0]
    bool read_tail;      // last tile of a read: leave tokens 126, 127 in edge_read[b]
    int w;               // this workgroup
};

template <int PREC>
__device__ __forceinline__ void zg_fir_inplace(f32x16 (&a)[4], const float4 f, float p2, float p3, int lane) {
    const int lhalf = lane >> 5, paddr = (lane ^ 32) << 2;
    // Every lane hands its tokens 2, 3 of pair i to its partner (e2[i], e3[i] = the PARTNER's): an lhalf = 1 lane needs the
    // partner's pair i, an lhalf = 0 lane its pair i - 1 (pair 0: the stash).  The select is made on the RECEIVED values: written
    // as a select between two elements of one accumulator vector, hipcc turns it into a dynamic vector index -- a 15-deep
    // v_cndmask chain per value (364 of them per block, and 50 spilled registers).
    float e2[16], e3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // (elements copied to scalars first: __builtin_bit_cast straight from a vector element reads element 0 with hipcc 7.2 --
        //  HISTORY.md section 4.7; here it made all 32 exchanges of a block fetch four values)
        const float s2 = a[i >> 2][4 * (i & 3) + 2], s3 = a[i >> 2][4 * (i & 3) + 3];
        e2[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(s2)));
        e3[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(s3)));
    }
    float r2[16], r3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        r2[i] = lhalf ? e2[i] : (i ? e2[i ? i - 1 : 0] : p2);
        r3[i] = lhalf ? e3[i] : (i ? e3[i ? i - 1 : 0] : p3);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int mt = i >> 2, q = 4 * (i & 3);
        const float z0 = a[mt][q], z1 = a[mt][q + 1], z2 = a[mt][q + 2], z3 = a[mt][q + 3];
        a[mt][q + 0] = fmaf(f.z, z0, fmaf(f.y, r3[i], fmaf(f.x, r2[i], f.w)));
        a[mt][q + 1] = fmaf(f.z, z1, fmaf(f.y, z0, fmaf(f.x, r3[i], f.w)));
        a[mt][q + 2] = fmaf(f.z, z2, fmaf(f.y, z1, fmaf(f.x, z0, f.w)));
        a[mt][q + 3] = fmaf(f.z, z3, fmaf(f.y, z2, fmaf(f.x, z1, f.w)));
    }
}

// the wave's 128 tokens x 32 channels in `a` -> 16-bit -> row `row0 + lrow` of z (through the wave-private staging tile)
template <int PREC>
__device__ __forceinline__ void zg_store_rows(const f32x16 (&a)[4], typename CT<PREC>::elem* zs, void* zout, int b, int row0,
                                              int t0, int Lp, int lane) {
    using elem = typename CT<PREC>::elem;
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u16x4 pk = {to_bits<PREC>(a[mt][4 * q + 0]), to_bits<PREC>(a[mt][4 * q + 1]), to_bits<PREC>(a[mt][4 * q + 2]),
                        to_bits<PREC>(a[mt][4 * q + 3])};
            *reinterpret_cast<u16x4*>(zs + lrow * RSOUT + mt * 32 + 8 * q + 4 * lhalf) = pk;
        }
    __builtin_amdgcn_sched_barrier(0);
    elem* zg = reinterpret_cast<elem*>(zout) + ((size_t)b * D3 + row0) * Lp + t0;
    const int col8 = (lane & 15) * 8;
    const bool in_row = t0 + col8 < Lp;                    // Lp is a multiple of 64: whole vectors in or out
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 4 + (lane >> 4);
        const uint4 v = *reinterpret_cast<const uint4*>(zs + row * RSOUT + col8);
        if (in_row && (!lab::NOZSTORE || v.x == 0x12345678u)) *reinterpret_cast<uint4*>(zg + (size_t)row * Lp + col8) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int PREC, typename Hook>
__device__ __forceinline__ void inproj_blocks_gated(const typename CT<PREC>::elem* As, typename CT<PREC>::elem* Zs, float* halo,
                                                    const u16x8* wp, const TailArgs& m, const GatedTile gt, int b, int t0, int wave,
                                                    int lane, u16x8 (&bs)[2][1][SETK], f32x16 (&accv)[4], f32x16 (&accx)[4],
                                                    Hook hook) {
    using elem = typename CT<PREC>::elem;
    constexpr int K = D;
    const int lrow = lane & 31, lhalf = lane >> 5;
    elem* zs = Zs + wave * 32 * RSOUT;
    float4 firq = make_float4(0.f, 0.f, 0.f, 0.f);
    static_for<0, 3>([&](auto jc) {
        constexpr int j = decltype(jc)::value, q = ZG_ORDER[j], qn = ZG_ORDER[(j + 1) % 3];
        f32x16 (&acc)[4] = (j == 1) ? accv : accx;          // x1 and x0 in accx, v in accv
        zero_acc(acc);
        int lrow_e = lrow;                                   // opaque copy: addresses formed here, not kept across the MFMA phase
        asm volatile("" : "+v"(lrow_e));
        const int c = wave * 32 + lrow_e;
        // this block's filter constants: requested now, behind the weight set the first MFMAs wait for (the hook slot)
        auto hook2 = [&](int step) {
            if (step == 2 * j) firq = m.n_fir[c * 3 + q];
            hook(step);
        };
        phase_tm<PREC, K, K, false>(As, wp, q, 0, wp, qn, 0, wave, lane, bs, acc, hook2, 2 * j);
        // ---- history of the tile's first tokens
        float* hs = halo + ((wave * 3 + q) * 32 + lrow_e) * 2;
        float p2, p3;
        if (gt.fresh) {                                      // uniform; one tile in 16 .. 64
            p2 = p3 = -m.n_bias[q * 256 + c];
        } else {
            const float2 hv2 = *reinterpret_cast<const float2*>(hs);
            p2 = hv2.x, p3 = hv2.y;
        }
        const float2 tl = make_float2(acc[3][14], acc[3][15]), hd = make_float2(acc[0][0], acc[0][1]);   // raw tokens 126, 127 / 0, 1
        if (lhalf) *reinterpret_cast<float2*>(hs) = tl;      // (same wave, after the read above: LDS operations of a wave are in order)
        if (gt.tail_bnd && lhalf) m.edge_bnd[((size_t)(gt.w + 1) * 2 + 0) * D3 + q * 256 + c] = tl;
        if (gt.head_bnd && !lhalf) m.edge_bnd[((size_t)gt.w * 2 + 1) * D3 + q * 256 + c] = hd;
        if (gt.read_tail && lhalf) m.edge_read[(size_t)b * D3 + q * 256 + c] = tl;
        if constexpr (!lab::NOFIR) zg_fir_inplace<PREC>(acc, firq, p2, p3, lane);
        // x1f waits for vf through a whole MFMA phase in which both accumulators, two weight sets and the fragment ring are live
        // (hipcc spilled 8 .. 26 of its registers to scratch, behind vmcnt waits): its upper half (tokens 64 .. 127) waits in the
        // wave's staging tile instead -- unused until g is staged -- as eight conflict-free 16-byte rows per lane
        float4* xs = reinterpret_cast<float4*>(zs) + lane;
        static_assert((size_t)32 * RSOUT * sizeof(elem) >= (size_t)8 * 64 * sizeof(float4), "half of x1f fits the staging tile");
        if constexpr (j == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                xs[k * 64] = make_float4(accx[2 + (k >> 2)][4 * (k & 3) + 0], accx[2 + (k >> 2)][4 * (k & 3) + 1],
                                         accx[2 + (k >> 2)][4 * (k & 3) + 2], accx[2 + (k >> 2)][4 * (k & 3) + 3]);
            asm volatile("" ::: "memory");                   // (the tile is re-read / rewritten below through other pointer types)
        } else if constexpr (j == 1) {                       // g = x1f * vf
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accv[mt][r] *= accx[mt][r];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 x = xs[k * 64];
                accv[2 + (k >> 2)][4 * (k & 3) + 0] *= x.x;
                accv[2 + (k >> 2)][4 * (k & 3) + 1] *= x.y;
                accv[2 + (k >> 2)][4 * (k & 3) + 2] *= x.z;
                accv[2 + (k >> 2)][4 * (k & 3) + 3] *= x.w;
            }
            asm volatile("" ::: "memory");
            zg_store_rows<PREC>(accv, zs, m.n_z, b, 256 + wave * 32, t0, m.Lp, lane);
        } else {
            zg_store_rows<PREC>(accx, zs, m.n_z, b, wave * 32, t0, m.Lp, lane);
        }
    });
}

// ---- fp16c (round 4): the gated in_proj stage with the LayerNorm-1 tile as hi + lo bytes and z leaving as hi + lo -------------------
// `Al` = the lo tile of the LayerNorm-1 output (ln_acc_to_tile<.., LO>): every product gets the activations' lo term (compute_tm
// LO2: 16 more live registers).  The v block therefore runs in TWO 64-token halves -- x1f stays whole in its accumulator (no
// stash in LDS: the lo tile took that space), v's accumulator is 32 registers at a time, and v's weights stream twice (+1/13 of a
// tile's weight bytes).  Each half of g leaves as soon as it exists: halfs through the wave's staging tile as whole 128-byte lines
// per channel row, then the values' lo bytes (lo8_pack4) through the same tile (LDS operations of one wave execute in order)
// into the lo planes in rows 512.. of z ([2][256][Lp] bytes: byte row c = x0f, 256 + c = g).
constexpr int ZG_WAVE_BYTES = 4608;                       // wave tile of the LO form: 32 rows x (64 halfs + 8)

// short filter on the row tiles [MT0, MT0 + NMT): p2 / p3 = raw tokens -2, -1 of the range on entry, of the NEXT range on return
template <int MT0, int NMT>
__device__ __forceinline__ void zg_fir_range(f32x16 (&a)[4], const float4 f, float& p2, float& p3, int lane) {
    const int lhalf = lane >> 5, paddr = (lane ^ 32) << 2;
    constexpr int NP = 4 * NMT;
    float e2[NP], e3[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float s2 = a[MT0 + (i >> 2)][4 * (i & 3) + 2], s3 = a[MT0 + (i >> 2)][4 * (i & 3) + 3];   // (scalars first: see zg_fir_inplace)
        e2[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(s2)));
        e3[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(s3)));
    }
    float r2[NP], r3[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        r2[i] = lhalf ? e2[i] : (i ? e2[i ? i - 1 : 0] : p2);
        r3[i] = lhalf ? e3[i] : (i ? e3[i ? i - 1 : 0] : p3);
    }
    p2 = e2[NP - 1], p3 = e3[NP - 1];                        // (what the lhalf = 0 lanes of the next range need; unused by the others)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int mt = MT0 + (i >> 2), q = 4 * (i & 3);
        const float z0 = a[mt][q], z1 = a[mt][q + 1], z2 = a[mt][q + 2], z3 = a[mt][q + 3];
        a[mt][q + 0] = fmaf(f.z, z0, fmaf(f.y, r3[i], fmaf(f.x, r2[i], f.w)));
        a[mt][q + 1] = fmaf(f.z, z1, fmaf(f.y, z0, fmaf(f.x, r3[i], f.w)));
        a[mt][q + 2] = fmaf(f.z, z2, fmaf(f.y, z1, fmaf(f.x, z0, f.w)));
        a[mt][q + 3] = fmaf(f.z, z3, fmaf(f.y, z2, fmaf(f.x, z1, f.w)));
    }
}

// tokens [64 hf, 64 hf + 64) of the wave's 32 channels: halfs to byte-row-free z row `row0 + lrow`, lo bytes to lo-plane row `row0 + lrow`
template <int HF>
__device__ __forceinline__ void zg_store_half(const f32x16 (&a)[4], f16_t* zs, void* zout, int b, int row0, int t0, int Lp, int lane_in) {
    constexpr int RSH = 72, RS8 = 80;                        // staged rows: 64 halfs + 8; 64 lo bytes + 16 (16-byte aligned rows)
    static_assert(32 * RSH * 2 <= ZG_WAVE_BYTES && 32 * RS8 <= ZG_WAVE_BYTES, "both stagings fit the wave's tile");
    // (an opaque copy of the lane per call: the four calls of a tile share most of their address terms, and shared terms are
    //  computed once, early, and SPILLED across the MFMA phases in between -- ten dwords per tile before this line)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int lrow = lane & 31, lhalf = lane >> 5;
    unsigned lo4[8];
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mt = 2 * HF + m2;
            u16x4 pk;
            lo4[4 * m2 + q] = lo8_pack4(a[mt][4 * q + 0], a[mt][4 * q + 1], a[mt][4 * q + 2], a[mt][4 * q + 3], pk);
            *reinterpret_cast<u16x4*>(zs + lrow * RSH + m2 * 32 + 8 * q + 4 * lhalf) = pk;
        }
    __builtin_amdgcn_sched_barrier(0);
    f16_t* zg = reinterpret_cast<f16_t*>(zout) + ((size_t)b * D3 + row0) * Lp + t0 + 64 * HF;
    const bool in_row = t0 + 64 * HF < Lp;                   // Lp is a multiple of 64: the whole half in or out (uniform)
    {
        const int col8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(zs + row * RSH + col8);
            if (in_row && (!lab::NOZSTORE || v.x == 0x12345678u)) *reinterpret_cast<uint4*>(zg + (size_t)row * Lp + col8) = v;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (lab::NOZLO) return;
    unsigned char* z8 = reinterpret_cast<unsigned char*>(zs);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<unsigned*>(z8 + lrow * RS8 + m2 * 32 + 8 * q + 4 * lhalf) = lo4[4 * m2 + q];
    __builtin_amdgcn_sched_barrier(0);
    unsigned char* zl = reinterpret_cast<unsigned char*>(reinterpret_cast<f16_t*>(zout) + ((size_t)b * D3 + 2 * D) * Lp) + (size_t)row0 * Lp + t0 + 64 * HF;
    {
        const int col16 = (lane & 3) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 16 + (lane >> 2);
            const uint4 v = *reinterpret_cast<const uint4*>(z8 + row * RS8 + col16);
            if (in_row) *reinterpret_cast<uint4*>(zl + (size_t)row * Lp + col16) = v;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <typename Hook>
__device__ __forceinline__ void inproj_blocks_gated_lo(const f16_t* As, const unsigned char* Al, unsigned char* Zs, float* halo,
                                                       const u16x8* wp, const TailArgs& m, const GatedTile gt, int b, int t0, int wave,
                                                       int lane, u16x8 (&bs)[2][1][SETK], f32x16 (&accv)[4], f32x16 (&accx)[4],
                                                       Hook hook) {
    constexpr int PREC = PREC_F16C, K = D;
    const int lrow = lane & 31, lhalf = lane >> 5;
    f16_t* zs = reinterpret_cast<f16_t*>(Zs + wave * ZG_WAVE_BYTES);
    float4 firq = make_float4(0.f, 0.f, 0.f, 0.f);
    // raw-row bookkeeping of one block q: history of the tile's first tokens in, the tile's last / first raw tokens out
    auto history = [&](int q, int c, float* hs, float& p2, float& p3) {
        if (gt.fresh) {                                      // uniform; one tile in 16 .. 64
            p2 = p3 = -m.n_bias[q * 256 + c];
        } else {
            const float2 hv2 = *reinterpret_cast<const float2*>(hs);
            p2 = hv2.x, p3 = hv2.y;
        }
    };
    auto leave_tail = [&](int q, int c, float* hs, const float2 tl) {       // raw tokens 126, 127
        if (lhalf) *reinterpret_cast<float2*>(hs) = tl;      // (same wave, after the read in history(): LDS operations of a wave are in order)
        if (gt.tail_bnd && lhalf) m.edge_bnd[((size_t)(gt.w + 1) * 2 + 0) * D3 + q * 256 + c] = tl;
        if (gt.read_tail && lhalf) m.edge_read[(size_t)b * D3 + q * 256 + c] = tl;
    };
    auto leave_head = [&](int q, int c, const float2 hd) {                  // raw tokens 0, 1
        if (gt.head_bnd && !lhalf) m.edge_bnd[((size_t)gt.w * 2 + 1) * D3 + q * 256 + c] = hd;
    };
    // ---- x1 (block 1), all four row tiles, stays in accx
    {
        constexpr int q = 1;
        zero_acc(accx);
        int lrow_e = lrow;                                   // opaque copy: addresses formed here, not kept across the MFMA phase
        asm volatile("" : "+v"(lrow_e));
        const int c = wave * 32 + lrow_e;
        auto hook2 = [&](int step) {
            if (step == 0) firq = m.n_fir[c * 3 + q];
            hook(step);
        };
        phase_tm<PREC, K, K, false, false, decltype(hook2), PREC_SAME, true>(As, wp, q, 0, wp, 2, 0, wave, lane, bs, accx, hook2, 0, Al);
        float* hs = halo + ((wave * 3 + q) * 32 + lrow_e) * 2;
        float p2, p3;
        history(q, c, hs, p2, p3);
        leave_tail(q, c, hs, make_float2(accx[3][14], accx[3][15]));
        leave_head(q, c, make_float2(accx[0][0], accx[0][1]));
        if constexpr (!lab::NOFIR) zg_fir_inplace<PREC>(accx, firq, p2, p3, lane);
        // x1f is COMPLETE here: left alone hipcc sinks the filter arithmetic of the upper row tiles to their first use, behind
        // both halves of v, and carries the 16 lane-exchange results through those MFMA phases instead -- nine of them in scratch
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            asm volatile("" : "+v"(accx[mt][0]), "+v"(accx[mt][1]), "+v"(accx[mt][2]), "+v"(accx[mt][3]), "+v"(accx[mt][4]), "+v"(accx[mt][5]),
                              "+v"(accx[mt][6]), "+v"(accx[mt][7]), "+v"(accx[mt][8]), "+v"(accx[mt][9]), "+v"(accx[mt][10]),
                              "+v"(accx[mt][11]), "+v"(accx[mt][12]), "+v"(accx[mt][13]), "+v"(accx[mt][14]), "+v"(accx[mt][15]));
    }
    // ---- v (block 2) in two halves of two row tiles: g = x1f * vf leaves half by half
    {
        constexpr int q = 2;
        int lrow_e = lrow;
        asm volatile("" : "+v"(lrow_e));
        const int c = wave * 32 + lrow_e;
        float* hs = halo + ((wave * 3 + q) * 32 + lrow_e) * 2;
        float p2, p3;
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[0][r] = 0.f, accv[1][r] = 0.f;
        auto hook2 = [&](int step) {
            if (step == 2) firq = m.n_fir[c * 3 + q];
            hook(step);
        };
        phase_tm<PREC, K, K, false, false, decltype(hook2), PREC_SAME, true, 0, 2>(As, wp, q, 0, wp, q, 0, wave, lane, bs, accv, hook2, 2, Al);
        history(q, c, hs, p2, p3);
        leave_head(q, c, make_float2(accv[0][0], accv[0][1]));
        if constexpr (!lab::NOFIR) zg_fir_range<0, 2>(accv, firq, p2, p3, lane);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accv[mt][r] *= accx[mt][r];
        zg_store_half<0>(accv, zs, m.n_z, b, 256 + wave * 32, t0, m.Lp, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[2][r] = 0.f, accv[3][r] = 0.f;
        phase_tm<PREC, K, K, false, false, NoHook, PREC_SAME, true, 2, 2>(As, wp, q, 0, wp, 0, 0, wave, lane, bs, accv, NoHook(), 0, Al);
        leave_tail(q, c, hs, make_float2(accv[3][14], accv[3][15]));
        if constexpr (!lab::NOFIR) zg_fir_range<2, 2>(accv, firq, p2, p3, lane);
#pragma unroll
        for (int mt = 2; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accv[mt][r] *= accx[mt][r];
        zg_store_half<1>(accv, zs, m.n_z, b, 256 + wave * 32, t0, m.Lp, lane);
    }
    // ---- x0 (block 0), all four row tiles, in accx (x1f is dead)
    {
        constexpr int q = 0;
        zero_acc(accx);
        int lrow_e = lrow;
        asm volatile("" : "+v"(lrow_e));
        const int c = wave * 32 + lrow_e;
        auto hook2 = [&](int step) {
            if (step == 4) firq = m.n_fir[c * 3 + q];
            if constexpr (!lab::YLATE) hook(step);
        };
        phase_tm<PREC, K, K, false, false, decltype(hook2), PREC_SAME, true>(As, wp, q, 0, wp, ZG_ORDER[0], 0, wave, lane, bs, accx, hook2, 4, Al);
        if constexpr (lab::YLATE) {
            hook(4);
            hook(5);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* hs = halo + ((wave * 3 + q) * 32 + lrow_e) * 2;
        float p2, p3;
        history(q, c, hs, p2, p3);
        leave_tail(q, c, hs, make_float2(accx[3][14], accx[3][15]));
        leave_head(q, c, make_float2(accx[0][0], accx[0][1]));
        if constexpr (!lab::NOFIR) zg_fir_inplace<PREC>(accx, firq, p2, p3, lane);
        zg_store_half<0>(accx, zs, m.n_z, b, wave * 32, t0, m.Lp, lane);
        zg_store_half<1>(accx, zs, m.n_z, b, wave * 32, t0, m.Lp, lane);
    }
}

// Tokens 0, 1 of the first tile of every workgroup range that starts inside a read (inproj_blocks_gated computed them without
// their history): recomputed from the raw values either side of the boundary.  One workgroup per boundary, one thread per channel.
template <typename T>
__global__ __launch_bounds__(256) void gated_patch_kernel(TailArgs m, int grid) {
    const int range = tail_range_len(m.tiles[0], grid);
    const int w = blockIdx.x + 1, c = threadIdx.x, tile = w * range;      // (`tile`: an index into the tile list, as in tail16_kernel)
    if (tile >= m.tiles[0] || !tile_cont(m.tiles[1 + tile])) return;
    const int b = tile_b(m.tiles[1 + tile]), t0 = tile_tx(m.tiles[1 + tile]) * 128;
    float zf[3][2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 f = m.n_fir[c * 3 + q];
        const float2 tl = m.edge_bnd[((size_t)w * 2 + 0) * D3 + q * 256 + c], hd = m.edge_bnd[((size_t)w * 2 + 1) * D3 + q * 256 + c];
        zf[q][0] = fmaf(f.z, hd.x, fmaf(f.y, tl.y, fmaf(f.x, tl.x, f.w)));
        zf[q][1] = fmaf(f.z, hd.y, fmaf(f.y, hd.x, fmaf(f.x, tl.y, f.w)));
    }
    T* z = reinterpret_cast<T*>(m.n_z) + (size_t)b * D3 * m.Lp + t0;
    const unsigned x0 = (unsigned)from_float<T>(zf[0][0]).bits | ((unsigned)from_float<T>(zf[0][1]).bits << 16);
    const unsigned g = (unsigned)from_float<T>(zf[1][0] * zf[2][0]).bits | ((unsigned)from_float<T>(zf[1][1] * zf[2][1]).bits << 16);
    *reinterpret_cast<unsigned*>(z + (size_t)c * m.Lp) = x0;
    *reinterpret_cast<unsigned*>(z + (size_t)(256 + c) * m.Lp) = g;
    if (m.zlo) {                                             // fp16c: the two tokens' lo bytes (rows 512.. of z: [2][256][Lp] bytes)
        unsigned char* zl = reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(m.n_z) + ((size_t)b * D3 + 2 * D) * m.Lp) + t0;
        u16x4 hb;
        const unsigned l = lo8_pack4(zf[0][0], zf[0][1], zf[1][0] * zf[2][0], zf[1][1] * zf[2][1], hb);
        *reinterpret_cast<unsigned short*>(zl + (size_t)c * m.Lp) = (unsigned short)(l & 0xffffu);
        *reinterpret_cast<unsigned short*>(zl + (size_t)(256 + c) * m.Lp) = (unsigned short)(l >> 16);
    }
}

__global__ __launch_bounds__(256) void fir_table_kernel(const float* __restrict__ sw, const float* __restrict__ sb,
                                                        const float* __restrict__ bias, float4* __restrict__ fir) {
    const int c = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = q * 256 + c;
        const float w0 = sw[n * 3 + 0], w1 = sw[n * 3 + 1], w2 = sw[n * 3 + 2];
        fir[c * 3 + q] = make_float4(w0, w1, w2, (float)((double)sb[n] + (double)bias[n] * ((double)w0 + (double)w1 + (double)w2)));
    }
}
void launch_fir_table(const float* short_w, const float* short_b, const float* in_bias, float4* fir, hipStream_t st) {
    hipLaunchKernelGGL(fir_table_kernel, dim3(1), dim3(256), 0, st, short_w, short_b, in_bias, fir);
}

// ================================================================================================ out_proj
template <int PREC>
__global__ __launch_bounds__(512) void out_proj16_kernel(GemmArgs a) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int BM = 128, K = D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* Ys = reinterpret_cast<elem*>(smem);              // [256 channels][RSKM], as y lies in HBM
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * BM, L = a.L, Lp = a.Lp;
    const frag* wp = reinterpret_cast<const frag*>(a.w);
    f32x16 acc[4];
    frag bs[2][1][SETK];

    load_set<PREC, K, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
    load_set<PREC, K, 1>(wp, 0, 0, 1, wave, lane, bs[1]);
    __builtin_amdgcn_sched_barrier(0);
    {
        const elem* src = reinterpret_cast<const elem*>(a.a_in) + (size_t)b * D * Lp + t0;
        const int tk = (tid & 15) * 8;
        const bool in_row = t0 + tk < Lp;
        const int tkc = in_row ? tk : 0;                       // clamped, branch-free loads; masked at the LDS store
        uint4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const uint4*>(src + (size_t)((tid >> 4) + 32 * i) * Lp + tkc);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<uint4*>(Ys + ((tid >> 4) + 32 * i) * RSKM + tk) = in_row ? x[i] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    zero_acc(acc);
    phase_km<PREC, K, K>(Ys, wp, 0, 0, wp, 0, 0, wave, lane, bs, acc);
    __syncthreads();                                       // every wave is done reading Ys: reuse it as the staging tile
    resid_epilogue(a.h_out, a.bias, acc, b, t0, L, wave, lane, reinterpret_cast<float*>(smem));
}

// ================================================================================================ fused MLP
struct MlpArgs {
    float* h;                 // residual stream [B, L, 256], read (LN2) and updated in place
    const float *ln_g, *ln_b;
    const void *w1, *w2;      // packed fc1 [1024 x 256], fc2 [256 x 1024]
    const float *b1, *b2;
    int B, L;
    float eps;
};

// One workgroup per 128-token tile.  Two variants were measured and rejected (r01 notes in HISTORY.md): a persistent
// loop that prefetches the next tile's rows during the epilogue, and taking the residual as the accumulator's initial
// value through an LDS half-tile (store-only epilogue): the extra barriers / LDS traffic / register pressure cost more
// than the 1 KiB/token re-read they save (4.0 ms vs 3.25 ms per 64 reads for this stage).
template <int PREC>
__global__ __launch_bounds__(512) void mlp16_kernel(MlpArgs m) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int BM = 128, NCH = DI / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);              // LN2(h) tile   [128][RS16]
    elem* Hs = As + BM * RS16;                              // gelu(fc1) chunk [128][RS16] (256 hidden units)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * BM, L = m.L;
    const frag* w1 = reinterpret_cast<const frag*>(m.w1);
    const frag* w2 = reinterpret_cast<const frag*>(m.w2);
    f32x16 acc1[4], acc2[4];
    frag bs[2][1][SETK];

    load_set<PREC, D, 1>(w1, 0, 0, 0, wave, lane, bs[0]);
    __builtin_amdgcn_sched_barrier(0);
    {
        GemmArgs a{};
        a.h_in = m.h; a.ln_g = m.ln_g; a.ln_b = m.ln_b; a.L = L; a.eps = m.eps;
        stage_a_tile<PREC, A_LN, D, 8>(a, As, b, t0, 0);
    }
    __syncthreads();
    zero_acc(acc2);
#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
        // ---- fc1, hidden units [256 j, 256 j + 256): rows = hidden unit, cols = token
        zero_acc(acc1);
        phase_tm<PREC, D, DI, true>(As, w1, j, 0, w2, 0, j, wave, lane, bs, acc1);
        // ---- GELU -> Hs[token][hidden]: every wave must be done reading the previous chunk
        __syncthreads();
        {
            const float* b1 = m.b1 + j * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + 8 * q);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x2 g0 = gelu_tanh2(f32x2{acc1[mt][4 * q + 0] + bb.x, acc1[mt][4 * q + 1] + bb.y});
                    const f32x2 g1 = gelu_tanh2(f32x2{acc1[mt][4 * q + 2] + bb.z, acc1[mt][4 * q + 3] + bb.w});
                    u16x4 pk = {to_bits<PREC>(g0.x), to_bits<PREC>(g0.y), to_bits<PREC>(g1.x), to_bits<PREC>(g1.y)};
                    *reinterpret_cast<u16x4*>(Hs + (mt * 32 + lrow) * RS16 + wave * 32 + 8 * q + 4 * lhalf) = pk;
                }
            }
        }
        __syncthreads();
        // ---- fc2, reduction chunk j: rows = output feature, cols = token
        phase_tm<PREC, DI, D, true>(Hs, w2, 0, j, w1, j + 1 < NCH ? j + 1 : 0, 0, wave, lane, bs, acc2);
    }
    __syncthreads();                                       // As / Hs are dead: reuse them as the staging tiles
    resid_epilogue(m.h, m.b2, acc2, b, t0, L, wave, lane, reinterpret_cast<float*>(smem));
}

// ================================================================================================ score + pool
// ln_f + attention.0 (256 -> 256) + GELU(erf) + attention.2 (256 -> 1) give the pooling score of every token; the same
// staged ln_f tile then feeds this tile's share of the attention pooling as an online-softmax partial
//     m = max_t s_t,   S = sum_t exp(s_t - m),   vec[c] = sum_t exp(s_t - m) * ln_f(h_t)[c]
// (BinarySequenceClassifier.forward, /root/reference/chimeralm/models/components/hyena.py:117-146; softmax over all L
// positions).  head_tiles_kernel (head.hip) merges the per-tile partials in a fixed order.  One pass over the residual
// stream instead of two (the separate score and pool kernels each read all of h).

// Scores and the online-softmax pooling partial of one staged ln_f tile.  On entry bs[0] / bs[1] hold the two half-sets
// of attention.0.weight; P / E / V: [8][128] + [128] + [4][256] floats of LDS behind the tile.
// LO2 (fp16c in the fused tail kernel): `Al` holds the lo bytes of the ln_f tile (ln_acc_to_tile<.., LO>): the score product gets
// the activations' lo term and the pooled vector is summed from hi + lo -- ln_f's fp16 rounding was a fifth of the mode's logit
// error (round 4, tests/error_model.py).
template <int PREC, bool LO2 = false>
__device__ __forceinline__ void score_pool_tile(const ScorePoolArgs& m, const typename CT<PREC>::elem* As, float* P, int b,
                                                int tile, int tid, u16x8 (&bs)[2][1][SETK], f32x16 (&acc)[4],
                                                const unsigned char* Al = nullptr) {
    using elem = typename CT<PREC>::elem;
    constexpr int BM = 128;
    float* E = P + 8 * BM;
    float* V = E + BM;
    const int lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5, t0 = tile * BM, L = m.L;
    zero_acc(acc);
    {   // rows = features (register quads), lane = token; the last slot re-requests set 0 (unconditional prefetch, unused)
        const u16x8* wsp = reinterpret_cast<const u16x8*>(m.w1);
        phase_tm<PREC, D, D, true, true, NoHook, PREC_SAME, LO2>(As, wsp, 0, 0, wsp, 0, 0, wave, lane, bs, acc, NoHook(), 0, Al);
    }
    {
        float4 b1v[4], w2v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b1v[q] = *reinterpret_cast<const float4*>(m.b1 + wave * 32 + 8 * q + 4 * lhalf);
            w2v[q] = *reinterpret_cast<const float4*>(m.w2 + wave * 32 + 8 * q + 4 * lhalf);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s = fmaf(gelu_erf(acc[mt][4 * q + 0] + b1v[q].x), w2v[q].x, s);
                s = fmaf(gelu_erf(acc[mt][4 * q + 1] + b1v[q].y), w2v[q].y, s);
                s = fmaf(gelu_erf(acc[mt][4 * q + 2] + b1v[q].z), w2v[q].z, s);
                s = fmaf(gelu_erf(acc[mt][4 * q + 3] + b1v[q].w), w2v[q].w, s);
            }
            s += __shfl_xor(s, 32, 64);
            if (lhalf == 0) P[wave * BM + mt * 32 + lrow] = s;
        }
    }
    __syncthreads();
    if (wave == 0) {                                         // 128 tokens: lane handles token lane and lane + 64
        float sc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = lane + 64 * j;
            sc[j] = (((P[t] + P[BM + t]) + (P[2 * BM + t] + P[3 * BM + t])) +
                     ((P[4 * BM + t] + P[5 * BM + t]) + (P[6 * BM + t] + P[7 * BM + t]))) + m.b2[0];
            if (t0 + t < L) m.scores[(size_t)b * L + t0 + t] = sc[j];
            else sc[j] = -INFINITY;
        }
        const float mx = wave_max(fmaxf(sc[0], sc[1]));       // token t0 is always valid: mx is finite
        const float e0 = expf(sc[0] - mx), e1 = expf(sc[1] - mx);   // exp(-inf) = 0 for the rows past L
        E[lane] = e0;
        E[lane + 64] = e1;
        const float ssum = wave_sum(e0 + e1);
        if (lane == 0) {
            float* out = m.partial + ((size_t)b * m.ntiles + tile) * POOL_PSTRIDE + D;
            out[0] = mx;
            out[1] = ssum;
        }
    }
    __syncthreads();
    {   // vec: thread = (channel pair, 32-token group); the staged ln_f tile is read back as packed pairs
        const int c2 = tid & 127, g = tid >> 7;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int t = g * 32 + i;
            const unsigned int pk = *reinterpret_cast<const unsigned int*>(As + t * RS16 + 2 * c2);
            const float e = E[t];
            elem lo, hi;
            lo.bits = (unsigned short)(pk & 0xffffu);
            hi.bits = (unsigned short)(pk >> 16);
            float v0 = to_float(lo), v1 = to_float(hi);
            if constexpr (LO2) {                            // channels 2 c2, 2 c2 + 1: adjacent bytes of the lo row
                typedef float f2 __attribute__((ext_vector_type(2)));
                const int l8 = *reinterpret_cast<const unsigned short*>(Al + t * RSL + lo_pos(2 * c2));
                const f2 d = __builtin_amdgcn_cvt_pk_f32_bf8(l8, false);
                v0 = fmaf(d.x, LO2_INV, v0), v1 = fmaf(d.y, LO2_INV, v1);
            }
            a0 = fmaf(e, v0, a0);
            a1 = fmaf(e, v1, a1);
        }
        V[g * D + 2 * c2] = a0;
        V[g * D + 2 * c2 + 1] = a1;
    }
    __syncthreads();
    if (tid < D)
        m.partial[((size_t)b * m.ntiles + tile) * POOL_PSTRIDE + tid] = (V[tid] + V[D + tid]) + (V[2 * D + tid] + V[3 * D + tid]);
}

template <int PREC>
__global__ __launch_bounds__(512) void score_pool16_kernel(ScorePoolArgs m) {
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int BM = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);
    float* P = reinterpret_cast<float*>(As + BM * RS16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, tile = blockIdx.x, t0 = tile * BM;
    const frag* wp = reinterpret_cast<const frag*>(m.w1);
    f32x16 acc[4];
    frag bs[2][1][SETK];
    load_set<PREC, D, 1>(wp, 0, 0, 0, wave, lane, bs[0]);
    load_set<PREC, D, 1>(wp, 0, 0, 1, wave, lane, bs[1]);
    __builtin_amdgcn_sched_barrier(0);
    GemmArgs a{};
    a.h_in = m.h; a.ln_g = m.ln_g; a.ln_b = m.ln_b; a.L = m.L; a.eps = m.eps;
    stage_a_tile<PREC, A_LN, D, 8>(a, As, b, t0, 0);
    __syncthreads();
    score_pool_tile<PREC>(m, As, P, b, tile, tid, bs, acc);
}

// ================================================================================================ out_proj + MLP fused
// h_new = r + fc2(gelu(fc1(LN2(r)))),  r = h + out_proj(y^T)        (second half of a HyenaBlock, one kernel)
// The out_proj accumulator (rows = output feature, cols = token) already IS the layout the fc2 accumulator needs, so r never
// leaves the registers: it is the initial value of the fc2 accumulation, and LayerNorm-2 is computed from the
// accumulator registers (row statistics across the 8 waves through an 8 KiB LDS table).  Compared with the separate
// out_proj16 + mlp16 kernels this removes one write and two reads of the fp32 residual stream (3 KiB of 6.5 KiB per
// token), both latency-bound row phases of the MLP kernel, and one launch.

// STAMP: developer build (CLM_DEBUG=stamp) that records s_memtime at the phase boundaries of wave 0 of every workgroup into
// a side buffer nothing else reads; the product instantiation (STAMP = false) contains no stamp.
// NEXT: what follows the block on the same tile while it is still on chip -- the residual stream is then read once and
// written once per block, and the separate in_proj / score launches (latency-bound on their own) disappear.
constexpr int TAIL_NSTAMP = 28;      // 0..20 phase boundaries, 21..26 the six half-block hooks of the in_proj stage

// residual rows of tile (b, t0) in accumulator layout (block 0 of the id path: embedding rows by token id); one piece =
// the 32 token rows of one accumulator tile
__device__ __forceinline__ void tail_load_resid_piece(const TailArgs& m, float4 (&hv)[4], int mt, int b, int t0, int wave,
                                                      int lrow, int lhalf) {
    if constexpr (lab::NORESID) {      // timing-only: no residual rows (what the loads at the tile boundary cost)
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int t = t0 + mt * 32 + lrow, tc = t < m.L ? t : 0;
    const float* row = m.ids8 ? m.emb + (size_t)m.ids8[(size_t)b * m.Lp + tc] * D + wave * 32 + 4 * lhalf
                              : m.h + ((size_t)b * m.L + tc) * D + wave * 32 + 4 * lhalf;
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(row + 8 * q);
}
__device__ __forceinline__ void tail_load_resid(const TailArgs& m, float4 (&hv)[4][4], int b, int t0, int wave, int lrow,
                                                int lhalf) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) tail_load_resid_piece(m, hv[mt], mt, b, t0, wave, lrow, lhalf);
}
// hook for inproj_blocks: the four pieces of the next tile's residual, one per half-block
// ... and the next tile's y tile (8 x 16 bytes per thread) behind the last two
template <typename E>
__device__ __forceinline__ void tail_load_y_piece(const TailArgs& m, uint4 (&yx)[8], int half, int b, int t0, int tid) {
    const E* src = reinterpret_cast<const E*>(m.y) + (size_t)b * D * m.Lp + t0;
    const int tk = (tid & 15) * 8, tkc = t0 + tk < m.Lp ? tk : 0;   // clamped, branch-free; masked at the LDS store
#pragma unroll
    for (int i = 4 * half; i < 4 * half + 4; ++i)
        yx[i] = *reinterpret_cast<const uint4*>(src + (size_t)((tid >> 4) + 32 * i) * m.Lp + tkc);
}
// Round 4 (fp16c): the y tile's lo bytes, [256 channels][128 tokens] in HBM as y itself.  A thread takes 4 channels x 16 tokens
// (four 16-byte loads, then 4 x 4 byte transposes into the token-major lo tile).  Thread -> (4-channel column cg, 16-token run):
// 16 columns x 4 runs per wave -- a load instruction reads 64 contiguous bytes of 16 rows, and the 32 lanes of a ds_write_b32
// group write 16 different columns (lo_pos maps cg = 0..15 onto 16 different dwords) of two token rows 16 apart: 2-way
// conflicted.  Same-box alternations, tail kernel per step: a lane = a column, a wave = one run (64 lines per load instruction,
// conflict-free writes) 21.30 / 21.34 / 21.34 against 21.16 / 21.16 / 21.18 ms for this form; eight lanes per 128-byte line
// (8 lines per instruction, 4-way conflicted writes) was no faster than the first either.  The loads themselves are what the
// plane costs (0.85 of 1.2 ms per step, profiles/r04_timing_only.txt), whatever their shape.
// thread -> (4-channel column, 16-token run) of the y lo tile
__device__ __forceinline__ int ylo_cg(int tid) { return lab::YLO_MAP64 ? (tid & 63) : (tid & 15) + 16 * ((tid >> 6) & 3); }
__device__ __forceinline__ int ylo_tk(int tid) { return lab::YLO_MAP64 ? (tid >> 6) * 16 : 16 * (((tid >> 4) & 3) + 4 * (tid >> 8)); }
__device__ __forceinline__ void tail_load_ylo(const TailArgs& m, uint4 (&yl)[4], int b, int t0, int tid) {
    if constexpr (lab::YLO_NOLOAD) {     // timing-only: zeros instead of the four loads (garbage would be NaN bytes: another clock)
#pragma unroll
        for (int r = 0; r < 4; ++r) yl[r] = make_uint4(0, 0, 0, 0);
        return;
    }
    const int cg = ylo_cg(tid), tk = ylo_tk(tid), tkc = t0 + tk < m.Lp ? tk : 0;   // clamped, masked at the LDS store
    const unsigned char* src = m.ylo + ((size_t)b * D + 4 * cg) * m.Lp + t0 + tkc;
#pragma unroll
    for (int r = 0; r < 4; ++r) yl[r] = *reinterpret_cast<const uint4*>(src + (size_t)r * m.Lp);
}
// ... and turns them into the token-major lo tile the MFMA reads (RSL, lo_pos): per token quad a 4 x 4 byte transpose in
// registers (8 v_perm_b32), then one dword (4 consecutive channels) per token.
__device__ __forceinline__ void tail_stage_ylo(unsigned char* Aly, const uint4 (&yl)[4], int t0, int Lp, int tid) {
    const int cg = ylo_cg(tid), tk = ylo_tk(tid);
    const bool in_row = t0 + tk < Lp;
    unsigned char* dst = Aly + tk * RSL + lo_pos(4 * cg);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const unsigned d0 = a == 0 ? yl[0].x : a == 1 ? yl[0].y : a == 2 ? yl[0].z : yl[0].w;
        const unsigned d1 = a == 0 ? yl[1].x : a == 1 ? yl[1].y : a == 2 ? yl[1].z : yl[1].w;
        const unsigned d2 = a == 0 ? yl[2].x : a == 1 ? yl[2].y : a == 2 ? yl[2].z : yl[2].w;
        const unsigned d3 = a == 0 ? yl[3].x : a == 1 ? yl[3].y : a == 2 ? yl[3].z : yl[3].w;
        const unsigned x0 = __builtin_amdgcn_perm(d1, d0, 0x05010400u), x1 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);
        const unsigned y0 = __builtin_amdgcn_perm(d3, d2, 0x05010400u), y1 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
        const unsigned o[4] = {lab::YLO_NOPERM ? d0 : __builtin_amdgcn_perm(y0, x0, 0x05040100u), lab::YLO_NOPERM ? d1 : __builtin_amdgcn_perm(y0, x0, 0x07060302u),
                               lab::YLO_NOPERM ? d2 : __builtin_amdgcn_perm(y1, x1, 0x05040100u), lab::YLO_NOPERM ? d3 : __builtin_amdgcn_perm(y1, x1, 0x07060302u)};
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(dst + (4 * a + e) * RSL) = in_row ? o[e] : 0u;
    }
}
// PIECES: how many of the four residual pieces the hooks request.  fp16c: none -- with all four (64 registers) plus the y
// pieces (32) live next to the accumulators and the weight sets, hipcc spilled one piece AS IT LOADED it (four times
// `global_load_dwordx4; s_waitcnt vmcnt(0); scratch_store`: four serialised HBM round trips inside the in_proj stage of every
// tile, and a scratch reload with its own vmcnt(0) in the next tile's out_proj epilogue).  All four are requested when the
// stage's accumulators are dead (tail16_kernel, after inproj_blocks); same-box timings of the in_proj variant with 2 / 1 / 0
// pieces in the hooks: 1.437 / 1.428 / 1.407 ms per launch.
template <typename E, int PIECES, int NYL = 1>
struct ResidHook {
    const TailArgs& m;
    float4 (&hv)[4][4];
    uint4 (&yx)[8];
    uint4 (&yl)[NYL];                                      // fp16c: y's lo bytes ride with the second y piece (NYL = 4)
    int b, t0, wave, lrow, lhalf, tid;
    unsigned long long* stamp;                             // developer build only (nullptr otherwise)
    __device__ __forceinline__ void operator()(int step) const {
        if (stamp && tid == 0) stamp[21 + step] = __builtin_amdgcn_s_memtime();
        if (step < PIECES) tail_load_resid_piece(m, hv[step], step, b, t0, wave, lrow, lhalf);
        else if (step >= 4) {
            tail_load_y_piece<E>(m, yx, step - 4, b, t0, tid);
            if constexpr (NYL == 4) {
                if (step == 5) tail_load_ylo(m, yl, b, t0, tid);
            }
        }
    }
};

// MLPC (fp16c only): fc1 / fc2 on hi + lo weights as well (TailArgs::mlp_lo; w1 / w2 are then packed as PREC_F16C).
template <int PREC, bool STAMP = false, int NEXT = NEXT_NONE, bool ZG = false, bool MLPC = false>
__global__ __launch_bounds__(512) void tail16_kernel(TailArgs m, unsigned long long* stamps) {
    static_assert(!MLPC || PREC == PREC_F16C, "compensated MLP weights are a form of the fp16c mode");
    static_assert(!ZG || NEXT == NEXT_INPROJ, "the gated hand-over is a form of the fused in_proj stage");
#define CLM_STAMP_AT(k)                                                                                   \
    do {                                                                                                  \
        if (STAMP && threadIdx.x == 0)                                                                    \
            stamps[(size_t)tile * TAIL_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    using elem = typename CT<PREC>::elem;
    using frag = u16x8;
    constexpr int BM = 128, NCH = DI / 256;
    // fp16c: fc1 and fc2 run on PLAIN fp16 weights (packed as such, launch side: MLP_PREC).  The rounding of the MLP's weights
    // does not show in the logits -- rms error over 16 reads with every weight as hi + lo against fc1 and fc2 rounded to fp16,
    // five weight draws: 3.40 / 3.57, 2.29 / 2.13, 2.94 / 3.09, 1.67 / 1.79, 1.01 / 1.62 e-4, where out_proj or in_proj rounded
    // alone give 4-19e-4 (tests/error_model.py, round 3) -- and the two products are 2/3 of a tile's weight bytes, lo MFMAs and
    // operand conversions.  Round 4: true of most weight draws, not of all -- with the activations compensated the MLP weights' fp16
    // rounding was 39 % of what was left on one of the eight study draws -- so the compensated form stays available as MLPC and
    // the guard (chimeralm_amd/hyena.py) switches it on when the plain form measures above its threshold on the loaded weights.
    constexpr int PF = MLPC ? (int)PREC_F16C : MLP_PREC<PREC>;
    static_assert(std::is_same<typename CT<PF>::elem, elem>::value, "the MLP products read the same activation tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem* As = reinterpret_cast<elem*>(smem);              // LN2(r) tile [128][RS16]      (aliases Ys during out_proj)
    elem* Hs = As + BM * RS16;                              // gelu(fc1) chunk [128][RS16]
    elem* Ys = reinterpret_cast<elem*>(smem);              // y tile [256 channels][RSKM], k-major
    float* P1 = reinterpret_cast<float*>(smem + (size_t)2 * BM * RS16 * 2);   // row-sum partials [16][128]
    float* P2 = P1 + 16 * BM;                                                 // squared-deviation partials
    // bias table, filled once per workgroup: b1[1024] | b2[256] | b_out[256] | in_proj bias of the next block[768].
    // (Read from global memory each of these loads sat in front of an `s_waitcnt vmcnt(0)`: vmcnt retires in order, so the
    // epilogues waited for every weight set and residual prefetch requested before them.)
    // ZG: the in_proj bias is folded into the filter constants (n_fir); its 768 floats and the 768 behind them are the stash of the
    // previous tile's last two raw tokens (ZG_HALO_FLOATS, inproj_blocks_gated)
    float* Bt = P2 + 16 * BM;
    constexpr int BT_B2 = DI, BT_BOUT = DI + D, BT_NB = DI + 2 * D, BT_SIZE = DI + 2 * D + (ZG ? 0 : D3);
    for (int i = threadIdx.x; i < BT_SIZE; i += 512)
        Bt[i] = i < BT_B2 ? m.b1[i] : i < BT_BOUT ? m.b2[i - BT_B2] : i < BT_NB ? m.b_out[i - BT_BOUT]
                : (NEXT == NEXT_INPROJ ? m.n_bias[i - BT_NB] : 0.f);
    // Persistent workgroup over a CONTIGUOUS range of tiles (tail_range_len): consecutive tiles of a read follow each other in
    // one workgroup, which is what lets the gated in_proj stage carry the short filter's two-token history from tile to tile.
    // Round 5: "tiles" are entries of the device list m.tiles (pad_prefix.hip) -- every tile of every read when no read is padded,
    // fewer when tiles lie wholly inside [PAD] prefixes; `tile` below is an index into that list, (b, t0) come from its entry.
    const int L = m.L, Lp = m.Lp, tiles_x = (m.Lmain + BM - 1) / BM, total = m.tiles[0];
    const int* const tlist = m.tiles + 1;
    const int range = tail_range_len(total, (int)gridDim.x), tile_begin = (int)blockIdx.x * range,
              tile_end = tile_begin + range < total ? tile_begin + range : total;
    if (tile_begin >= total) return;                        // (whole workgroup, before any barrier)
    const frag* wo = reinterpret_cast<const frag*>(m.w_out);
    const frag* w1 = reinterpret_cast<const frag*>(m.w1);
    const frag* w2 = reinterpret_cast<const frag*>(m.w2);
    const frag* wn = reinterpret_cast<const frag*>(NEXT == NEXT_SCORE ? m.sp.w1 : m.n_w);
    f32x16 acc1[4], acc2[4];
    frag bs[2][1][SETK];

    // The residual rows of the NEXT tile are requested
    // before the in_proj / score stage of the current one, so that 128 KiB of the 192 KiB a tile has to pull from HBM
    // arrive under compute (one workgroup per CU: nothing else would hide them; stamps showed 20 % of a tile's time there).
    // (Starting the workgroups staggered by fractions of a tile, to spread the HBM-heavy phases of the chip over time,
    // was measured too: 2 % slower -- the phases are latency-bound per CU, not a chip-wide bandwidth burst.)
    float4 hv[4][4];                                       // residual in accumulator layout: token mt*32+lrow, 4 features
    uint4 yx[8];                                           // y tile pieces of this thread
    // fp16c, round 4: y comes as hi + lo bytes; the lo plane is staged into a token-major tile behind the y tile (under As / Hs,
    // dead during out_proj) and adds the activations' lo term to out_proj (compute_km LO2)
    constexpr bool LOY = PREC == PREC_F16C && !lab::NOYLO;
    uint4 yl[LOY ? 4 : 1];
    unsigned char* Aly = smem + (size_t)D * RSKM * 2;
    static_assert((size_t)D * RSKM * 2 % 16 == 0 && (size_t)D * RSKM * 2 + 128 * RSL <= (size_t)2 * BM * RS16 * 2, "y lo tile fits behind the y tile");
    {
        const int e0 = tlist[tile_begin], fb = tile_b(e0), ft0 = tile_tx(e0) * BM;
        tail_load_resid(m, hv, fb, ft0, (int)threadIdx.x >> 6, (int)threadIdx.x & 31, ((int)threadIdx.x >> 5) & 1);
        tail_load_y_piece<elem>(m, yx, 0, fb, ft0, threadIdx.x);
        tail_load_y_piece<elem>(m, yx, 1, fb, ft0, threadIdx.x);
        if constexpr (LOY) tail_load_ylo(m, yl, fb, ft0, threadIdx.x);
    }
#pragma unroll 1
    for (int tile = tile_begin; tile < tile_end; ++tile) {
    // the thread index is made opaque once per trip: every address below is re-derived inside the trip instead of being
    // hoisted out of the tile loop and kept (that costs ~190 spilled registers)
    int tid_l = threadIdx.x;
    asm volatile("" : "+v"(tid_l));
    // (the wave index as a scalar: every address term derived from it stays out of the vector registers)
    const int tid = tid_l, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lrow = lane & 31, lhalf = lane >> 5;
    const int entry = tlist[tile], b = tile_b(entry), tx = tile_tx(entry), t0 = tx * BM;
    CLM_STAMP_AT(0);
    if (STAMP && threadIdx.x == 0) stamps[(size_t)tile * TAIL_NSTAMP + 27] = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    // ---- 0. everything that only depends on addresses is requested first
    load_set<PREC, D, 1>(wo, 0, 0, 0, wave, lane, bs[0]);
    load_set<PREC, D, 1>(wo, 0, 0, 1, wave, lane, bs[1]);
    // (gathering these rows as whole 128-byte lines through a wave-private LDS transpose was measured: no faster -- the
    // phase is bound by how many misses one workgroup per CU keeps in flight, not by the address unit)
    __builtin_amdgcn_sched_barrier(0);
    {   // y tile: requested during the previous tile's in_proj (or before the loop) -> LDS, k-major
        const int tk = (tid & 15) * 8;
        const bool in_row = t0 + tk < Lp;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<uint4*>(Ys + ((tid >> 4) + 32 * i) * RSKM + tk) = in_row ? yx[i] : make_uint4(0, 0, 0, 0);
    }
    // (staged BEFORE the two weight-set requests above, to free its 16 registers earlier, hipcc spills more, not less: 16 vs 12 bytes)
    if constexpr (LOY) tail_stage_ylo(Aly, yl, t0, Lp, tid);
    __syncthreads();
    CLM_STAMP_AT(1);
    // ---- 1. r = h + b_out + out_proj(y) (kept in acc2): the accumulators START from the residual rows, so their 64 registers
    // are free again before the first MFMA instead of staying live through the phase (where hipcc spilled a quad of them at
    // the end of the previous tile, behind an s_waitcnt vmcnt(0)).  The rows were requested a tile ago and the weight sets
    // requested after them cannot be waited for before they land anyway (vmcnt retires in order).
    {
        const float* bo = Bt + BT_BOUT + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bo + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc2[mt][4 * q + 0] = hv[mt][q].x + bb.x;
                acc2[mt][4 * q + 1] = hv[mt][q].y + bb.y;
                acc2[mt][4 * q + 2] = hv[mt][q].z + bb.z;
                acc2[mt][4 * q + 3] = hv[mt][q].w + bb.w;
            }
        }
    }
    phase_km<PREC, D, D, PF, LOY && !lab::YLO_NOMFMA>(Ys, wo, 0, 0, w1, 0, 0, wave, lane, bs, acc2, Aly);   // (first fc1 set requested under the last set)
    CLM_STAMP_AT(2);
    // ---- 2./3. LayerNorm-2 of r straight from the accumulators -> As (16-bit)
    ln_acc_to_tile<PREC>(acc2, P1, P2, m.ln_g, m.ln_b, m.eps, As, t0, L, wave, lrow, lhalf);
    CLM_STAMP_AT(3);
    CLM_STAMP_AT(4);
    // ---- 4. MLP chunks (acc2 already holds r)
#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
        zero_acc(acc1);
        phase_tm<PF, D, DI, true>(As, w1, j, 0, w2, 0, j, wave, lane, bs, acc1);
        // GELU biases first, then the second fc2 half-set: it streams in while the workgroup is in its VALU-only GELU
        // phase and the L2 -> CU path is otherwise idle (the MFMA phases are bound by exactly that path)
        float4 b1v[4];
        {
            const float* b1 = Bt + j * 256 + wave * 32 + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; ++q) b1v[q] = *reinterpret_cast<const float4*>(b1 + 8 * q);
        }
        load_set<PF, DI, 1>(w2, 0, j, 1, wave, lane, bs[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        CLM_STAMP_AT(5 + 3 * j);
        {
            // the row address is re-derived here from an opaque copy of the lane: kept from the top of the tile it was spilled
            // (fp16c), and its reload -- a vector memory load -- put an `s_waitcnt vmcnt(0)` in front of the first store of this
            // VALU-only phase, i.e. a wait for the fc2 half-set just requested
            int lane_g = lane;
            asm volatile("" : "+v"(lane_g));
            elem* hrow = Hs + (lane_g & 31) * RS16 + wave * 32 + 4 * (lane_g >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = b1v[q];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x2 g0 = gelu_tanh2(f32x2{acc1[mt][4 * q + 0] + bb.x, acc1[mt][4 * q + 1] + bb.y});
                    const f32x2 g1 = gelu_tanh2(f32x2{acc1[mt][4 * q + 2] + bb.z, acc1[mt][4 * q + 3] + bb.w});
                    u16x4 pk = {to_bits<PREC>(g0.x), to_bits<PREC>(g0.y), to_bits<PREC>(g1.x), to_bits<PREC>(g1.y)};
                    *reinterpret_cast<u16x4*>(hrow + mt * 32 * RS16 + 8 * q) = pk;
                }
            }
        }
        __syncthreads();
        CLM_STAMP_AT(6 + 3 * j);
        // last slot: next fc1 set; on the last trip the first set of what follows (wrap-around keeps the prefetch unconditional)
        // (ZG: the in_proj stage starts with block ZG_ORDER[0])
        // (the successor's packing depends on j: its first set is addressed here, requested as raw fragments there)
        const frag* nxt = (NEXT != NEXT_NONE && j + 1 == NCH) ? set_base<PREC, D>(wn, ZG ? ZG_ORDER[0] : 0, 0, 0, wave, lane)
                                                              : set_base<PF, D>(w1, j + 1 < NCH ? j + 1 : 0, 0, 0, wave, lane);
        phase_tm<PF, DI, D, true, true, NoHook, PREC_RAWNEXT>(Hs, w2, 0, j, nxt, 0, 0, wave, lane, bs, acc2);
        CLM_STAMP_AT(7 + 3 * j);
    }
    __syncthreads();                                       // As / Hs are dead: reuse them as the staging tiles
    CLM_STAMP_AT(17);
    // ---- 5. h_new = acc2 + b2: transposed through LDS, stored as whole 128-byte lines (no read)
    {
        const float* b2p = Bt + BT_B2 + wave * 32 + 4 * lhalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(b2p + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc2[mt][4 * q + 0] += bb.x;
                acc2[mt][4 * q + 1] += bb.y;
                acc2[mt][4 * q + 2] += bb.z;
                acc2[mt][4 * q + 3] += bb.w;
            }
        }
        // (the last block's output is consumed on chip by the score / pooling stage below and by nothing else: not stored)
        if constexpr (NEXT != NEXT_SCORE) {
            float* rs = reinterpret_cast<float*>(smem) + wave * (128 * 32);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int tok = mt * 32 + lrow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chunk = (2 * q + lhalf) ^ (tok & 7);
                    *reinterpret_cast<float4*>(rs + tok * 32 + 4 * chunk) =
                        make_float4(acc2[mt][4 * q + 0], acc2[mt][4 * q + 1], acc2[mt][4 * q + 2], acc2[mt][4 * q + 3]);
                }
            }
            const int c = lane & 7, rsub = lane >> 3;
            float* hrow = m.h + ((size_t)b * L + t0) * D + wave * 32 + 4 * c;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tr = i * 8 + rsub;
                const float4 a = *reinterpret_cast<const float4*>(rs + tr * 32 + 4 * (c ^ (tr & 7)));
                if (t0 + tr < L) *reinterpret_cast<float4*>(hrow + (size_t)tr * D) = a;
            }
        }
    }
    CLM_STAMP_AT(18);
    // residual rows of this workgroup's next tile (clamped to the current one on the last trip: unconditional loads)
    const int nentry = tlist[tile + 1 < tile_end ? tile + 1 : tile];
    const int nb_ = tile_b(nentry), nt0 = tile_tx(nentry) * BM;
    if constexpr (NEXT == NEXT_NONE) {
        tail_load_resid(m, hv, nb_, nt0, wave, lrow, lhalf);
        tail_load_y_piece<elem>(m, yx, 0, nb_, nt0, tid);
        tail_load_y_piece<elem>(m, yx, 1, nb_, nt0, tid);
        if constexpr (LOY) tail_load_ylo(m, yl, nb_, nt0, tid);
    }
    // ---- 6. what follows, on the tile still in registers
    if constexpr (NEXT != NEXT_NONE) {
        load_set<PREC, D, 1>(wn, ZG ? ZG_ORDER[0] : 0, 0, 1, wave, lane, bs[1]);
        __builtin_amdgcn_sched_barrier(0);
        // (the first barrier inside orders the staging reads above before the As writes)
        // fp16c, round 4: the normalised tile as hi + lo bytes.  Score variant: the lo tile sits in the Hs region behind the 8.5 KiB
        // of P / E / V.  Gated in_proj variant: at the head of the Hs region; the eight 6-KiB wave tiles of the stage (x1f stash,
        // z staging) follow it, over the rest of Hs and the by then dead LayerNorm tables.
        constexpr bool LOF = PREC == PREC_F16C && (NEXT == NEXT_SCORE || ZG);
        unsigned char* Alf = reinterpret_cast<unsigned char*>(Hs) + (NEXT == NEXT_SCORE ? 16384 : 0);
        static_assert(16384 >= (8 * 128 + 128 + 4 * D) * 4 && 16384 + 128 * RSL <= 128 * RS16 * 2, "lo tile of ln_f fits the Hs region");
        static_assert((size_t)128 * RSL + 8 * ZG_WAVE_BYTES <= (size_t)128 * RS16 * 2 + (size_t)2 * 16 * 128 * 4,
                      "LayerNorm-1 lo tile + the gated stage's wave tiles fit Hs + the (by then dead) LayerNorm tables");
        ln_acc_to_tile<PREC, false, LOF>(acc2, P1, P2, NEXT == NEXT_SCORE ? m.sp.ln_g : m.n_g, NEXT == NEXT_SCORE ? m.sp.ln_b : m.n_b,
                                         m.eps, As, t0, L, wave, lrow, lhalf, Alf);
        CLM_STAMP_AT(19);
        if constexpr (NEXT == NEXT_INPROJ) {
            // bs[1] is re-requested by the block loop (same addresses, L2-resident): keeps the loop identical to in_proj16
            // the 128 KiB of residual rows trickle in as four pieces behind the weight requests of the first four half-blocks
            // (requested in one go before the LayerNorm they stalled every later load of the stage: +9k cycles)
            // (ZG: none in the hooks in any mode -- x1f occupies the fc2 accumulators while v is computed)
            constexpr int PIECES = (PREC == PREC_F16C || ZG) ? 0 : 4;
            if constexpr (ZG) {
                // (history: from the previous LIST entry where that is the previous tile of the same read -- tile_cont)
                const bool cont = tile_cont(entry);
                const GatedTile gt{tile == tile_begin || !cont, tile == tile_begin && cont,
                                   tile + 1 == tile_end && tile + 1 < total && tile_cont(tlist[tile + 1 < total ? tile + 1 : tile]),
                                   tx == tiles_x - 1 && m.edge_read != nullptr, (int)blockIdx.x};
                const ResidHook<elem, PIECES, LOY ? 4 : 1> rhook{m, hv, yx, yl, nb_, nt0, wave, lrow, lhalf, tid,
                                                                 STAMP ? stamps + (size_t)tile * TAIL_NSTAMP : nullptr};
                if constexpr (PREC == PREC_F16C)
                    inproj_blocks_gated_lo(As, Alf, Alf + 128 * RSL, Bt + BT_NB, wn, m, gt, b, t0, wave, lane, bs, acc1, acc2, rhook);
                else
                    inproj_blocks_gated<PREC>(As, Hs, Bt + BT_NB, wn, m, gt, b, t0, wave, lane, bs, acc1, acc2, rhook);
            } else
            inproj_blocks<PREC>(As, Hs, wn, Bt + BT_NB, m.n_z, b, t0, Lp, wave, lane, bs, acc1,
                                ResidHook<elem, PIECES, LOY ? 4 : 1>{m, hv, yx, yl, nb_, nt0, wave, lrow, lhalf, tid,
                                                                     STAMP ? stamps + (size_t)tile * TAIL_NSTAMP : nullptr});
#pragma unroll
            for (int mt = PIECES; mt < 4; ++mt) tail_load_resid_piece(m, hv[mt], mt, nb_, nt0, wave, lrow, lhalf);
        } else {
            score_pool_tile<PREC, LOF>(m.sp, As, reinterpret_cast<float*>(Hs), b, tx, tid, bs, acc1, Alf);
            // (requested before the score stage these 96 registers spill through its erf epilogue: one launch in four)
            tail_load_resid(m, hv, nb_, nt0, wave, lrow, lhalf);
            tail_load_y_piece<elem>(m, yx, 0, nb_, nt0, tid);
            tail_load_y_piece<elem>(m, yx, 1, nb_, nt0, tid);
            if constexpr (LOY) tail_load_ylo(m, yl, nb_, nt0, tid);
        }
        CLM_STAMP_AT(20);
    }
    __syncthreads();                                       // every wave is done with the tiles in LDS before the next y tile lands
    }
#undef CLM_STAMP_AT
}

// ================================================================================================ launchers
template <typename Kern, typename Args>
static void launch16_inst(Kern kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, const Args& args) {
    CLM_SET_LDS(kern, lds);                                  // (one static per instantiation of this launcher = per kernel)
    hipLaunchKernelGGL(kern, grid, block, lds, st, args);
}
// one instantiation per 16-bit arithmetic mode (bf16, fp16, fp16 with hi + lo weights)
#define CLM_LAUNCH16(prec, KERN, grid, block, lds, st, args)                                  \
    do {                                                                                      \
        if ((prec) == PREC_BF16) launch16_inst(KERN<PREC_BF16>, grid, block, lds, st, args);  \
        else if ((prec) == PREC_F16C) launch16_inst(KERN<PREC_F16C>, grid, block, lds, st, args); \
        else launch16_inst(KERN<PREC_F16>, grid, block, lds, st, args);                       \
    } while (0)

void launch_inproj16(int prec, const float* h, const float* g, const float* bta, const void* w, const float* bias,
                     void* z, int B, int L, int Lp, float eps, hipStream_t st) {
    GemmArgs a{};
    a.h_in = h; a.ln_g = g; a.ln_b = bta; a.w = w; a.bias = bias; a.out = z; a.B = B; a.L = L; a.Lp = Lp; a.eps = eps;
    constexpr size_t lds = (size_t)(128 * RS16 + 8 * 32 * RSOUT) * 2;
    dim3 grid((L + 127) / 128, B), block(512);
    CLM_LAUNCH16(prec, in_proj16_kernel, grid, block, lds, st, a);
}

void launch_outproj16(int prec, const void* y, const void* w, const float* bias, float* h, int B, int L, int Lp,
                      hipStream_t st) {
    GemmArgs a{};
    a.a_in = y; a.w = w; a.bias = bias; a.h_out = h; a.B = B; a.L = L; a.Lp = Lp;
    constexpr size_t lds = (size_t)8 * 128 * 32 * 4;       // staging tiles of the residual epilogue (>= 256*RSKM*2)
    static_assert(lds >= (size_t)D * RSKM * 2, "k-major tile must fit");
    dim3 grid((L + 127) / 128, B), block(512);
    CLM_LAUNCH16(prec, out_proj16_kernel, grid, block, lds, st, a);
}

void launch_mlp16(int prec, float* h, const float* g, const float* bta, const void* w1, const float* b1, const void* w2,
                  const float* b2, int B, int L, float eps, hipStream_t st) {
    MlpArgs m{h, g, bta, w1, w2, b1, b2, B, L, eps};
    constexpr size_t lds = (size_t)2 * 128 * RS16 * 2;
    dim3 grid((L + 127) / 128, B), block(512);
    // (fp16c: LayerNorm-2 + fc1 + GELU + fc2 is the plain fp16 kernel on fp16-packed weights, as inside tail16_kernel)
    CLM_LAUNCH16(prec == PREC_F16C ? (int)PREC_F16 : prec, mlp16_kernel, grid, block, lds, st, m);
}

// developer stamps (CLM_DEBUG=stamp): per-phase mean cycles of wave 0 over all workgroups, printed by clm_destroy
static unsigned long long* s_stamp_buf = nullptr;
static size_t s_stamp_wgs = 0;
void tail16_dump_stamps() {
    if (!s_stamp_buf || !s_stamp_wgs) return;
    std::vector<unsigned long long> hst(s_stamp_wgs * TAIL_NSTAMP);
    if (hipMemcpy(hst.data(), s_stamp_buf, hst.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    double sum[TAIL_NSTAMP] = {};
    size_t n = 0;
    for (size_t w = 0; w < s_stamp_wgs; ++w) {
        const unsigned long long* p = &hst[w * TAIL_NSTAMP];
        if (!p[0] || !p[20]) continue;
        for (int k = 1; k <= 20; ++k) sum[k] += double(p[k] - p[k - 1]);
        ++n;
    }
    const char* names[21] = {"", "issue+y_stage", "out_proj", "ln2", "-", "fc1.0", "gelu.0", "fc2.0", "fc1.1", "gelu.1",
                             "fc2.1", "fc1.2", "gelu.2", "fc2.2", "fc1.3", "gelu.3", "fc2.3", "barrier", "epilogue", "next_ln", "next_inproj"};
    double tot = 0;
    for (int k = 1; k <= 20; ++k) tot += sum[k] / (n ? n : 1);
    std::fprintf(stderr, "[tail16 stamps] %zu workgroups, mean s_memtime ticks per phase (total %.0f):\n", n, tot);
    for (int k = 1; k <= 20; ++k) std::fprintf(stderr, "  %-14s %9.0f  %5.1f %%\n", names[k], sum[k] / (n ? n : 1), 100.0 * sum[k] / (n ? n : 1) / tot);
    // in_proj stage by half-block hook: ln end -> hook 0 -> ... -> hook 5 -> end
    double hs[7] = {};
    for (size_t w = 0; w < s_stamp_wgs; ++w) {
        const unsigned long long* p = &hst[w * TAIL_NSTAMP];
        if (!p[0] || !p[20] || !p[21]) continue;
        hs[0] += double(p[21] - p[19]);
        for (int k = 1; k < 6; ++k) hs[k] += double(p[21 + k] - p[20 + k]);
        hs[6] += double(p[20] - p[26]);
    }
    // shader clock during the kernel: s_memtime ticks per s_memrealtime tick (constant 100 MHz) between consecutive tiles of
    // one workgroup (tile w and w + grid)
    {
        const int range = tail_range_len((int)s_stamp_wgs, tail16_grid((int)s_stamp_wgs));
        double dm = 0, dr = 0;
        for (size_t w = 0; w + 1 < s_stamp_wgs; ++w) {
            if ((w + 1) % range == 0) continue;             // w + 1 starts another workgroup's range
            const unsigned long long *p = &hst[w * TAIL_NSTAMP], *q = &hst[(w + 1) * TAIL_NSTAMP];
            if (!p[0] || !q[0] || !p[27] || !q[27]) continue;
            dm += double(q[0] - p[0]);
            dr += double(q[27] - p[27]);
        }
        if (dr > 0) std::fprintf(stderr, "[tail16 stamps] s_memtime / s_memrealtime = %.3f -> %.0f MHz if s_memtime is the shader clock\n", dm / dr, dm / dr * 100.0);
    }
    std::fprintf(stderr, "[tail16 stamps] in_proj stage at the hooks:");
    for (int k = 0; k < 7; ++k) std::fprintf(stderr, " %.0f", hs[k] / (n ? n : 1));
    std::fprintf(stderr, "\n");
}

template <int PREC, int NEXT, bool ZG = false, bool MLPC = false>
static void launch_tail_inst(const TailArgs& m, dim3 grid, size_t lds, hipStream_t st) {
    if (ZG) lds += (size_t)(ZG_HALO_FLOATS - D3) * 4;        // the stash takes the in_proj bias table's place and 3 KiB more
    CLM_SET_LDS((tail16_kernel<PREC, false, NEXT, ZG, MLPC>), lds);
    hipLaunchKernelGGL((tail16_kernel<PREC, false, NEXT, ZG, MLPC>), grid, dim3(512), lds, st, m, (unsigned long long*)nullptr);
}

static int tail_cus() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}
int tail16_grid(int total_tiles) { return total_tiles < tail_cus() ? total_tiles : tail_cus(); }

void launch_gated_patch(int prec, const TailArgs& m, hipStream_t st) {
    // (the grid of the tail launch this patches: sized for every tile of every read; the list may hold fewer -- the ranges follow it)
    const int tiles_x = (m.Lmain + 127) / 128, grid = tail16_grid(tiles_x * m.B);
    if (grid < 2) return;
    if (prec == PREC_BF16) hipLaunchKernelGGL(gated_patch_kernel<bf16_t>, dim3(grid - 1), dim3(256), 0, st, m, grid);
    else hipLaunchKernelGGL(gated_patch_kernel<f16_t>, dim3(grid - 1), dim3(256), 0, st, m, grid);
}

void launch_tail16(int prec, const TailArgs& m, int next, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * 128 * RS16 * 2 + (size_t)2 * 16 * 128 * 4 + (size_t)(DI + 2 * D + D3) * 4;
    static_assert(lds + (size_t)(ZG_HALO_FLOATS - D3) * 4 <= 160 * 1024, "tail kernel LDS (gated form: + the history stash)");
    static_assert((size_t)D * RSKM * 2 <= (size_t)2 * 128 * RS16 * 2, "y tile must fit under the partial tables");
    static_assert((size_t)8 * 32 * RSOUT * 2 <= (size_t)128 * RS16 * 2 + (size_t)2 * 16 * 128 * 4,
                  "in_proj staging tiles must fit in the Hs region + the (by then dead) LayerNorm tables");
    const int total = ((m.Lmain + 127) / 128) * m.B;
    dim3 grid(tail16_grid(total)), block(512);                 // persistent: one workgroup per CU (LDS-limited anyway)
    const bool zg = m.zg != 0 && next == NEXT_INPROJ;
    static const bool stamp = debug_flag("stamp");
    if (stamp && prec == PREC_F16C && next == NEXT_INPROJ) {   // developer build: the in_proj variant of the benched mode, stamped
        const size_t wgs = (size_t)total;
        if (wgs > s_stamp_wgs) {
            if (s_stamp_buf) (void)hipFree(s_stamp_buf);
            (void)hipMalloc((void**)&s_stamp_buf, wgs * TAIL_NSTAMP * 8);
            s_stamp_wgs = wgs;
        }
        (void)hipMemsetAsync(s_stamp_buf, 0, wgs * TAIL_NSTAMP * 8, st);
        if (zg) {
            constexpr size_t ldz = lds + (size_t)(ZG_HALO_FLOATS - D3) * 4;
            CLM_SET_LDS((tail16_kernel<PREC_F16C, true, NEXT_INPROJ, true>), ldz);
            hipLaunchKernelGGL((tail16_kernel<PREC_F16C, true, NEXT_INPROJ, true>), grid, block, ldz, st, m, s_stamp_buf);
        } else {
            CLM_SET_LDS((tail16_kernel<PREC_F16C, true, NEXT_INPROJ>), lds);
            hipLaunchKernelGGL((tail16_kernel<PREC_F16C, true, NEXT_INPROJ>), grid, block, lds, st, m, s_stamp_buf);
        }
        return;
    }
    if (prec == PREC_F16C && m.mlp_lo) {                       // fc1 / fc2 on hi + lo weights too (the guard's second level)
        if (zg) launch_tail_inst<PREC_F16C, NEXT_INPROJ, true, true>(m, grid, lds, st);
        else if (next == NEXT_INPROJ) launch_tail_inst<PREC_F16C, NEXT_INPROJ, false, true>(m, grid, lds, st);
        else if (next == NEXT_SCORE) launch_tail_inst<PREC_F16C, NEXT_SCORE, false, true>(m, grid, lds, st);
        else launch_tail_inst<PREC_F16C, NEXT_NONE, false, true>(m, grid, lds, st);
        return;
    }
    if (zg) {
        if (prec == PREC_BF16) launch_tail_inst<PREC_BF16, NEXT_INPROJ, true>(m, grid, lds, st);
        else if (prec == PREC_F16C) launch_tail_inst<PREC_F16C, NEXT_INPROJ, true>(m, grid, lds, st);
        else launch_tail_inst<PREC_F16, NEXT_INPROJ, true>(m, grid, lds, st);
        return;
    }
    if (prec == PREC_BF16) {
        if (next == NEXT_INPROJ) launch_tail_inst<PREC_BF16, NEXT_INPROJ>(m, grid, lds, st);
        else if (next == NEXT_SCORE) launch_tail_inst<PREC_BF16, NEXT_SCORE>(m, grid, lds, st);
        else launch_tail_inst<PREC_BF16, NEXT_NONE>(m, grid, lds, st);
    } else if (prec == PREC_F16C) {
        if (next == NEXT_INPROJ) launch_tail_inst<PREC_F16C, NEXT_INPROJ>(m, grid, lds, st);
        else if (next == NEXT_SCORE) launch_tail_inst<PREC_F16C, NEXT_SCORE>(m, grid, lds, st);
        else launch_tail_inst<PREC_F16C, NEXT_NONE>(m, grid, lds, st);
    } else {
        if (next == NEXT_INPROJ) launch_tail_inst<PREC_F16, NEXT_INPROJ>(m, grid, lds, st);
        else if (next == NEXT_SCORE) launch_tail_inst<PREC_F16, NEXT_SCORE>(m, grid, lds, st);
        else launch_tail_inst<PREC_F16, NEXT_NONE>(m, grid, lds, st);
    }
}

void launch_score_pool16(int prec, const float* h, const float* g, const float* bta, const void* w1, const float* b1,
                         const float* w2, const float* b2, float* scores, float* partial, int B, int L, float eps,
                         hipStream_t st) {
    const int ntiles = (L + 127) / 128;
    ScorePoolArgs m{h, g, bta, w1, b1, w2, b2, scores, partial, B, L, ntiles, eps};
    constexpr size_t lds = (size_t)128 * RS16 * 2 + (size_t)(8 * 128 + 128 + 4 * D) * 4;
    dim3 grid(ntiles, B), block(512);
    CLM_LAUNCH16(prec, score_pool16_kernel, grid, block, lds, st, m);
}

}  // namespace clm
