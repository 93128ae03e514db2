// clm_api.hip -- C ABI (include/chimeralm_hip.h) and host-side orchestration of the forward pass.
// The call sequence mirrors the reference protocol HyenaDna.forward -> backbone -> head
// (/root/reference/chimeralm/models/components/hyena.py:244-256); the operator order inside is the
// HyenaDNA block order of SURVEY.md section 8(a) rows 5-13.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <algorithm>
#include <string>
#include <vector>

#include "chimeralm_hip.h"
#include "clm_common.h"

using namespace clm;

namespace clm {
// "name" present in the comma-separated CLM_DEBUG list?  Read from the environment at every call (handles created one after the other
// in one process may differ); the per-launch users cache their answer.
bool debug_flag(const char* name) {
    const char* e = std::getenv("CLM_DEBUG");
    if (!e) return false;
    const size_t n = std::strlen(name);
    for (const char* p = e; *p;) {
        const char* q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        if (len == n && std::strncmp(p, name, n) == 0) return true;
        p += len + (q ? 1 : 0);
    }
    return false;
}
}  // namespace clm

namespace {

struct Tensor {
    float* d = nullptr;
    std::vector<int64_t> shape;
    size_t numel = 0;
    bool loaded = false;
};

// The filters of the four layers for one CLASS of lengths, built once and kept: the implicit filter's taps do not depend on L
// (k[t] is a function of t and the weights), and the spectrum of the first N/2 + 1 taps serves every L that runs through the
// N-point transform (L <= N/2 + 1: extra taps only reach outputs beyond L and nothing aliases, except the one product the
// convolution kernel already removes at L = N/2 + 1).  key = log2 N for single-shot lengths; KEY_LONG for L > 8193: the
// partition spectra K'_j of ALL taps (every long L uses a prefix of them).  Ragged real-world batches therefore never
// rebuild a filter (a rebuild per batch cost 3x at the reference's default batch of 12).
constexpr int KEY_LONG = 100;
struct ReversedFilter {           // conv_lone_tail(L): [256][stride] per layer, one per such L (16385, 24577, 32769)
    int L = 0, stride = 0;
    float* p[NLAYER] = {};
};
struct FilterSet {
    int key = 0, Lf = 0, logn = 0, KS = 1;   // Lf taps; KS partition spectra per channel in kf ([256][KS][N])
    float* ktime[NLAYER] = {};
    float2* kf[NLAYER] = {};
    float2* tw = nullptr;
    float2* kfp[NLAYER] = {};     // 16384-point class only: kf lane-packed for the persistent kernel (launch_spectrum_lanepack)
    std::vector<ReversedFilter> krev;
};

struct ProfRec {
    int stage;
    hipEvent_t e0, e1;
};

std::string g_create_error;

}  // namespace

struct clm_handle {
    clm_config cfg{};
    int device = 0;
    std::string err;
    std::map<std::string, Tensor> w;
    bool finalized = false;
    // packed / derived weights
    void* packed[NLAYER][4] = {};
    void* packed_score = nullptr;
    // 16-bit handles: exact-fp32 packing of the same weights (fp16c's reads shorter than f16c_min_len, clm_selfcheck, clm_set_fallback)
    void* packed32[NLAYER][4] = {};
    void* packed32t[NLAYER][4] = {};   // exact fp32, the fused tail's packing (tail32.hip): in_proj, out_proj, fc1, fc2
    // the same weights as hi + lo halfs (launch_pack_x3; tail32.hip AR_X3): the arithmetic of a CLM_PREC_F16X3 handle AND, round 5,
    // of every 16-bit handle's short reads and first fall-back level -- lwx = the handle's fp32-path LayerW with these
    void* packed32x[NLAYER][4] = {};
    LayerW lwx[NLAYER] = {};
    bool referee = false;              // inside clm_selfcheck's second pass: exact fp32, whatever the handle's mode or fall-back level
    void* packed_score32 = nullptr;
    void *packed_score32t = nullptr, *packed_score32x = nullptr;   // attention.0.weight in the fused tail's packings (tail32.hip T32_SCORE)
    LayerW lw32[NLAYER]{};
    // PREC_F16C: fc1 / fc2 packed as hi + lo as well (the mode's second level, clm_set_mlp_compensation; lw.w_fc1 / w_fc2 are plain fp16)
    void* packed_mlpc[NLAYER][2] = {};
    bool mlp_lo = false;
    int f16c_min_len = 2048;
    // clm_selfcheck / clm_set_fallback: the exact-fp32 kernels of the same handle as referee of, and replacement for, the 16-bit path
    int force_prec = -1;          // >= 0 inside clm_selfcheck: the arithmetic forward_chunk runs in, whatever the length
    // clm_set_fallback: 0 = the handle's own mode; 1 = the next arithmetic INSIDE the gate (a 16-bit handle: fp16x3 -- fp32-class
    // results at about twice the exact rate; an fp16x3 handle: exact fp32); 2 = exact fp32 on every handle
    int fallback = 0;
    float* sc_logits = nullptr;   // [2][sc_cap][2] device: logits of the two passes of a self-check
    int sc_cap = 0;
    // host batches: two device staging buffers fed by the handle's own copy stream
    struct Stage {
        void* buf = nullptr;
        size_t cap = 0;
        int dtype = 0, B = 0, L = 0;
        int64_t stride = 0;
        hipEvent_t copied = nullptr, consumed = nullptr;
        bool used = false, pending = false;
    } stage[2];
    hipStream_t copy_stream = nullptr;
    int next_stage = 0;
    int* bad_ids = nullptr;       // host-mapped flag the id kernels set for a token id outside [0, vocab_rows)
    float* ztab = nullptr;        // [16][768] block-0 in_proj rows per token id (16-bit modes)
    // gated hand-over of z (TailArgs::zg): filter constants per layer, raw rows either side of the tail kernel's workgroup-range
    // boundaries, raw rows of every read's last two tiled tokens
    float4* fir[NLAYER] = {};
    float2* edge_bnd = nullptr;
    float2* edge_read = nullptr;
    int edge_read_cap = 0;
    bool raw_z = false;           // CLM_DEBUG=raw_z: the fused in_proj stage writes x0 | x1 | v as before round 3 (A/B runs, tests)
    unsigned char* ids8 = nullptr;   // workspace: clamped ids [B][Lp]
    float* head_t[5] = {};
    LayerW lw[NLAYER]{};
    HeadW hw{};
    std::vector<FilterSet> filters;
    uint64_t clock = 0;
    // workspace (one chunk of reads)
    size_t ws_cap[15] = {};       // bytes of each workspace buffer (ensure_workspace: WS_H .. WS_TILES)
    size_t ws_es = 0;             // element size z / y were last written with
    float* h = nullptr;
    void *z = nullptr, *y = nullptr, *u = nullptr;
    float *scores = nullptr, *stats = nullptr, *partial = nullptr, *pooled = nullptr, *lone_ws = nullptr;
    float2* gscratch = nullptr;                     // segment spectra of the long-read convolution
    unsigned char* ylo = nullptr;                   // PREC_F16C: lo bytes of y [B][256][Lp] (round 4, clm_common.h lo8_pack4)
    // Round 5, the [PAD] prefix of left-padded batches (pad_prefix.hip): per read the 128-token tiles wholly inside its leading run
    // of [PAD], the list of tiles the tail kernels compute, and per arithmetic one table of what an all-[PAD] read leaves behind
    int* pad_p0 = nullptr;                          // [B]
    int* tile_list = nullptr;                       // [1 + B * tiles_x]
    struct PadTable {
        int prec = -1;                              // the arithmetic it was computed in (effective precision, fp16c's level, fp32 path: x3?)
        bool mlp_lo = false, x3 = false;
        int L = 0, Lp = 0;                          // tokens of the all-[PAD] read, row pitch of its z blocks
        void* z[NLAYER] = {};                       // [i], i >= 1: the z block layer i's convolution reads ([D3][Lp] elements incl. lo planes)
        float* scores = nullptr;                    // [L] pooling scores                                   (16-bit fused path)
        float* partial = nullptr;                   // [ceil(L / 128)][POOL_PSTRIDE] pooling partials       (16-bit fused path)
        float* hfin = nullptr;                      // [L][256] the last block's residual rows              (fp32 path; its partials: per 64 tokens)
        // 16-bit fused path, tables of long reads (S = segments of L > 1): per block the segment spectra of the all-[PAD] read's
        // gated signal ([256][S][N], the one read's convolution scratch as it stood) and the running per-thread sums of the last
        // token's dot product -- segments inside the [PAD] prefix of both reads of a pair are not transformed (hyena_conv.hip SegPrefix)
        int S = 1;
        float2* gspec[NLAYER] = {};
        float* dots[NLAYER] = {};
    };
    std::vector<PadTable> pad_tables;
    PadTable* capture = nullptr;                    // inside the forward that fills a table
    unsigned char* pad_ids = nullptr;               // device, all PAD_ID: the table read's ids
    size_t pad_ids_cap = 0;
    float* pad_logits = nullptr;                    // device [2]: that read's logits (unused)
    bool no_pad_skip = false;                       // CLM_DEBUG=no_pad_skip: every tile of every read is computed (A/B runs, tests)
    bool no_seg_skip = false;                       // CLM_DEBUG=no_seg_skip: ... but every segment of the long-read convolution is transformed
    int last_B = 0, last_L = 0, last_Lp = 0;
    // debug / profiling
    int stop_layer = -1, stop_stage = -1;
    bool split_tail = false;      // CLM_DEBUG=split_tail: separate out_proj16 + mlp16 kernels instead of the fused tail (A/B runs)
    bool no_fuse_next = false;    // CLM_DEBUG=no_fuse_next: separate in_proj / score kernels instead of fusing them into the tail
    bool no_idconv = false;       // CLM_DEBUG=no_idconv: run block 0's in_proj instead of the id-table convolution (A/B runs)
    int conv_flags = 0;           // CLM_DEBUG=conv_oneshot / conv_no_xcd: CONV_* switches of the convolution launchers (A/B runs, tests)
    bool no_lone_peel = false;    // CLM_DEBUG=no_lone_peel: keep the lone last token of 128 k + 1-token reads in a tile of its own (A/B runs)
    bool x3 = false;              // CLM_PREC_F16X3: cfg.precision is PREC_F32 inside the engine, the fused tails run on hi + lo halfs (tail32.hip AR_X3)
    bool unfused_fp32 = false;    // CLM_DEBUG=unfused_fp32: exact fp32 through the separate GEMM kernels of rounds 1-3 (tests cross-check the fused tail)
    bool force_generic = false;   // CLM_DEBUG=generic_gemm: route 16-bit modes through the generic kernels (A/B runs)
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    double prof_ms[CLM_N_STAGES] = {};
    int64_t prof_n[CLM_N_STAGES] = {};
};

namespace {

int fail(clm_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(h, CLM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
    } while (0)

size_t elem_size(int prec) { return prec == PREC_F32 ? 4 : 2; }
// The arithmetic a chunk of L-token reads runs in.  fp16c keeps fp16 activation operands; their roundings are independent
// from token to token and average out in the attention pooling like 1/sqrt(L) (measured max |dlogit| at 3x head scale:
// 1.6e-4 at 8193 tokens, 5.6e-4 at 1000, 1.5e-3 at 100), so reads too short to average them take the exact-fp32 kernels --
// they are cheap there -- and the mode stays within the reference's 1e-3 at every length.
int effective_prec(const clm_handle* h, int L);
int round_up(int v, int m) { return (v + m - 1) / m * m; }
// Row pitch (tokens) of the channel-major planes z / y and their lo-byte planes.  (A multiple of the 128-token tile, so that a
// tile's piece of a BYTE row is one whole 128-byte line instead of straddling two on every other row, was measured in round 4:
// tail kernel 21.7 vs 21.5 ms per step on one box, three alternations -- not kept.)
constexpr int LP_ALIGN = 64;

// ---- expected weights -------------------------------------------------------------------------------
struct KeySpec {
    std::string key;
    std::vector<int64_t> shape;
};

std::vector<KeySpec> expected_keys(const clm_config& c) {
    std::vector<KeySpec> k;
    const int64_t d = c.d_model, di = c.d_inner, fo = c.filter_order, hh = c.head_hidden;
    k.push_back({"bb.embeddings.word_embeddings.weight", {c.vocab_rows, d}});
    for (int i = 0; i < c.n_layer; ++i) {
        std::string p = "bb.layers." + std::to_string(i) + ".";
        for (const char* n : {"norm1", "norm2"}) {
            k.push_back({p + n + ".weight", {d}});
            k.push_back({p + n + ".bias", {d}});
        }
        k.push_back({p + "mixer.in_proj.weight", {3 * d, d}});
        k.push_back({p + "mixer.in_proj.bias", {3 * d}});
        k.push_back({p + "mixer.out_proj.weight", {d, d}});
        k.push_back({p + "mixer.out_proj.bias", {d}});
        k.push_back({p + "mixer.short_filter.weight", {3 * d, 1, 3}});
        k.push_back({p + "mixer.short_filter.bias", {3 * d}});
        std::string f = p + "mixer.filter_fn.";
        k.push_back({f + "bias", {d}});
        k.push_back({f + "pos_emb.z", {1, c.max_seq_len, c.emb_dim}});
        k.push_back({f + "pos_emb.t", {1, c.max_seq_len, 1}});
        k.push_back({f + "implicit_filter.0.weight", {fo, c.emb_dim}});
        k.push_back({f + "implicit_filter.0.bias", {fo}});
        k.push_back({f + "implicit_filter.1.freq", {1, fo}});
        k.push_back({f + "implicit_filter.2.weight", {fo, fo}});
        k.push_back({f + "implicit_filter.2.bias", {fo}});
        k.push_back({f + "implicit_filter.4.weight", {fo, fo}});
        k.push_back({f + "implicit_filter.4.bias", {fo}});
        k.push_back({f + "implicit_filter.6.weight", {d, fo}});
        k.push_back({f + "modulation.deltas", {1, 1, d}});
        k.push_back({p + "mlp.fc1.weight", {di, d}});
        k.push_back({p + "mlp.fc1.bias", {di}});
        k.push_back({p + "mlp.fc2.weight", {d, di}});
        k.push_back({p + "mlp.fc2.bias", {d}});
    }
    k.push_back({"bb.ln_f.weight", {d}});
    k.push_back({"bb.ln_f.bias", {d}});
    k.push_back({"head.attention.0.weight", {hh / 2, d}});
    k.push_back({"head.attention.0.bias", {hh / 2}});
    k.push_back({"head.attention.2.weight", {1, hh / 2}});
    k.push_back({"head.attention.2.bias", {1}});
    k.push_back({"head.classifier.0.weight", {hh, d}});
    k.push_back({"head.classifier.0.bias", {hh}});
    k.push_back({"head.classifier.3.weight", {hh, hh}});
    k.push_back({"head.classifier.3.bias", {hh}});
    k.push_back({"head.classifier.6.layers.0.weight", {hh, hh}});
    k.push_back({"head.classifier.6.layers.0.bias", {hh}});
    k.push_back({"head.classifier.6.layers.3.weight", {hh, hh}});
    k.push_back({"head.classifier.6.layers.3.bias", {hh}});
    k.push_back({"head.output_layer.weight", {c.n_classes, hh}});
    k.push_back({"head.output_layer.bias", {c.n_classes}});
    return k;
}

// "net.backbone.backbone.X" | "backbone.backbone.X" | "backbone.X" -> "bb.X";  "net.head.Y" | "head.Y" -> "head.Y"
bool canonical_key(const char* key, std::string& out) {
    std::string s(key);
    if (s.rfind("net.", 0) == 0) s = s.substr(4);
    if (s.rfind("backbone.backbone.", 0) == 0) {
        out = "bb." + s.substr(18);
        return true;
    }
    if (s.rfind("backbone.", 0) == 0) {
        out = "bb." + s.substr(9);
        return true;
    }
    if (s.rfind("head.", 0) == 0) {
        out = s;
        return true;
    }
    return false;
}

__global__ void convert_to_f32_kernel(const void* in, float* out, size_t n, int dtype) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dtype == CLM_DT_F64)
        out[i] = (float)reinterpret_cast<const double*>(in)[i];
    else if (dtype == CLM_DT_BF16)
        out[i] = to_float(reinterpret_cast<const bf16_t*>(in)[i]);
    else if (dtype == CLM_DT_F16)
        out[i] = to_float(reinterpret_cast<const f16_t*>(in)[i]);
}

void free_filter_set(FilterSet& f) {
    for (int i = 0; i < NLAYER; ++i) {
        if (f.ktime[i]) (void)hipFree(f.ktime[i]);
        if (f.kf[i]) (void)hipFree(f.kf[i]);
        if (f.kfp[i]) (void)hipFree(f.kfp[i]);
        f.kfp[i] = nullptr;
        for (auto& r : f.krev)
            if (r.p[i]) (void)hipFree(r.p[i]);
        f.ktime[i] = nullptr;
        f.kf[i] = nullptr;
    }
    f.krev.clear();
    if (f.tw) (void)hipFree(f.tw);
    f.tw = nullptr;
}

void free_filters(clm_handle* h) {
    for (auto& f : h->filters) free_filter_set(f);
    h->filters.clear();
}

// The per-chunk workspace: twelve buffers, each with its own capacity in bytes and grown on its own -- a call needs Bc x (its
// own length) of each, and chunk_for() bounds that product whatever the read length, so a handle that has seen 256 x 8k-token and
// 32 x 32k-token batches holds the larger of the two needs per buffer, not 256 x 32k (the round-2 shape bookkeeping did).
enum { WS_H, WS_Z, WS_Y, WS_U, WS_SCORES, WS_STATS, WS_PARTIAL, WS_POOLED, WS_GSCRATCH, WS_IDS8, WS_LONE, WS_EDGE_READ, WS_YLO, WS_P0, WS_TILES, WS_N };
static_assert(WS_N == sizeof(clm_handle::ws_cap) / sizeof(size_t), "one capacity per buffer");
void** ws_slot(clm_handle* h, int i) {
    switch (i) {
        case WS_H: return (void**)&h->h;
        case WS_Z: return &h->z;
        case WS_Y: return &h->y;
        case WS_U: return &h->u;
        case WS_SCORES: return (void**)&h->scores;
        case WS_STATS: return (void**)&h->stats;
        case WS_PARTIAL: return (void**)&h->partial;
        case WS_POOLED: return (void**)&h->pooled;
        case WS_GSCRATCH: return (void**)&h->gscratch;
        case WS_IDS8: return (void**)&h->ids8;
        case WS_LONE: return (void**)&h->lone_ws;
        case WS_YLO: return (void**)&h->ylo;
        case WS_P0: return (void**)&h->pad_p0;
        case WS_TILES: return (void**)&h->tile_list;
        default: return (void**)&h->edge_read;
    }
}

void free_workspace(clm_handle* h) {
    for (int i = 0; i < WS_N; ++i) {
        void** p = ws_slot(h, i);
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        h->ws_cap[i] = 0;
    }
    h->ws_es = 0;
}

void free_packed(clm_handle* h) {
    for (int i = 0; i < NLAYER; ++i)
        for (int j = 0; j < 4; ++j)
        {
            if (h->packed[i][j]) { (void)hipFree(h->packed[i][j]); h->packed[i][j] = nullptr; }
            if (h->packed32[i][j]) { (void)hipFree(h->packed32[i][j]); h->packed32[i][j] = nullptr; }
            if (h->packed32t[i][j]) { (void)hipFree(h->packed32t[i][j]); h->packed32t[i][j] = nullptr; }
            if (h->packed32x[i][j]) { (void)hipFree(h->packed32x[i][j]); h->packed32x[i][j] = nullptr; }
            if (j < 2 && h->packed_mlpc[i][j]) { (void)hipFree(h->packed_mlpc[i][j]); h->packed_mlpc[i][j] = nullptr; }
        }
    if (h->packed_score) { (void)hipFree(h->packed_score); h->packed_score = nullptr; }
    if (h->packed_score32) { (void)hipFree(h->packed_score32); h->packed_score32 = nullptr; }
    if (h->packed_score32t) { (void)hipFree(h->packed_score32t); h->packed_score32t = nullptr; }
    if (h->packed_score32x) { (void)hipFree(h->packed_score32x); h->packed_score32x = nullptr; }
    if (h->ztab) { (void)hipFree(h->ztab); h->ztab = nullptr; }
    for (int i = 0; i < NLAYER; ++i)
        if (h->fir[i]) { (void)hipFree(h->fir[i]); h->fir[i] = nullptr; }
    for (int j = 0; j < 5; ++j)
        if (h->head_t[j]) { (void)hipFree(h->head_t[j]); h->head_t[j] = nullptr; }
}

void free_pad_table_spectra(clm_handle::PadTable& t) {
    for (int i = 0; i < NLAYER; ++i) {
        if (t.gspec[i]) (void)hipFree(t.gspec[i]);
        if (t.dots[i]) (void)hipFree(t.dots[i]);
        t.gspec[i] = nullptr; t.dots[i] = nullptr;
    }
}

void free_pad_tables(clm_handle* h) {
    for (auto& t : h->pad_tables) {
        free_pad_table_spectra(t);
        for (int i = 0; i < NLAYER; ++i)
            if (t.z[i]) (void)hipFree(t.z[i]);
        if (t.scores) (void)hipFree(t.scores);
        if (t.partial) (void)hipFree(t.partial);
        if (t.hfin) (void)hipFree(t.hfin);
    }
    h->pad_tables.clear();
}

const float* W(clm_handle* h, const std::string& key) { return h->w[key].d; }

// exact fp32 runs its block tails fused (tail32.hip) unless a debug stop wants an intermediate or CLM_DEBUG=unfused_fp32 asks
bool fused_fp32(const clm_handle* h) { return !h->unfused_fp32 && h->stop_stage < 0; }

int ensure_workspace(clm_handle* h, int Bc, int L, hipStream_t st) {
    const int prec = effective_prec(h, L);                   // (honours the self-check's referee pass and the fallback)
    const size_t es = elem_size(prec), Lp = (size_t)round_up(L, LP_ALIGN), nb = (size_t)Bc, nl = (size_t)L;
    size_t need[WS_N] = {};
    need[WS_H] = nb * nl * D * 4;
    need[WS_Z] = nb * D3 * Lp * es;
    need[WS_Y] = nb * D * Lp * es;
    need[WS_U] = ((prec == PREC_F32 && !fused_fp32(h)) || (prec != PREC_F32 && h->force_generic)) ? nb * DI * nl * es : 0;   // the 1024-wide fc1 output: unfused paths only
    need[WS_SCORES] = nb * nl * 4;
    need[WS_STATS] = nb * 2 * 4;
    // pooling partials: [POOL_SPLIT][4][256] per read (unfused fp32 path) or one POOL_PSTRIDE row per tile -- 128 tokens in the
    // 16-bit tail kernel, T32_TILE = 64 in the exact / fp16x3 one
    need[WS_PARTIAL] = nb * std::max((size_t)POOL_SPLIT * 4 * D, (size_t)((nl + T32_TILE - 1) / T32_TILE) * POOL_PSTRIDE) * 4;
    need[WS_POOLED] = nb * D * 4;
    need[WS_LONE] = lone_token_ws_floats((int)nb) * 4;
    need[WS_IDS8] = nb * Lp;
    need[WS_EDGE_READ] = nb * D3 * sizeof(float2);
    need[WS_YLO] = h->cfg.precision == PREC_F16C ? nb * D * Lp : 0;
    need[WS_P0] = 3 * nb * sizeof(int);                     // p0 | pair order | pair partner (pad_prefix.hip)
    need[WS_TILES] = (1 + nb * ((nl + 127) / 128)) * sizeof(int);
    if (conv_segments_for(L) > 1) need[WS_GSCRATCH] = ((nb + 1) / 2) * D * (size_t)conv_segments_for(L) * 16384 * sizeof(float2);
    bool grow = false;
    for (int i = 0; i < WS_N; ++i) grow |= need[i] > h->ws_cap[i];
    if (!h->edge_bnd) HIPCHK(h, hipMalloc((void**)&h->edge_bnd, (size_t)1024 * 2 * D3 * sizeof(float2)));   // >= any grid (one workgroup per CU)
    if (!grow) return CLM_OK;
    HIPCHK(h, hipStreamSynchronize(st));
    for (int i = 0; i < WS_N; ++i) {
        if (need[i] <= h->ws_cap[i]) continue;
        void** p = ws_slot(h, i);
        if (*p) HIPCHK(h, hipFree(*p));
        *p = nullptr;
        h->ws_cap[i] = 0;
        HIPCHK(h, hipMalloc(p, need[i]));
        h->ws_cap[i] = need[i];
        if (i == WS_Z || i == WS_Y || i == WS_YLO) HIPCHK(h, hipMemset(*p, 0, need[i]));   // padding columns [L, Lp) must never hold NaN garbage
    }
    return CLM_OK;
}

int ensure_filters(clm_handle* h, int L, hipStream_t st, FilterSet** out, const ReversedFilter** krev_out) {
    *krev_out = nullptr;
    const int S = conv_segments_for(L);
    const int logn = S > 1 ? 14 : conv_logn_for(L);
    if (logn < 0 || L > h->cfg.max_seq_len)
        return fail(h, CLM_E_UNSUPPORTED, "sequence length " + std::to_string(L) + " tokens exceeds max_seq_len " +
                                              std::to_string(h->cfg.max_seq_len));
    const int key = S > 1 ? KEY_LONG : logn;
    FilterSet* fs = nullptr;
    for (auto& f : h->filters)
        if (f.key == key) fs = &f;
    const int N = 1 << logn;
    if (!fs) {
        HIPCHK(h, hipStreamSynchronize(st));
        FilterSet f;
        f.key = key;
        f.logn = logn;
        f.Lf = S > 1 ? h->cfg.max_seq_len : std::min(N / 2 + 1, h->cfg.max_seq_len);
        f.KS = S > 1 ? conv_segments_for(h->cfg.max_seq_len) : 1;
        double2* scratch = nullptr;
        HIPCHK(h, hipMalloc((void**)&scratch, (size_t)D * N * sizeof(double2)));
        HIPCHK(h, hipMalloc((void**)&f.tw, (size_t)(N / 2) * sizeof(float2)));
        launch_twiddles(f.tw, logn, st);
        for (int i = 0; i < NLAYER; ++i) {
            HIPCHK(h, hipMalloc((void**)&f.ktime[i], (size_t)f.Lf * D * 4));
            HIPCHK(h, hipMalloc((void**)&f.kf[i], (size_t)D * f.KS * N * sizeof(float2)));
            std::string p = "bb.layers." + std::to_string(i) + ".mixer.filter_fn.";
            launch_filter(W(h, p + "pos_emb.z"), W(h, p + "pos_emb.t"), W(h, p + "implicit_filter.0.weight"),
                          W(h, p + "implicit_filter.0.bias"), W(h, p + "implicit_filter.1.freq"),
                          W(h, p + "implicit_filter.2.weight"), W(h, p + "implicit_filter.2.bias"),
                          W(h, p + "implicit_filter.4.weight"), W(h, p + "implicit_filter.4.bias"),
                          W(h, p + "implicit_filter.6.weight"), W(h, p + "modulation.deltas"), f.ktime[i], f.Lf, st);
            if (S == 1) {
                launch_filter_spectrum(f.ktime[i], W(h, p + "bias"), f.kf[i], scratch, f.Lf, logn, 0, f.Lf, -1, st);
                if (logn == 14) {                            // (round 5: the exact / fp16x3 engine's fp32 rows take the persistent kernel too)
                    HIPCHK(h, hipMalloc((void**)&f.kfp[i], (size_t)D * N * sizeof(float2)));
                    launch_spectrum_lanepack(f.kf[i], f.kfp[i], 1, 0, st);
                }
            } else {   // kf [256][KS][N], lane-packed for the segmented kernel: one launch per partition, through a temporary
                float2* tmp = nullptr;
                HIPCHK(h, hipMalloc((void**)&tmp, (size_t)D * N * sizeof(float2)));
                for (int j = 0; j < f.KS; ++j) {
                    launch_filter_spectrum(f.ktime[i], W(h, p + "bias"), tmp, scratch, f.Lf, logn, j * SEG_LEN, SEG_LEN,
                                           (j - 1) * SEG_LEN, st);
                    launch_spectrum_lanepack(tmp, f.kf[i], f.KS, j, st);
                }
                HIPCHK(h, hipStreamSynchronize(st));
                HIPCHK(h, hipFree(tmp));
            }
        }
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, hipFree(scratch));
        h->filters.push_back(f);
        fs = &h->filters.back();
    }
    if (conv_lone_tail(L)) {               // the reversed taps [0, L) of the dot product for the lone last token
        for (auto& r : fs->krev)
            if (r.L == L) *krev_out = &r;
        if (!*krev_out) {
            ReversedFilter r;
            r.L = L;
            r.stride = round_up(L, 8);
            for (int i = 0; i < NLAYER; ++i) {
                std::string p = "bb.layers." + std::to_string(i) + ".mixer.filter_fn.";
                HIPCHK(h, hipMalloc((void**)&r.p[i], (size_t)D * r.stride * 4));
                launch_filter_reversed(fs->ktime[i], W(h, p + "bias"), r.p[i], L, r.stride, st);
            }
            fs->krev.push_back(r);
            *krev_out = &fs->krev.back();
        }
    }
    *out = fs;
    return CLM_OK;
}

struct StageTimer {
    clm_handle* h;
    hipStream_t st;
    int stage;
    bool on;
    hipEvent_t e0{}, e1{};
    StageTimer(clm_handle* h_, hipStream_t st_, int stage_) : h(h_), st(st_), stage(stage_), on(h_->prof) {
        if (!on) return;
        if (h->recs.size() > 200000) { on = false; return; }
        if (!h->free_events.empty()) {
            e0 = h->free_events.back().first;
            e1 = h->free_events.back().second;
            h->free_events.pop_back();
        } else if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(e0, st);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(e1, st);
        h->recs.push_back({stage, e0, e1});
    }
};

int effective_prec(const clm_handle* h, int L) {
    if (h->force_prec >= 0) return h->force_prec;
    if (h->fallback > 0) return (int)PREC_F32;
    return (h->cfg.precision == PREC_F16C && L < h->f16c_min_len) ? (int)PREC_F32 : h->cfg.precision;
}
// Does the fp32 path of this handle multiply hi + lo halfs (three fp16 MFMAs per product, tail32.hip AR_X3) right now?  An fp16x3
// handle: unless told to fall back; a 16-bit handle (short reads of fp16c, fall-back level 1): unless told to fall back all the
// way (level 2).  Never inside the referee pass of a self-check, never on an fp32 handle.
bool fp32_path_is_x3(const clm_handle* h) {
    if (h->referee) return false;
    if (h->x3) return h->fallback == 0;
    return h->cfg.precision != PREC_F32 && h->fallback < 2;
}

// Asynchronous error of an EARLIER forward (the call itself returned before its kernels ran): reported once, by the next
// forward / stage_wait / profile_read / clm_check call.  The reference raises IndexError inside nn.Embedding for such ids.
int check_bad_ids(clm_handle* h) {
    if (h->bad_ids && *reinterpret_cast<volatile int*>(h->bad_ids)) {
        *reinterpret_cast<volatile int*>(h->bad_ids) = 0;
        return fail(h, CLM_E_INVALID, "an earlier batch held token ids outside [0, " + std::to_string(h->cfg.vocab_rows) +
                                          ") (its logits were computed with those ids clamped; the reference raises IndexError)");
    }
    return CLM_OK;
}

bool stop_here(clm_handle* h, int layer, int stage) { return h->stop_layer == layer && h->stop_stage == stage; }

// Reads pushed through all layers together: cfg.chunk_reads, capped by tokens so that a chunk's workspace stays bounded whatever
// the read length -- 256 x 8,256 tokens in the 16-bit modes (z + y + h: 6.5 GB; measured round 3, same box: 7,976 / 8,047 / 8,164
// reads/s at 64 / 128 / 256 reads per chunk: the ~26 small launches of a chunk, the persistent kernels' ramps and the head are paid
// per chunk, and 288 GB of HBM have room), a quarter of that in exact fp32 (its 1024-wide fc1 output is 4 KiB per token, and the
// proven size).  Even, so that a chunk boundary never splits a read pair of the packed transform.
int chunk_for(const clm_handle* h, int L) {
    // (exact fp32 with the fused tail has no fc1 output in HBM either: the same cap as the 16-bit modes)
    const long long cap_tokens = (effective_prec(h, L) == PREC_F32 && !fused_fp32(h)) ? 64LL * 8256 : 256LL * 8256;
    long long c = cap_tokens / round_up(L, LP_ALIGN);
    if (c > h->cfg.chunk_reads) c = h->cfg.chunk_reads;
    if (c > TILE_LIST_MAX_READS) c = TILE_LIST_MAX_READS;     // (a tile-list entry holds its read in 12 bits)
    if (c > 1) c &= ~1LL;
    return c < 1 ? 1 : (int)c;
}

int forward_chunk(clm_handle* h, const void* ids, int ids_dtype, int64_t row_stride, int Bc, int L, float* logits,
                  hipStream_t st);

// The all-[PAD] table of the arithmetic reads of this chunk run in (pad_prefix.hip), long enough for L tokens: built on first
// use -- ONE forward of one all-[PAD] read through this very engine, with forward_chunk's capture hooks copying out what the
// later stages read of it -- and kept until the weights change.  Lengths come in classes (1,025 ... 32,769 tokens, then
// max_seq_len) so that a file of ragged batches builds at most a handful.  The values at position t do not depend on the length of
// the read they were computed in (causal backbone, taps independent of L) beyond the rounding of its transform size.
int ensure_pad_table(clm_handle* h, int prec, bool x3, int L, hipStream_t st, clm_handle::PadTable** out) {
    for (auto& t : h->pad_tables)
        if (t.prec == prec && t.x3 == x3 && t.mlp_lo == (prec == PREC_F16C && h->mlp_lo) && t.L >= L) { *out = &t; return CLM_OK; }
    int LT = 1025;
    while (LT < L && LT < 32769) LT = 2 * (LT - 1) + 1;
    if (LT < L) LT = h->cfg.max_seq_len;
    if (LT > h->cfg.max_seq_len) LT = h->cfg.max_seq_len;
    if (LT < L) return fail(h, CLM_E_INVALID, "ensure_pad_table: read longer than max_seq_len");
    HIPCHK(h, hipStreamSynchronize(st));
    for (size_t k = 0; k < h->pad_tables.size(); ++k) {      // a shorter table of the same arithmetic is replaced
        auto& t = h->pad_tables[k];
        if (t.prec == prec && t.x3 == x3 && t.mlp_lo == (prec == PREC_F16C && h->mlp_lo)) {
            free_pad_table_spectra(t);
            for (int i = 0; i < NLAYER; ++i)
                if (t.z[i]) (void)hipFree(t.z[i]);
            if (t.scores) (void)hipFree(t.scores);
            if (t.partial) (void)hipFree(t.partial);
            if (t.hfin) (void)hipFree(t.hfin);
            h->pad_tables.erase(h->pad_tables.begin() + (long)k);
            break;
        }
    }
    clm_handle::PadTable t;
    t.prec = prec; t.x3 = x3; t.mlp_lo = prec == PREC_F16C && h->mlp_lo;
    t.L = LT; t.Lp = round_up(LT, LP_ALIGN);
    const size_t es = elem_size(prec);
    for (int i = 1; i < NLAYER; ++i) HIPCHK(h, hipMalloc(&t.z[i], (size_t)D3 * t.Lp * es));
    // (the exact path: partials per 64-token tile, and the final residual rows as well -- clm_debug_fetch("hidden") shows them)
    if (prec == PREC_F32) HIPCHK(h, hipMalloc((void**)&t.hfin, (size_t)LT * D * 4));
    HIPCHK(h, hipMalloc((void**)&t.scores, (size_t)LT * 4));
    HIPCHK(h, hipMalloc((void**)&t.partial, (size_t)((LT + T32_TILE - 1) / T32_TILE) * POOL_PSTRIDE * 4));
    t.S = conv_segments_for(LT);
    if (t.S > 1 && prec != PREC_F32 && !h->no_seg_skip)           // (the 16-bit fused path's segmented convolution skips prefix segments)
        for (int i = 0; i < NLAYER; ++i) {
            HIPCHK(h, hipMalloc((void**)&t.gspec[i], (size_t)D * t.S * 16384 * sizeof(float2)));
            HIPCHK(h, hipMalloc((void**)&t.dots[i], (size_t)D * (t.S - 1) * SEG_DOT_THREADS * 4));
            HIPCHK(h, hipMemsetAsync(t.dots[i], 0, (size_t)D * (t.S - 1) * SEG_DOT_THREADS * 4, st));
        }
    if ((size_t)t.Lp > h->pad_ids_cap) {
        if (h->pad_ids) HIPCHK(h, hipFree(h->pad_ids));
        h->pad_ids = nullptr; h->pad_ids_cap = 0;
        HIPCHK(h, hipMalloc((void**)&h->pad_ids, (size_t)t.Lp));
        h->pad_ids_cap = (size_t)t.Lp;
        HIPCHK(h, hipMemsetAsync(h->pad_ids, PAD_ID, (size_t)t.Lp, st));
    }
    if (!h->pad_logits) HIPCHK(h, hipMalloc((void**)&h->pad_logits, NCLS * 4));
    h->pad_tables.push_back(t);
    clm_handle::PadTable* tp = &h->pad_tables.back();
    const int force = h->force_prec;
    const bool prof = h->prof;
    h->force_prec = prec;                                    // (fp16c: the class length may lie on the other side of the length switch)
    h->prof = false;
    h->capture = tp;
    const int rc = forward_chunk(h, h->pad_ids, CLM_DT_U8, t.Lp, 1, LT, h->pad_logits, st);
    h->capture = nullptr;
    h->prof = prof;
    h->force_prec = force;
    if (rc) {
        free_pad_table_spectra(*tp);
        for (int i = 0; i < NLAYER; ++i)
            if (tp->z[i]) (void)hipFree(tp->z[i]);
        if (tp->scores) (void)hipFree(tp->scores);
        if (tp->partial) (void)hipFree(tp->partial);
        if (tp->hfin) (void)hipFree(tp->hfin);
        h->pad_tables.pop_back();
        return rc;
    }
    *out = tp;
    return CLM_OK;
}

int forward_chunk(clm_handle* h, const void* ids, int ids_dtype, int64_t row_stride, int Bc, int L, float* logits,
                  hipStream_t st) {
    const int prec = effective_prec(h, L), Lp = round_up(L, LP_ALIGN);
    const bool alt32 = prec != h->cfg.precision;              // fp16c engine, short reads: exact-fp32 kernels and packing
    const float eps = h->cfg.ln_eps;
    FilterSet* fs = nullptr;
    const ReversedFilter* kr = nullptr;
    int rc = ensure_filters(h, L, st, &fs, &kr);
    const int S = conv_segments_for(L);
    if (rc) return rc;
    rc = ensure_workspace(h, Bc, L, st);
    if (rc) return rc;
    h->last_B = Bc; h->last_L = L; h->last_Lp = Lp;
    if (h->ws_es != elem_size(prec)) {   // fp16c: fp32 and fp16 chunks share z / y -- what one type left in the padding
        if (h->ws_es) {                  // columns may read as NaN in the other
            HIPCHK(h, hipMemsetAsync(h->z, 0, h->ws_cap[WS_Z], st));
            HIPCHK(h, hipMemsetAsync(h->y, 0, h->ws_cap[WS_Y], st));
        }
        h->ws_es = elem_size(prec);
    }
    const bool tuned16 = prec != PREC_F32 && (!h->force_generic || prec == PREC_F16C);   // fp16c has no generic kernels
    // 16-bit modes, no debug stop: block 0 never touches the fp32 embedding rows in HBM --
    // its in_proj is a 16-row table looked up by the convolution and its residual is gathered from the embedding table
    const bool idpath = tuned16 && !h->no_idconv && !h->split_tail && h->stop_stage < 0;
    // ... and every block's tail kernel goes on, on the tile it has just produced, with LayerNorm-1 + in_proj of the next
    // block (the last block: ln_f + attention scores + pooling partials): no separate in_proj / score launches
    const bool fuse_next = tuned16 && !h->split_tail && !h->no_fuse_next && h->stop_stage < 0;
    // reads of 128 k + 1 tokens (every 8k-bp read: 8192 bases + [SEP]): the last token would be a tile of its own, a whole extra
    // round of the tail kernel for one token per read; it is causally isolated, so a per-read matrix-vector kernel takes it
    const bool peel = fuse_next && !h->no_lone_peel && L > 128 && L % 128 == 1;
    // ... and hands z over in the form the convolution reads: x0f and g = x1f * vf, filtered and gated by the in_proj stage itself
    // (two rows per channel instead of three; gemm16.hip inproj_blocks_gated)
    const bool zgated = fuse_next && !h->raw_z;
    // exact fp32: one fused kernel per block tail, the next block's in_proj included (tail32.hip)
    const bool fused32 = prec == PREC_F32 && fused_fp32(h);
    const bool x3 = fused32 && fp32_path_is_x3(h);              // fp16x3: hi + lo halfs in the fused tails (the referee pass: exact)
    // Round 5: tiles wholly inside a read's [PAD] prefix are not computed, their rows come from the all-[PAD] table (pad_prefix.hip).
    // In the fused paths only (the debug / unfused paths keep computing everything), never inside the forward that fills a table,
    // and only for reads long enough to hold a whole prefix tile next to a real token.
    const int Lmain = (tuned16 && peel) ? L - 1 : L;
    const bool tail16_path = tuned16 && !h->split_tail;      // the persistent tail kernel runs: it walks the tile list
    const bool pad_skip = !h->no_pad_skip && !h->capture && L >= 256 && ((tail16_path && fuse_next) || fused32);
    clm_handle::PadTable* ptab = nullptr;
    if (pad_skip) {                                          // (before this chunk's ids land in the workspace: the build runs through it)
        rc = ensure_pad_table(h, prec, x3, L, st, &ptab);
        if (rc) return rc;
        rc = ensure_workspace(h, Bc, L, st);                 // (the build may have regrown -- never shrunk -- the buffers; cheap when not)
        if (rc) return rc;
        h->last_B = Bc; h->last_L = L; h->last_Lp = Lp;
        if (h->ws_es != elem_size(prec)) {                   // (as above: the build ran in this chunk's own arithmetic, so this is a no-op)
            HIPCHK(h, hipMemsetAsync(h->z, 0, h->ws_cap[WS_Z], st));
            HIPCHK(h, hipMemsetAsync(h->y, 0, h->ws_cap[WS_Y], st));
            h->ws_es = elem_size(prec);
        }
        rc = ensure_filters(h, L, st, &fs, &kr);             // (the build may have added a filter class: the vector behind fs moved)
        if (rc) return rc;
    }
    // fp16c, round 4: y (every block) and the gated rows of z carry one lo byte per element next to the halfs
    unsigned char* const ylo = (prec == PREC_F16C && tuned16) ? h->ylo : nullptr;
    if (zgated && tail16_grid(((peel ? L - 1 : L) + 127) / 128 * Bc) > 1024)
        return fail(h, CLM_E_UNSUPPORTED, "more than 1024 compute units: edge_bnd is sized for 1024 workgroups");
    const void* packed_score = alt32 ? h->packed_score32 : h->packed_score;
    const ScorePoolArgs spa{h->h, W(h, "bb.ln_f.weight"), W(h, "bb.ln_f.bias"), packed_score,
                            W(h, "head.attention.0.bias"), W(h, "head.attention.2.weight"),
                            W(h, "head.attention.2.bias"), h->scores, h->partial, Bc, L, (L + 127) / 128, eps};
    {
        StageTimer t(h, st, CLM_STAGE_EMBED);
        // (exact / fp16x3 engine, one-shot convolution: block 0 reads z from the id table and its tail gathers the residual rows from
        //  the embedding table -- h is first written by that tail kernel, as in the 16-bit id path)
        const bool id32 = fused32 && S == 1 && !h->no_idconv;
        launch_embed(ids, ids_dtype, row_stride, W(h, "bb.embeddings.word_embeddings.weight"), (idpath || id32) ? nullptr : h->h,
                     h->ids8, Bc, L, Lp, st, h->bad_ids);
    }
    if (stop_here(h, -1, CLM_STAGE_EMBED)) return CLM_OK;
    if (tail16_path || pad_skip) launch_pad_tiles(h->ids8, Bc, Lp, Lmain, pad_skip ? 1 : 0, h->pad_p0, h->tile_list, st);
    // does block j's (segmented) convolution leave out the segments inside the [PAD] prefix of both reads of a pair (SegPrefix)?  Block 0
    // looks z up by token id, the others read the gated hand-over; reads of S * 8192 + 1 tokens need the table's dot-product sums, which
    // belong to ONE length
    // (with it the pairs of those convolutions are formed by descending prefix: perm / partner live behind p0)
    int* const pair_perm = h->pad_p0 + Bc;
    int* const pair_partner = h->pad_p0 + 2 * Bc;
    if (pad_skip && S > 1) launch_pair_order(h->pad_p0, Bc, pair_perm, pair_partner, st);
    auto seg_skip_layer = [&](int j) {
        return pad_skip && S > 1 && tail16_path && fuse_next && !h->capture && (j == 0 ? idpath : zgated) && ptab->gspec[j] != nullptr &&
               (!kr || L == ptab->L);
    };
    for (int i = 0; i < NLAYER; ++i) {
        const LayerW& lw = x3 ? h->lwx[i] : (alt32 ? h->lw32[i] : h->lw[i]);
        // block 0 in the 16-bit modes: its in_proj output is a function of the token id alone, the convolution looks it
        // up (ztab), single-shot and segmented kernel alike -- unless a debug stop asks for z itself or CLM_DEBUG=no_idconv
        // (exact fp32 with the fused tail, single-shot convolution: the same table -- it is fp32 -- so block 0 needs no in_proj launch)
        const bool idconv = i == 0 && (idpath || (tuned16 && !h->no_idconv && !stop_here(h, 0, CLM_STAGE_INPROJ)) ||
                                       (fused32 && S == 1 && !h->no_idconv));
        if (!idconv && !(fuse_next && i > 0) && !(fused32 && i > 0)) {
            StageTimer t(h, st, CLM_STAGE_INPROJ);
            if (tuned16) launch_inproj16(prec, h->h, lw.ln1_g, lw.ln1_b, lw.w_in, lw.b_in, h->z, Bc, L, Lp, eps, st);
            else launch_inproj(prec, h->h, lw.ln1_g, lw.ln1_b, lw.w_in, lw.b_in, h->z, Bc, L, Lp, eps, st);
        }
        if (stop_here(h, i, CLM_STAGE_INPROJ)) return CLM_OK;
        {
            StageTimer t(h, st, CLM_STAGE_CONV);
            if (S == 1)
                launch_hyena_conv(prec, h->z, h->y, fs->kf[i], fs->tw, fs->ktime[i], lw.short_w, lw.short_b, Bc, L, Lp,
                                  fs->logn, idconv ? h->ids8 : nullptr, idconv ? h->ztab : nullptr, st,
                                  h->conv_flags | ((zgated && i > 0) ? CONV_GATED : 0), fs->kfp[i], ylo);
            else {
                // [PAD]-prefix reuse: segments inside the prefix of both reads of a pair come from the table (SegPrefix).  Reads of
                // S * 8192 + 1 tokens carry the last token's dot product through the segments: its table sums belong to ONE length
                SegPrefix pfx;
                if (h->capture && h->capture->gspec[i]) {
                    pfx.dots_out = kr ? h->capture->dots[i] : nullptr;
                    pfx.dots_segs = h->capture->S - 1;
                } else if (seg_skip_layer(i)) {
                    pfx.p0 = h->pad_p0;
                    pfx.perm = pair_perm;
                    pfx.tab = ptab->gspec[i];
                    pfx.tab_segs = ptab->S;
                    pfx.dots_in = ptab->dots[i];
                    pfx.dots_segs = ptab->S - 1;
                }
                launch_hyena_conv_seg(prec, h->z, h->y, fs->kf[i], fs->KS, fs->tw, lw.short_w, lw.short_b, h->gscratch, Bc,
                                      L, Lp, S, kr ? kr->p[i] : nullptr, kr ? kr->stride : 0, idconv ? h->ids8 : nullptr,
                                      idconv ? h->ztab : nullptr, st, h->conv_flags | ((zgated && i > 0) ? CONV_GATED : 0), ylo, pfx);
                if (h->capture && h->capture->gspec[i]) {   // (one read = pair 0: [256][S][N] at the head of the scratch), then pair form
                    HIPCHK(h, hipMemcpyAsync(h->capture->gspec[i], h->gscratch, (size_t)D * S * 16384 * sizeof(float2), hipMemcpyDeviceToDevice, st));
                    launch_spectra_pair_form(h->capture->gspec[i], S, st);
                }
            }
        }
        if (stop_here(h, i, CLM_STAGE_CONV)) return CLM_OK;
        const bool stop_mid = stop_here(h, i, CLM_STAGE_OUTPROJ);
        if (tuned16 && !h->split_tail && !stop_mid) {   // out_proj + LN2 + fc1 + GELU + fc2 + both residuals: one kernel
            StageTimer t(h, st, CLM_STAGE_TAIL);
            const bool mlpc = prec == PREC_F16C && h->mlp_lo;
            TailArgs ta{h->y, h->h, lw.w_out, mlpc ? h->packed_mlpc[i][0] : lw.w_fc1, mlpc ? h->packed_mlpc[i][1] : lw.w_fc2, lw.b_out,
                        lw.ln2_g, lw.ln2_b, lw.b_fc1, lw.b_fc2, Bc, L, Lp,
                        eps, peel ? L - 1 : L, (idpath && i == 0) ? h->ids8 : nullptr, W(h, "bb.embeddings.word_embeddings.weight"),
                        nullptr, nullptr, nullptr, nullptr, nullptr, spa};
            ta.ylo = ylo;
            ta.mlp_lo = mlpc;
            ta.tiles = h->tile_list;
            int next = NEXT_NONE;
            if (fuse_next && i + 1 < NLAYER) {
                const LayerW& nx = h->lw[i + 1];
                ta.n_w = nx.w_in; ta.n_bias = nx.b_in; ta.n_g = nx.ln1_g; ta.n_b = nx.ln1_b; ta.n_z = h->z;
                next = NEXT_INPROJ;
                if (zgated) {
                    ta.zg = 1; ta.n_fir = h->fir[i + 1]; ta.edge_bnd = h->edge_bnd;
                    ta.edge_read = peel ? h->edge_read : nullptr;
                    ta.zlo = ylo != nullptr;
                }
            } else if (fuse_next) {
                next = NEXT_SCORE;
            }
            launch_tail16(prec, ta, next, st);
            if (ta.zg) launch_gated_patch(prec, ta, st);        // tokens 0, 1 of the workgroup ranges that start inside a read
            if (peel) {
                const std::string p = "bb.layers." + std::to_string(i) + ".", pn = "bb.layers." + std::to_string(i + 1) + ".";
                const bool last = i + 1 == NLAYER;
                LoneTokenArgs la{};
                la.y = h->y; la.h = h->h; la.ids8 = ta.ids8; la.emb = ta.emb;
                la.w_out = W(h, p + "mixer.out_proj.weight"); la.b_out = lw.b_out; la.ln2_g = lw.ln2_g; la.ln2_b = lw.ln2_b;
                la.w_fc1 = W(h, p + "mlp.fc1.weight"); la.b_fc1 = lw.b_fc1; la.w_fc2 = W(h, p + "mlp.fc2.weight"); la.b_fc2 = lw.b_fc2;
                la.last = last;
                if (last) {
                    la.n_g = W(h, "bb.ln_f.weight"); la.n_b = W(h, "bb.ln_f.bias");
                    la.att_w1 = W(h, "head.attention.0.weight"); la.att_b1 = W(h, "head.attention.0.bias");
                    la.att_w2 = W(h, "head.attention.2.weight"); la.att_b2 = W(h, "head.attention.2.bias");
                    la.scores = h->scores; la.partial = h->partial;
                } else {
                    const LayerW& nx = h->lw[i + 1];
                    la.n_g = nx.ln1_g; la.n_b = nx.ln1_b; la.n_w = W(h, pn + "mixer.in_proj.weight"); la.n_bias = nx.b_in; la.n_z = h->z;
                    if (ta.zg) { la.n_fir = ta.n_fir; la.edge_read = h->edge_read; }
                }
                la.ws = h->lone_ws;
                la.B = Bc; la.L = L; la.Lp = Lp; la.ntiles = (L + 127) / 128; la.eps = eps;
                la.ylo = ylo; la.zlo = ta.zlo;
                launch_lone_token(prec, la, st);
            }
            // the [PAD] prefix: what this block leaves for the next stage, copied out of (capture) or in from (pad_skip) the table
            const int nrow16 = ta.zg ? 2 * D : D3, nlo = ta.zlo ? 2 * D : 0;
            if (h->capture && fuse_next) {
                if (next == NEXT_INPROJ) HIPCHK(h, hipMemcpyAsync(h->capture->z[i + 1], h->z, (size_t)D3 * Lp * elem_size(prec), hipMemcpyDeviceToDevice, st));
                else {
                    HIPCHK(h, hipMemcpyAsync(h->capture->scores, h->scores, (size_t)L * 4, hipMemcpyDeviceToDevice, st));
                    HIPCHK(h, hipMemcpyAsync(h->capture->partial, h->partial, (size_t)((L + 127) / 128) * POOL_PSTRIDE * 4, hipMemcpyDeviceToDevice, st));
                }
            } else if (pad_skip) {
                if (next == NEXT_INPROJ)      // (rows of segments the next convolution will not read are not copied)
                    launch_prefix_fill_z(h->pad_p0, h->z, ptab->z[i + 1], Bc, Lp, ptab->Lp, Lmain, (int)elem_size(prec), nrow16, nlo, st,
                                         seg_skip_layer(i + 1) ? S : 0, pair_partner);
                else
                    launch_prefix_fill_pool(h->pad_p0, h->scores, h->partial, ptab->scores, ptab->partial, Bc, L, (L + 127) / 128, Lmain, st);
            }
        } else if (fused32) {
            StageTimer t(h, st, CLM_STAGE_TAIL);
            const LayerW* nx = i + 1 < NLAYER ? &(x3 ? h->lwx[i + 1] : (alt32 ? h->lw32[i + 1] : h->lw[i + 1])) : nullptr;
            // the last block: ln_f + pooling scores + per-tile pooling partials on the tile still on chip (tail32.hip T32_SCORE)
            const int nt32 = (L + T32_TILE - 1) / T32_TILE;
            const Tail32Score ts{x3 ? h->packed_score32x : h->packed_score32t, W(h, "head.attention.0.bias"), W(h, "head.attention.2.weight"),
                                 W(h, "head.attention.2.bias"), W(h, "bb.ln_f.weight"), W(h, "bb.ln_f.bias"), h->scores, h->partial};
            launch_tail32(reinterpret_cast<const float*>(h->y), h->h, lw.t_out, lw.t_fc1, lw.t_fc2, nx ? nx->t_in : nullptr, lw.b_out,
                          lw.b_fc1, lw.b_fc2, nx ? nx->b_in : nullptr, lw.ln2_g, lw.ln2_b, nx ? nx->ln1_g : nullptr,
                          nx ? nx->ln1_b : nullptr, reinterpret_cast<float*>(h->z), Bc, L, Lp, eps, st, x3, pad_skip ? h->pad_p0 : nullptr,
                          nx ? nullptr : &ts, (i == 0 && idconv) ? h->ids8 : nullptr, W(h, "bb.embeddings.word_embeddings.weight"));
            if (h->capture) {
                if (nx) HIPCHK(h, hipMemcpyAsync(h->capture->z[i + 1], h->z, (size_t)D3 * Lp * 4, hipMemcpyDeviceToDevice, st));
                else {
                    HIPCHK(h, hipMemcpyAsync(h->capture->hfin, h->h, (size_t)L * D * 4, hipMemcpyDeviceToDevice, st));
                    HIPCHK(h, hipMemcpyAsync(h->capture->scores, h->scores, (size_t)L * 4, hipMemcpyDeviceToDevice, st));
                    HIPCHK(h, hipMemcpyAsync(h->capture->partial, h->partial, (size_t)nt32 * POOL_PSTRIDE * 4, hipMemcpyDeviceToDevice, st));
                }
            } else if (pad_skip) {
                if (nx) launch_prefix_fill_z(h->pad_p0, h->z, ptab->z[i + 1], Bc, Lp, ptab->Lp, L, 4, D3, 0, st);
                else {
                    launch_prefix_fill_h(h->pad_p0, h->h, ptab->hfin, Bc, L, L, st);
                    launch_prefix_fill_pool(h->pad_p0, h->scores, h->partial, ptab->scores, ptab->partial, Bc, L, nt32, L, st, 128 / T32_TILE);
                }
            }
        } else {
            {
                StageTimer t(h, st, CLM_STAGE_OUTPROJ);
                if (tuned16) launch_outproj16(prec, h->y, lw.w_out, lw.b_out, h->h, Bc, L, Lp, st);
                else launch_outproj(prec, h->y, lw.w_out, lw.b_out, h->h, Bc, L, Lp, st);
            }
            if (stop_mid) return CLM_OK;
            if (tuned16) {   // fc1 + GELU + fc2 + residual fused
                StageTimer t(h, st, CLM_STAGE_MLP);
                launch_mlp16(prec, h->h, lw.ln2_g, lw.ln2_b, lw.w_fc1, lw.b_fc1, lw.w_fc2, lw.b_fc2, Bc, L, eps, st);
            } else {
                {
                    StageTimer t(h, st, CLM_STAGE_FC1);
                    launch_fc1(prec, h->h, lw.ln2_g, lw.ln2_b, lw.w_fc1, lw.b_fc1, h->u, Bc, L, eps, st);
                }
                if (stop_here(h, i, CLM_STAGE_FC1)) return CLM_OK;
                StageTimer t(h, st, CLM_STAGE_FC2);
                launch_fc2(prec, h->u, lw.w_fc2, lw.b_fc2, h->h, Bc, L, st);
            }
        }
        if (stop_here(h, i, CLM_STAGE_FC2) || (tuned16 && stop_here(h, i, CLM_STAGE_FC1))) return CLM_OK;
    }
    if (tuned16) {   // score + pooling partials in one pass over h, merged by the classifier kernel
        if (!fuse_next) {
            StageTimer t(h, st, CLM_STAGE_SCORE);
            launch_score_pool16(prec, h->h, W(h, "bb.ln_f.weight"), W(h, "bb.ln_f.bias"), packed_score,
                                W(h, "head.attention.0.bias"), W(h, "head.attention.2.weight"),
                                W(h, "head.attention.2.bias"), h->scores, h->partial, Bc, L, eps, st);
        }
        StageTimer t(h, st, CLM_STAGE_HEADMLP);
        launch_head_tiles(h->partial, (L + 127) / 128, h->hw, h->pooled, logits, Bc, st);
    } else if (fused32) {   // scores and per-tile pooling partials came out of the last block's tail kernel
        StageTimer t(h, st, CLM_STAGE_HEADMLP);
        launch_head_tiles(h->partial, (L + T32_TILE - 1) / T32_TILE, h->hw, h->pooled, logits, Bc, st);
    } else {
        {
            StageTimer t(h, st, CLM_STAGE_SCORE);
            launch_score(prec, h->h, W(h, "bb.ln_f.weight"), W(h, "bb.ln_f.bias"), packed_score,
                         W(h, "head.attention.0.bias"), W(h, "head.attention.2.weight"), W(h, "head.attention.2.bias"),
                         h->scores, Bc, L, eps, st);
        }
        {
            StageTimer t(h, st, CLM_STAGE_POOL);
            launch_softmax_stats(h->scores, h->stats, Bc, L, st);
            launch_pool(h->h, W(h, "bb.ln_f.weight"), W(h, "bb.ln_f.bias"), h->scores, h->stats, h->partial, Bc, L, eps,
                        st);
        }
        StageTimer t(h, st, CLM_STAGE_HEADMLP);
        launch_head_mlp(h->partial, h->hw, h->pooled, logits, Bc, st);
    }
    HIPCHK(h, hipGetLastError());
    return CLM_OK;
}

}  // namespace

// ======================================================================================== C ABI
extern "C" {

int clm_abi_version(void) { return CLM_ABI_VERSION; }

int clm_default_config(clm_config* c) {
    if (!c) return CLM_E_INVALID;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(clm_config);
    c->d_model = D; c->n_layer = NLAYER; c->d_inner = DI; c->vocab_rows = VOCAB; c->filter_order = FORDER;
    c->emb_dim = EMB; c->max_seq_len = 32770; c->head_hidden = HH; c->n_classes = NCLS;
    c->ln_eps = 1e-5f;
    c->precision = CLM_PREC_F32;
    c->chunk_reads = 256;
    return CLM_OK;
}

int clm_create(const clm_config* cfg, int device, clm_handle** out) {
    if (!cfg || !out) return fail(nullptr, CLM_E_INVALID, "clm_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(clm_config)) return fail(nullptr, CLM_E_INVALID, "clm_config size mismatch");
    if (cfg->d_model != D || cfg->n_layer != NLAYER || cfg->d_inner != DI || cfg->vocab_rows != VOCAB ||
        cfg->filter_order != FORDER || cfg->emb_dim != EMB || cfg->head_hidden != HH || cfg->n_classes != NCLS)
        return fail(nullptr, CLM_E_UNSUPPORTED,
                    "only the HyenaDNA-small-32k + 512-wide attention-pooling head of chimeralm/models/lm.py is built");
    if (cfg->precision < CLM_PREC_F32 || cfg->precision > CLM_PREC_F16X3 || cfg->chunk_reads < 1 ||
        cfg->max_seq_len < 2)
        return fail(nullptr, CLM_E_INVALID, "clm_create: bad precision / chunk_reads / max_seq_len");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(nullptr, CLM_E_HIP, "clm_create: no such HIP device " + std::to_string(device));
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, CLM_E_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, CLM_E_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, CLM_E_UNSUPPORTED, std::string("this engine is built for gfx950 (MI355X) only, found ") + prop.gcnArchName);
    clm_handle* h = new clm_handle();
    // developer switches (A/B runs, tests): ONE variable, CLM_DEBUG, a comma-separated list read when a handle is created
    // (clm_common.h debug_flag) -- no product behaviour hangs on the environment
    h->force_generic = debug_flag("generic_gemm");     // 16-bit modes through the generic kernels
    h->unfused_fp32 = debug_flag("unfused_fp32");      // exact fp32 through the separate GEMM kernels instead of tail32_kernel
    h->no_fuse_next = debug_flag("no_fuse_next");      // separate in_proj / score kernels instead of fusing them into the tail
    h->no_idconv = debug_flag("no_idconv");            // block 0's in_proj instead of the id-table convolution
    h->split_tail = debug_flag("split_tail");          // separate out_proj16 + mlp16 kernels instead of the fused tail
    h->no_lone_peel = debug_flag("no_lone_peel");      // the lone last token of 128 k + 1-token reads in a tile of its own
    if (debug_flag("conv_oneshot")) h->conv_flags |= CONV_ONESHOT;
    if (debug_flag("conv_no_xcd")) h->conv_flags |= CONV_NO_XCD;
    h->raw_z = debug_flag("raw_z");                    // the fused in_proj stage writes x0 | x1 | v as before round 3
    h->no_pad_skip = debug_flag("no_pad_skip");        // tiles inside a read's [PAD] prefix are computed like any other
    h->no_seg_skip = debug_flag("no_seg_skip");        // ... and segments inside it are transformed like any other (pad_prefix.hip, SegPrefix)
    h->cfg = *cfg;
    if (cfg->precision == CLM_PREC_F16X3) {             // an exact-fp32 engine whose fused tails multiply hi + lo halfs
        h->x3 = true;
        h->cfg.precision = CLM_PREC_F32;
        if (h->unfused_fp32) { delete h; return fail(nullptr, CLM_E_UNSUPPORTED, "CLM_PREC_F16X3 exists in the fused tail kernels only (CLM_DEBUG=unfused_fp32 is set)"); }
    }
    h->device = device;
    if (hipHostMalloc((void**)&h->bad_ids, sizeof(int), hipHostMallocMapped) == hipSuccess) *h->bad_ids = 0;
    else h->bad_ids = nullptr;
    *out = h;
    return CLM_OK;
}

int clm_load_weight(clm_handle* h, const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!h || !key || !data || !shape || ndim < 1 || ndim > 4) return fail(h, CLM_E_INVALID, "clm_load_weight: bad argument");
    std::string ck;
    if (!canonical_key(key, ck)) return fail(h, CLM_E_INVALID, std::string("unknown weight key: ") + key);
    if (ck.find("implicit_filter.3.freq") != std::string::npos || ck.find("implicit_filter.5.freq") != std::string::npos)
        return CLM_OK;  // aliases of the shared sine module's parameter (implicit_filter.1.freq)
    const KeySpec* spec = nullptr;
    static thread_local std::vector<KeySpec> specs;
    specs = expected_keys(h->cfg);
    for (auto& s : specs)
        if (s.key == ck) spec = &s;
    if (!spec) return fail(h, CLM_E_INVALID, std::string("unknown weight key: ") + key);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != spec->shape) {
        std::string got, want;
        for (auto v : shp) got += std::to_string(v) + ",";
        for (auto v : spec->shape) want += std::to_string(v) + ",";
        return fail(h, CLM_E_INVALID, std::string(key) + ": shape [" + got + "] does not match [" + want + "]");
    }
    size_t n = 1;
    for (auto v : shp) n *= (size_t)v;
    HIPCHK(h, hipSetDevice(h->device));
    Tensor& t = h->w[ck];
    if (!t.d) HIPCHK(h, hipMalloc((void**)&t.d, n * 4));
    t.shape = shp;
    t.numel = n;
    if (dtype == CLM_DT_F32) {
        HIPCHK(h, hipMemcpy(t.d, data, n * 4, hipMemcpyDefault));
    } else if (dtype == CLM_DT_F64 || dtype == CLM_DT_BF16 || dtype == CLM_DT_F16) {
        size_t es = dtype == CLM_DT_F64 ? 8 : 2;
        void* stage = nullptr;
        HIPCHK(h, hipMalloc(&stage, n * es));
        HIPCHK(h, hipMemcpy(stage, data, n * es, hipMemcpyDefault));
        hipLaunchKernelGGL(convert_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, stage, t.d, n, dtype);
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipFree(stage));
    } else {
        return fail(h, CLM_E_INVALID, "clm_load_weight: dtype must be f32/f64/bf16/f16");
    }
    t.loaded = true;
    h->finalized = false;
    return CLM_OK;
}

int clm_finalize(clm_handle* h) {
    if (!h) return CLM_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    for (auto& s : expected_keys(h->cfg)) {
        auto it = h->w.find(s.key);
        if (it == h->w.end() || !it->second.loaded) return fail(h, CLM_E_MISSING, "missing weight: " + s.key);
    }
    HIPCHK(h, hipDeviceSynchronize());
    free_packed(h);
    free_filters(h);
    free_pad_tables(h);                                      // (functions of the weights)
    const int prec = h->cfg.precision;
    // hi + lo halfs of the tail weights: an fp16x3 handle's own arithmetic; a 16-bit handle's short reads and first fall-back level
    const bool pack_x3 = (h->x3 || prec != PREC_F32) && !h->unfused_fp32;
    hipStream_t st = 0;
    auto pack_as = [&](int pr, const std::string& key, int n, int k, void** out) -> int {
        HIPCHK(h, hipMalloc(out, packed_weight_bytes(pr, n, k)));
        HIPCHK(h, hipMemsetAsync(*out, 0, packed_weight_bytes(pr, n, k), st));
        launch_pack_weight(pr, W(h, key), *out, n, k, st);
        return CLM_OK;
    };
    auto pack = [&](const std::string& key, int n, int k, void** out) -> int { return pack_as(prec, key, n, k, out); };
    for (int i = 0; i < NLAYER; ++i) {
        std::string p = "bb.layers." + std::to_string(i) + ".";
        int rc;
        if ((rc = pack(p + "mixer.in_proj.weight", D3, D, &h->packed[i][0]))) return rc;
        if ((rc = pack(p + "mixer.out_proj.weight", D, D, &h->packed[i][1]))) return rc;
        // (fp16c: the two MLP products run on plain fp16 weights -- tail16_kernel, MLP_PREC)
        const int mlp_prec = prec == PREC_F16C ? (int)PREC_F16 : prec;
        if ((rc = pack_as(mlp_prec, p + "mlp.fc1.weight", DI, D, &h->packed[i][2]))) return rc;
        if ((rc = pack_as(mlp_prec, p + "mlp.fc2.weight", D, DI, &h->packed[i][3]))) return rc;
        if (prec == PREC_F16C) {
            if ((rc = pack_as(PREC_F16C, p + "mlp.fc1.weight", DI, D, &h->packed_mlpc[i][0]))) return rc;
            if ((rc = pack_as(PREC_F16C, p + "mlp.fc2.weight", D, DI, &h->packed_mlpc[i][1]))) return rc;
        }
        LayerW& lw = h->lw[i];
        lw.ln1_g = W(h, p + "norm1.weight"); lw.ln1_b = W(h, p + "norm1.bias");
        lw.ln2_g = W(h, p + "norm2.weight"); lw.ln2_b = W(h, p + "norm2.bias");
        lw.w_in = h->packed[i][0]; lw.w_out = h->packed[i][1]; lw.w_fc1 = h->packed[i][2]; lw.w_fc2 = h->packed[i][3];
        lw.b_in = W(h, p + "mixer.in_proj.bias"); lw.b_out = W(h, p + "mixer.out_proj.bias");
        lw.b_fc1 = W(h, p + "mlp.fc1.bias"); lw.b_fc2 = W(h, p + "mlp.fc2.bias");
        lw.short_w = W(h, p + "mixer.short_filter.weight"); lw.short_b = W(h, p + "mixer.short_filter.bias");
        lw.filt_bias = W(h, p + "mixer.filter_fn.bias");
        {   // exact fp32 (the engine's own mode, or the referee / short-read / fall-back path of a 16-bit engine): the fused tail's packing
            struct { const char* key; int n, k; } tw[4] = {{"mixer.in_proj.weight", D3, D}, {"mixer.out_proj.weight", D, D},
                                                           {"mlp.fc1.weight", DI, D}, {"mlp.fc2.weight", D, DI}};
            for (int j = 0; j < 4; ++j) {
                HIPCHK(h, hipMalloc(&h->packed32t[i][j], (size_t)tw[j].n * tw[j].k * 4));
                launch_pack_f32t(W(h, p + tw[j].key), h->packed32t[i][j], tw[j].n, tw[j].k, st);
                if (pack_x3) {
                    HIPCHK(h, hipMalloc(&h->packed32x[i][j], (size_t)tw[j].n * tw[j].k * 4));
                    launch_pack_x3(W(h, p + tw[j].key), h->packed32x[i][j], tw[j].n, tw[j].k, st);
                }
            }
            lw.t_in = h->packed32t[i][0]; lw.t_out = h->packed32t[i][1]; lw.t_fc1 = h->packed32t[i][2]; lw.t_fc2 = h->packed32t[i][3];
        }
        if (prec != PREC_F32) {   // the exact-fp32 packing next to the 16-bit one: fp16c's short reads, clm_selfcheck, clm_set_fallback
            if ((rc = pack_as(PREC_F32, p + "mixer.in_proj.weight", D3, D, &h->packed32[i][0]))) return rc;
            if ((rc = pack_as(PREC_F32, p + "mixer.out_proj.weight", D, D, &h->packed32[i][1]))) return rc;
            if ((rc = pack_as(PREC_F32, p + "mlp.fc1.weight", DI, D, &h->packed32[i][2]))) return rc;
            if ((rc = pack_as(PREC_F32, p + "mlp.fc2.weight", D, DI, &h->packed32[i][3]))) return rc;
            h->lw32[i] = lw;
            h->lw32[i].w_in = h->packed32[i][0]; h->lw32[i].w_out = h->packed32[i][1];
            h->lw32[i].w_fc1 = h->packed32[i][2]; h->lw32[i].w_fc2 = h->packed32[i][3];
        }
        if (pack_x3) {            // the fp32 path's LayerW with the fused tail's weights as hi + lo halfs
            h->lwx[i] = prec != PREC_F32 ? h->lw32[i] : lw;
            h->lwx[i].t_in = h->packed32x[i][0]; h->lwx[i].t_out = h->packed32x[i][1];
            h->lwx[i].t_fc1 = h->packed32x[i][2]; h->lwx[i].t_fc2 = h->packed32x[i][3];
        }
    }
    {
        int rc;
        if ((rc = pack("head.attention.0.weight", D, D, &h->packed_score))) return rc;
        if (prec != PREC_F32 && (rc = pack_as(PREC_F32, "head.attention.0.weight", D, D, &h->packed_score32))) return rc;
        HIPCHK(h, hipMalloc(&h->packed_score32t, (size_t)D * D * 4));
        launch_pack_f32t(W(h, "head.attention.0.weight"), h->packed_score32t, D, D, st);
        if (pack_x3) {
            HIPCHK(h, hipMalloc(&h->packed_score32x, (size_t)D * D * 4));
            launch_pack_x3(W(h, "head.attention.0.weight"), h->packed_score32x, D, D, st);
        }
    }
    if (prec != PREC_F32)
        for (int i = 0; i < NLAYER; ++i) {
            HIPCHK(h, hipMalloc((void**)&h->fir[i], (size_t)D * 3 * sizeof(float4)));
            launch_fir_table(h->lw[i].short_w, h->lw[i].short_b, h->lw[i].b_in, h->fir[i], st);
        }
    HIPCHK(h, hipMalloc((void**)&h->ztab, (size_t)VOCAB * D3 * 4));
    launch_ztab(W(h, "bb.embeddings.word_embeddings.weight"), W(h, "bb.layers.0.norm1.weight"),
                W(h, "bb.layers.0.norm1.bias"), W(h, "bb.layers.0.mixer.in_proj.weight"),
                W(h, "bb.layers.0.mixer.in_proj.bias"), h->ztab, h->cfg.ln_eps, st);
    struct { const char* key; int rows, cols; } tr[5] = {
        {"head.classifier.0.weight", HH, D}, {"head.classifier.3.weight", HH, HH},
        {"head.classifier.6.layers.0.weight", HH, HH}, {"head.classifier.6.layers.3.weight", HH, HH},
        {"head.output_layer.weight", NCLS, HH}};
    for (int j = 0; j < 5; ++j) {
        HIPCHK(h, hipMalloc((void**)&h->head_t[j], (size_t)tr[j].rows * tr[j].cols * 4));
        launch_transpose(W(h, tr[j].key), h->head_t[j], tr[j].rows, tr[j].cols, st);
    }
    h->hw.w0t = h->head_t[0]; h->hw.b0 = W(h, "head.classifier.0.bias");
    h->hw.w3t = h->head_t[1]; h->hw.b3 = W(h, "head.classifier.3.bias");
    h->hw.w60t = h->head_t[2]; h->hw.b60 = W(h, "head.classifier.6.layers.0.bias");
    h->hw.w63t = h->head_t[3]; h->hw.b63 = W(h, "head.classifier.6.layers.3.bias");
    h->hw.wot = h->head_t[4]; h->hw.bo = W(h, "head.output_layer.bias");
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipDeviceSynchronize());
    h->finalized = true;
    return CLM_OK;
}

int clm_reserve(clm_handle* h, int B, int L) {
    if (!h || B < 1 || L < 1) return fail(h, CLM_E_INVALID, "clm_reserve: bad argument");
    if (!h->finalized) return fail(h, CLM_E_STATE, "clm_reserve before clm_finalize");
    HIPCHK(h, hipSetDevice(h->device));
    FilterSet* fs = nullptr;
    const ReversedFilter* kr = nullptr;
    int rc = ensure_filters(h, L, 0, &fs, &kr);
    if (rc) return rc;
    const int chunk = chunk_for(h, L);
    int Bc = B < chunk ? B : chunk;
    return ensure_workspace(h, Bc, L, 0);
}

int clm_forward(clm_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L,
                float* logits_out, void* stream) {
    if (!h) return CLM_E_INVALID;
    if (!h->finalized) return fail(h, CLM_E_STATE, "clm_forward before clm_finalize");
    if (!ids || !logits_out || B < 1 || L < 1 || ids_row_stride < L)
        return fail(h, CLM_E_INVALID, "clm_forward: bad argument");
    if (ids_dtype != CLM_DT_I64 && ids_dtype != CLM_DT_I32 && ids_dtype != CLM_DT_U8)
        return fail(h, CLM_E_INVALID, "clm_forward: ids dtype must be i64, i32 or u8");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = check_bad_ids(h)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t ies = ids_dtype == CLM_DT_I64 ? 8 : (ids_dtype == CLM_DT_I32 ? 4 : 1);
    const int chunk = chunk_for(h, L);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        int Bc = B - b0 < chunk ? B - b0 : chunk;
        const char* p = reinterpret_cast<const char*>(ids) + (size_t)b0 * ids_row_stride * ies;
        int rc = forward_chunk(h, p, ids_dtype, ids_row_stride, Bc, L, logits_out + (size_t)b0 * NCLS, st);
        if (rc) return rc;
    }
    return CLM_OK;
}

int clm_stage_ids(clm_handle* h, const void* host_ids, int ids_dtype, int64_t ids_row_stride, int B, int L, int* staged) {
    if (!h) return CLM_E_INVALID;
    if (!host_ids || !staged || B < 1 || L < 1 || ids_row_stride < L)
        return fail(h, CLM_E_INVALID, "clm_stage_ids: bad argument");
    if (ids_dtype != CLM_DT_I64 && ids_dtype != CLM_DT_I32 && ids_dtype != CLM_DT_U8)
        return fail(h, CLM_E_INVALID, "clm_stage_ids: ids dtype must be i64, i32 or u8");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    const int k = h->next_stage;
    clm_handle::Stage& s = h->stage[k];
    if (s.pending) return fail(h, CLM_E_STATE, "clm_stage_ids: both staging buffers hold batches not yet run (clm_forward_staged)");
    const size_t ies = ids_dtype == CLM_DT_I64 ? 8 : (ids_dtype == CLM_DT_I32 ? 4 : 1);
    const size_t bytes = (size_t)B * (size_t)ids_row_stride * ies;
    if (!s.copied) {
        HIPCHK(h, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    if (s.used) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, s.consumed, 0));   // the forward that read this buffer is done
    if (bytes > s.cap) {
        if (s.buf) {
            HIPCHK(h, hipEventSynchronize(s.consumed));
            HIPCHK(h, hipFree(s.buf));
            s.buf = nullptr;
        }
        HIPCHK(h, hipMalloc(&s.buf, bytes));
        s.cap = bytes;
    }
    HIPCHK(h, hipMemcpyAsync(s.buf, host_ids, bytes, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(s.copied, h->copy_stream));
    s.dtype = ids_dtype; s.B = B; s.L = L; s.stride = ids_row_stride;
    s.pending = true;
    h->next_stage = k ^ 1;
    *staged = k;
    return CLM_OK;
}

int clm_forward_staged(clm_handle* h, int staged, float* logits_out, void* stream) {
    if (!h) return CLM_E_INVALID;
    if (staged < 0 || staged > 1 || !h->stage[staged].pending)
        return fail(h, CLM_E_STATE, "clm_forward_staged: no batch staged in that buffer");
    clm_handle::Stage& s = h->stage[staged];
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIPCHK(h, hipStreamWaitEvent(st, s.copied, 0));
    const int rc = clm_forward(h, s.buf, s.dtype, s.stride, s.B, s.L, logits_out, stream);
    // whatever happened, the buffer is no longer "staged and waiting": a failed forward must not wedge it for good
    s.pending = false;
    s.used = true;
    (void)hipEventRecord(s.consumed, st);
    return rc;
}

int clm_stage_wait(clm_handle* h, int staged) {
    if (!h) return CLM_E_INVALID;
    if (staged < 0 || staged > 1 || !h->stage[staged].copied) return fail(h, CLM_E_STATE, "clm_stage_wait: nothing was staged there");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->stage[staged].copied));
    return check_bad_ids(h);
}

int clm_check(clm_handle* h, void* stream) {
    if (!h) return CLM_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return check_bad_ids(h);
}

int clm_selfcheck(clm_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, void* stream,
                  float* max_abs_diff, int* labels_differ) {
    if (!h) return CLM_E_INVALID;
    if (!h->finalized) return fail(h, CLM_E_STATE, "clm_selfcheck before clm_finalize");
    if (!ids || !max_abs_diff || B < 1 || L < 1 || ids_row_stride < L) return fail(h, CLM_E_INVALID, "clm_selfcheck: bad argument");
    if (ids_dtype != CLM_DT_I64 && ids_dtype != CLM_DT_I32 && ids_dtype != CLM_DT_U8)
        return fail(h, CLM_E_INVALID, "clm_selfcheck: ids dtype must be i64, i32 or u8");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    *max_abs_diff = 0.f;
    if (labels_differ) *labels_differ = 0;
    if (h->cfg.precision == PREC_F32 && !h->x3) return CLM_OK;   // the handle IS the referee
    if (B > h->sc_cap) {
        HIPCHK(h, hipStreamSynchronize(st));
        if (h->sc_logits) HIPCHK(h, hipFree(h->sc_logits));
        h->sc_logits = nullptr;
        HIPCHK(h, hipMalloc((void**)&h->sc_logits, (size_t)2 * B * NCLS * 4));
        h->sc_cap = B;
    }
    // pass 0: the arithmetic the handle's mode runs reads of this length in (whatever clm_set_fallback says); pass 1: exact fp32
    const int mode_prec = (h->cfg.precision == PREC_F16C && L < h->f16c_min_len) ? (int)PREC_F32 : h->cfg.precision;
    const bool prof = h->prof;
    const int fallback = h->fallback;
    h->prof = false;                                            // not part of anybody's timed region
    h->fallback = 0;                                            // the MODE is on trial (short reads of fp16c: its fp16x3 kernels)
    const size_t ies = ids_dtype == CLM_DT_I64 ? 8 : (ids_dtype == CLM_DT_I32 ? 4 : 1);
    int rc = CLM_OK;
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        h->force_prec = pass == 0 ? mode_prec : (int)PREC_F32;
        h->referee = pass == 1;
        const int chunk = chunk_for(h, L);
        for (int b0 = 0; b0 < B && !rc; b0 += chunk) {
            const int Bc = B - b0 < chunk ? B - b0 : chunk;
            const char* p = reinterpret_cast<const char*>(ids) + (size_t)b0 * ids_row_stride * ies;
            rc = forward_chunk(h, p, ids_dtype, ids_row_stride, Bc, L, h->sc_logits + ((size_t)pass * h->sc_cap + b0) * NCLS, st);
        }
    }
    h->force_prec = -1;
    h->referee = false;
    h->prof = prof;
    h->fallback = fallback;
    if (rc) return rc;
    std::vector<float> host((size_t)2 * h->sc_cap * NCLS);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->sc_logits, host.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    float worst = 0.f;
    int differ = 0;
    (void)clm_logit_deviation(host.data(), host.data() + (size_t)h->sc_cap * NCLS, B, NCLS, &worst, &differ);
    *max_abs_diff = worst;
    if (labels_differ) *labels_differ = differ;
    return CLM_OK;
}

int clm_logit_deviation(const float* a, const float* b, int B, int n_classes, float* max_abs_diff, int* labels_differ) {
    if (!a || !b || !max_abs_diff || B < 0 || n_classes < 1) return CLM_E_INVALID;
    float worst = 0.f;
    int differ = 0;
    for (int r = 0; r < B; ++r) {
        const float *x = a + (size_t)r * n_classes, *y = b + (size_t)r * n_classes;
        int ax = 0, ay = 0;
        for (int c = 0; c < n_classes; ++c) {
            const float d = std::fabs(x[c] - y[c]);
            if (!(d <= 3.0e38f)) worst = INFINITY;             // NaN or inf anywhere = infinitely wrong, and it STAYS so
            else if (d > worst) worst = d;
            if (x[c] > x[ax]) ax = c;
            if (y[c] > y[ay]) ay = c;
        }
        differ += ax != ay;
    }
    *max_abs_diff = worst;
    if (labels_differ) *labels_differ = differ;
    return CLM_OK;
}

int clm_set_fallback(clm_handle* h, int on) {
    if (!h) return CLM_E_INVALID;
    if (!h->finalized) return fail(h, CLM_E_STATE, "clm_set_fallback before clm_finalize");
    if (on < 0 || on > 2) return fail(h, CLM_E_INVALID, "clm_set_fallback: level must be 0, 1 or 2");
    h->fallback = (h->cfg.precision == PREC_F32 && !h->x3) ? 0 : on;   // (an exact-fp32 handle has nothing to fall back to)
    return CLM_OK;
}

int clm_set_mlp_compensation(clm_handle* h, int on) {
    if (!h) return CLM_E_INVALID;
    if (h->cfg.precision != PREC_F16C) return fail(h, CLM_E_UNSUPPORTED, "clm_set_mlp_compensation: not a CLM_PREC_F16C handle");
    h->mlp_lo = on != 0;
    return CLM_OK;
}

int clm_set_short_read_len(clm_handle* h, int min_len) {
    if (!h || min_len < 1) return fail(h, CLM_E_INVALID, "clm_set_short_read_len: bad argument");
    if (h->cfg.precision != PREC_F16C) return fail(h, CLM_E_UNSUPPORTED, "clm_set_short_read_len: not a CLM_PREC_F16C handle");
    h->f16c_min_len = min_len;
    return CLM_OK;
}

int clm_effective_precision(const clm_handle* h, int L) {
    if (!h || L < 1) return CLM_E_INVALID;
    const int p = effective_prec(h, L);
    return (p == PREC_F32 && fused_fp32(h) && fp32_path_is_x3(h)) ? CLM_PREC_F16X3 : p;
}

int clm_debug_stop_after(clm_handle* h, int layer, int stage) {
    if (!h) return CLM_E_INVALID;
    h->stop_layer = layer;
    h->stop_stage = stage;
    return CLM_OK;
}

int clm_debug_fetch(clm_handle* h, const char* name, void* host_out, size_t bytes) {
    if (!h || !name || !host_out) return fail(h, CLM_E_INVALID, "clm_debug_fetch: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t es = elem_size(effective_prec(h, h->last_L));
    const size_t B = h->last_B, L = h->last_L, Lp = h->last_Lp;
    const void* src = nullptr;
    size_t have = 0;
    std::string n(name);
    if (n == "hidden" || n == "h") { src = h->h; have = B * L * D * 4; }
    else if (n == "z") { src = h->z; have = B * D3 * Lp * es; }
    else if (n == "y") { src = h->y; have = B * D * Lp * es; }
    else if (n == "u") { src = h->u; have = B * L * DI * es; }
    else if (n == "scores") { src = h->scores; have = B * L * 4; }
    else if (n == "pooled") { src = h->pooled; have = B * D * 4; }
    else if (n.rfind("filter.", 0) == 0) {
        int i = std::atoi(n.c_str() + 7);
        const int key = conv_segments_for((int)L) > 1 ? KEY_LONG : conv_logn_for((int)L);
        for (auto& f : h->filters)          // the first L taps of the class's filter (they do not depend on L)
            if (f.key == key && i >= 0 && i < NLAYER && (int)L <= f.Lf) { src = f.ktime[i]; have = L * D * 4; }
    }
    if (!src) return fail(h, CLM_E_INVALID, "clm_debug_fetch: unknown or empty buffer " + n);
    if (bytes > have) return fail(h, CLM_E_INVALID, "clm_debug_fetch: " + n + " holds only " + std::to_string(have) + " bytes");
    HIPCHK(h, hipMemcpy(host_out, src, bytes, hipMemcpyDeviceToHost));
    return CLM_OK;
}

int clm_profile_enable(clm_handle* h, int on) {
    if (!h) return CLM_E_INVALID;
    h->prof = on != 0;
    return CLM_OK;
}

int clm_profile_read(clm_handle* h, double* ms_out, int64_t* launches_out, int reset) {
    if (!h) return CLM_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    for (auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            h->prof_ms[r.stage] += ms;
            h->prof_n[r.stage] += 1;
        }
        h->free_events.push_back({r.e0, r.e1});
    }
    h->recs.clear();
    for (int i = 0; i < CLM_N_STAGES; ++i) {
        if (ms_out) ms_out[i] = h->prof_ms[i];
        if (launches_out) launches_out[i] = h->prof_n[i];
        if (reset) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
    }
    return CLM_OK;
}

const char* clm_profile_stage_name(int stage) {
    static const char* names[CLM_N_STAGES] = {"embed", "ln1_in_proj", "short_long_conv", "out_proj", "ln2_fc1_gelu",
                                              "fc2", "lnf_pool_score", "softmax_pool", "head_mlp", "filter",
                                              "out_proj_ln2_mlp", "ln2_mlp"};
    return (stage >= 0 && stage < CLM_N_STAGES) ? names[stage] : "?";
}

const char* clm_last_error(const clm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int clm_destroy(clm_handle* h) {
    if (!h) return CLM_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    tail16_dump_stamps();
    conv_dump_stamps();
    for (auto& r : h->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto& e : h->free_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    free_workspace(h);
    free_filters(h);
    free_packed(h);
    free_pad_tables(h);
    if (h->pad_ids) (void)hipFree(h->pad_ids);
    if (h->pad_logits) (void)hipFree(h->pad_logits);
    if (h->sc_logits) (void)hipFree(h->sc_logits);
    if (h->edge_bnd) (void)hipFree(h->edge_bnd);
    for (auto& sg : h->stage) {
        if (sg.buf) (void)hipFree(sg.buf);
        if (sg.copied) (void)hipEventDestroy(sg.copied);
        if (sg.consumed) (void)hipEventDestroy(sg.consumed);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->bad_ids) (void)hipHostFree(h->bad_ids);
    for (auto& kv : h->w)
        if (kv.second.d) (void)hipFree(kv.second.d);
    delete h;
    return CLM_OK;
}

}  // extern "C"
