// bam_feeder.cpp -- native BAM feeder of the predict path (include/chimeralm_feed.h).
//
// Host-side mirror of the reference's Python data path, file:line under /root/reference/chimeralm/:
//   data/bam.py:21-38 (read selection), data/tokenizer.py:85-114 (token stream + id row), :136-187 (collation),
//   data/bam.py:142-174 (per-device batches), SURVEY.md Appendix B (rank r sees selected reads r, r+G, ...).
// BGZF / BAM layouts follow the SAM/BAM format specification (sections 4.1 "The BGZF compression format" and 4.2 "The BAM
// format"): a BGZF file is a series of gzip members whose extra field carries the member size ("BC" subfield); the
// inflated stream is  magic "BAM\1" | l_text, text | n_ref, (l_name, name, l_ref)*  followed by alignment records
//   block_size | refID pos | l_read_name mapq bin | n_cigar_op flag | l_seq | next_refID next_pos tlen | read_name\0 |
//   cigar[4*n_cigar_op] | seq[(l_seq+1)/2] (4-bit codes "=ACMGRSVTWYHKDBN", high nibble first) | qual[l_seq] | aux fields.
//
// One decoder thread produces batches into a ring of host slots (page-locked when cfg.pinned); the consumer thread takes
// them in order (clm_feeder_next) and returns them (clm_feeder_release).  Producer and consumer meet only at the two ring
// counters; a condition variable parks whichever side has nothing to do.
#include "chimeralm_feed.h"
#include "chimeralm_hip.h"

#include <hip/hip_runtime_api.h>

#include "bgzf.h"

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int ID_ROW = 256;                 // MAX_ID_LENGTH, tokenizer.py:111
constexpr uint8_t SEP_ID = 1, PAD_ID = 4, UNK_ID = 6;
constexpr uint16_t FLAG_UNMAPPED = 0x4, FLAG_SECONDARY = 0x100, FLAG_SUPPLEMENTARY = 0x800;

std::string g_open_error;

// 4-bit base code -> token id.  The reference decodes the code to a character ("=ACMGRSVTWYHKDBN") and the tokenizer maps
// A C G T N to 7..11 and everything else to [UNK] = 6 (vocabulary tokenizer.py:230-239).
constexpr uint8_t CODE_TO_ID[16] = {6, 7, 8, 6, 9, 6, 6, 6, 10, 6, 6, 6, 6, 6, 6, 11};

struct Slot {
    uint8_t* ids = nullptr;     // [batch_size][max_tokens]
    int8_t* names = nullptr;    // [batch_size][256]
    int n_reads = 0, n_tokens = 0;
    int64_t first_index = 0;
};

}  // namespace

struct clm_feeder {
    clm_feeder_config cfg{};
    std::string path, err;
    clmbgzf::ParallelReader rd;
    std::vector<Slot> ring;
    uint8_t* slab_ids = nullptr;
    int8_t* slab_names = nullptr;
    bool slab_pinned = false;
    // ring state: slot i is owned by the consumer while released <= i_seq < produced is false ... see below
    std::mutex mu;
    std::condition_variable cv;
    int64_t produced = 0;       // batches completely written by the decoder
    int64_t consumed = 0;       // batches handed to the consumer
    std::vector<char> busy;     // per slot: handed out and not yet released
    bool done = false, failed = false, stop = false;
    std::thread worker;
    std::atomic<int64_t> n_records{0}, n_selected{0}, n_delivered{0}, n_truncated{0};
};

namespace {

int fail_open(int code, const std::string& msg) {
    g_open_error = msg;
    return code;
}

using clmbgzf::le16;
using clmbgzf::le32;

// makes n inflated bytes available; forwards the reader's error text
int need(clm_feeder* f, size_t n) {
    const int rc = f->rd.need(n);
    if (rc < 0) f->err = f->rd.err;
    return rc;
}

// auxiliary fields: TAG(2) TYPE(1) VALUE; true if a field with tag SA exists (bam.py:21-23 `read.has_tag("SA")`)
int has_sa_tag(const uint8_t* aux, size_t n) {
    size_t p = 0;
    while (p + 3 <= n) {
        if (aux[p] == 'S' && aux[p + 1] == 'A') return 1;
        const char t = (char)aux[p + 2];
        p += 3;
        switch (t) {
            case 'A': case 'c': case 'C': p += 1; break;
            case 's': case 'S': p += 2; break;
            case 'i': case 'I': case 'f': p += 4; break;
            case 'Z': case 'H':
                while (p < n && aux[p] != 0) ++p;
                ++p;
                break;
            case 'B': {
                if (p + 5 > n) return -1;
                const char st = (char)aux[p];
                const int64_t cnt = (uint32_t)le32(aux + p + 1);
                const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
                if (!es) return -1;
                p += 5 + (size_t)(cnt * es);
                break;
            }
            default: return -1;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ decoder thread
struct PendingRead {
    std::vector<uint8_t> ids;   // token ids incl. [SEP]
    int8_t name[ID_ROW];
};

bool publish(clm_feeder* f, std::vector<PendingRead>& pend, int64_t& first_index) {
    // wait for the slot of batch number `produced` to be free
    const int nslots = (int)f->ring.size();
    std::unique_lock<std::mutex> lk(f->mu);
    const int64_t seq = f->produced;
    f->cv.wait(lk, [&] { return f->stop || (seq - f->consumed < nslots && !f->busy[(size_t)(seq % nslots)]); });
    if (f->stop) return false;
    lk.unlock();
    Slot& s = f->ring[(size_t)(seq % nslots)];
    int longest = 0;
    for (const PendingRead& r : pend) longest = (int)r.ids.size() > longest ? (int)r.ids.size() : longest;
    s.n_reads = (int)pend.size();
    s.n_tokens = longest;
    s.first_index = first_index;
    for (size_t i = 0; i < pend.size(); ++i) {
        uint8_t* row = s.ids + i * (size_t)longest;
        const size_t n = pend[i].ids.size(), padn = (size_t)longest - n;
        if (f->cfg.pad_left) {
            std::memset(row, PAD_ID, padn);
            std::memcpy(row + padn, pend[i].ids.data(), n);
        } else {
            std::memcpy(row, pend[i].ids.data(), n);
            std::memset(row + n, PAD_ID, padn);
        }
        std::memcpy(s.names + i * ID_ROW, pend[i].name, ID_ROW);
    }
    first_index += (int64_t)pend.size();
    f->n_delivered += (int64_t)pend.size();
    pend.clear();
    lk.lock();
    f->produced = seq + 1;
    lk.unlock();
    f->cv.notify_all();
    return true;
}

void decode_loop(clm_feeder* f) {
    auto finish = [&](bool failed) {
        std::lock_guard<std::mutex> lk(f->mu);
        f->failed = failed;
        f->done = true;
        f->cv.notify_all();
    };
    // header: magic was checked by open(); skip text and references
    if (need(f, 8) <= 0) return finish(true);
    const int32_t l_text = le32(f->rd.data() + 4);
    f->rd.advance(8);
    if (l_text < 0 || need(f, (size_t)l_text + 4) <= 0) { if (f->err.empty()) f->err = f->path + ": corrupt BAM header"; return finish(true); }
    f->rd.advance((size_t)l_text);
    const int32_t n_ref = le32(f->rd.data());
    f->rd.advance(4);
    for (int32_t i = 0; i < n_ref; ++i) {
        if (need(f, 4) <= 0) { if (f->err.empty()) f->err = f->path + ": corrupt BAM reference list"; return finish(true); }
        const int32_t l_name = le32(f->rd.data());
        f->rd.advance(4);
        if (l_name < 0 || need(f, (size_t)l_name + 4) <= 0) { if (f->err.empty()) f->err = f->path + ": corrupt BAM reference list"; return finish(true); }
        f->rd.advance((size_t)l_name + 4);
    }
    std::vector<PendingRead> pend;
    pend.reserve((size_t)f->cfg.batch_size);
    int64_t first_index = 0, selected = 0;
    const int max_bases = f->cfg.max_tokens - 1;
    while (true) {
        if (f->cfg.max_reads >= 0 && selected >= f->cfg.max_reads) break;
        int rc = need(f, 4);
        if (rc < 0) return finish(true);
        if (rc == 0) break;
        const int32_t block_size = le32(f->rd.data());
        f->rd.advance(4);
        if (block_size < 32 || need(f, (size_t)block_size) <= 0) {
            if (f->err.empty()) f->err = f->path + ": corrupt BAM record";
            return finish(true);
        }
        const uint8_t* rec = f->rd.data();
        f->rd.advance((size_t)block_size);
        ++f->n_records;
        const unsigned l_read_name = rec[8];
        const unsigned n_cigar = le16(rec + 12);
        const uint16_t flag = le16(rec + 14);
        const int64_t l_seq = le32(rec + 16);
        const size_t off_seq = 32 + (size_t)l_read_name + 4 * (size_t)n_cigar;
        const size_t off_aux = off_seq + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
        if (l_seq < 0 || l_read_name == 0 || off_aux > (size_t)block_size) {
            f->err = f->path + ": corrupt BAM record (field lengths exceed the record)";
            return finish(true);
        }
        if (flag & (FLAG_UNMAPPED | FLAG_SECONDARY | FLAG_SUPPLEMENTARY)) continue;
        const int sa = has_sa_tag(rec + off_aux, (size_t)block_size - off_aux);
        if (sa < 0) {
            f->err = f->path + ": corrupt BAM auxiliary field";
            return finish(true);
        }
        if (!sa) continue;
        const int64_t idx = selected++;
        ++f->n_selected;
        if (idx % f->cfg.world != f->cfg.rank) continue;
        pend.emplace_back();
        PendingRead& r = pend.back();
        const int64_t nb = l_seq < max_bases ? l_seq : max_bases;
        if (l_seq > nb) f->n_truncated += l_seq - nb;
        r.ids.resize((size_t)nb + 1);
        const uint8_t* sq = rec + off_seq;
        for (int64_t i = 0; i + 1 < nb; i += 2) {
            const uint8_t b = sq[i >> 1];
            r.ids[(size_t)i] = CODE_TO_ID[b >> 4];
            r.ids[(size_t)i + 1] = CODE_TO_ID[b & 15];
        }
        if (nb & 1) r.ids[(size_t)nb - 1] = CODE_TO_ID[sq[(nb - 1) >> 1] >> 4];
        r.ids[(size_t)nb] = SEP_ID;
        // id row: [len(name)] + code points, cut / zero-padded to 256 entries, stored as int8 (tokenizer.py:105-112,168)
        const size_t name_len = std::strlen(reinterpret_cast<const char*>(rec + 32)) < l_read_name - 1
                                    ? std::strlen(reinterpret_cast<const char*>(rec + 32)) : l_read_name - 1;
        std::memset(r.name, 0, ID_ROW);
        r.name[0] = (int8_t)(uint8_t)(name_len & 0xFF);
        std::memcpy(r.name + 1, rec + 32, name_len < ID_ROW - 1 ? name_len : ID_ROW - 1);
        if ((int)pend.size() == f->cfg.batch_size && !publish(f, pend, first_index)) return finish(false);
    }
    if (!pend.empty() && !publish(f, pend, first_index)) return finish(false);
    finish(false);
}

}  // namespace

extern "C" {

int clm_feeder_default_config(clm_feeder_config* cfg) {
    if (!cfg) return CLM_E_INVALID;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(*cfg);
    cfg->batch_size = 12;          // __main__.py:253
    cfg->max_tokens = 32769;       // hyenadna-small-32k tokenizer: max_len_single_sentence = model_max_length (32770) - 1
    cfg->slots = 4;
    cfg->rank = 0;
    cfg->world = 1;
    cfg->pad_left = 1;
    cfg->pinned = 1;
    cfg->max_reads = -1;
    cfg->inflate_threads = 0;      // automatic
    return CLM_OK;
}

int clm_feeder_open(const char* bam_path, const clm_feeder_config* cfg, clm_feeder** out) {
    if (!bam_path || !cfg || !out) return fail_open(CLM_E_INVALID, "clm_feeder_open: null argument");
    if (cfg->struct_size != (int32_t)sizeof(clm_feeder_config)) return fail_open(CLM_E_INVALID, "clm_feeder_open: struct_size mismatch");
    if (cfg->batch_size < 1 || cfg->max_tokens < 2 || cfg->slots < 2 || cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world ||
        cfg->inflate_threads < 0 || cfg->inflate_threads > 256)
        return fail_open(CLM_E_INVALID, "clm_feeder_open: bad batch_size / max_tokens / slots / rank / world / inflate_threads");
    clm_feeder* f = new clm_feeder();
    f->cfg = *cfg;
    f->path = bam_path;
    int threads = cfg->inflate_threads;
    if (threads <= 0) {   // every rank inflates the whole file (rank::world selection needs every record): share the cores
        const int cores = (int)std::thread::hardware_concurrency();
        threads = cores / cfg->world - 2;
        threads = threads < 1 ? 1 : threads > 8 ? 8 : threads;
    }
    if (!f->rd.open(bam_path, threads)) {
        const std::string msg = f->rd.err;
        delete f;
        return fail_open(CLM_E_INVALID, msg);
    }
    // magic check up front so that a wrong file fails at open, like the reference's pysam.AlignmentFile
    if (need(f, 4) <= 0 || std::memcmp(f->rd.data(), "BAM\1", 4) != 0) {
        const std::string msg = f->err.empty() ? std::string(bam_path) + ": not a BAM file" : f->err;
        clm_feeder_close(f);
        return fail_open(CLM_E_INVALID, msg);
    }
    const size_t ids_bytes = (size_t)cfg->slots * cfg->batch_size * (size_t)cfg->max_tokens;
    const size_t name_bytes = (size_t)cfg->slots * cfg->batch_size * ID_ROW;
    if (cfg->pinned) {
        void* p = nullptr;
        if (hipHostMalloc(&p, ids_bytes + name_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            clm_feeder_close(f);
            return fail_open(CLM_E_HIP, "clm_feeder_open: hipHostMalloc of the slot ring failed (pinned = 1 needs a HIP device)");
        }
        f->slab_ids = static_cast<uint8_t*>(p);
        f->slab_pinned = true;
    } else {
        f->slab_ids = static_cast<uint8_t*>(std::malloc(ids_bytes + name_bytes));
        if (!f->slab_ids) {
            clm_feeder_close(f);
            return fail_open(CLM_E_INVALID, "clm_feeder_open: out of memory");
        }
    }
    f->slab_names = reinterpret_cast<int8_t*>(f->slab_ids + ids_bytes);
    f->ring.resize((size_t)cfg->slots);
    f->busy.assign((size_t)cfg->slots, 0);
    for (int i = 0; i < cfg->slots; ++i) {
        f->ring[(size_t)i].ids = f->slab_ids + (size_t)i * cfg->batch_size * (size_t)cfg->max_tokens;
        f->ring[(size_t)i].names = f->slab_names + (size_t)i * cfg->batch_size * ID_ROW;
    }
    f->worker = std::thread(decode_loop, f);
    *out = f;
    return CLM_OK;
}

int clm_feeder_next(clm_feeder* f, clm_feed_batch* out) {
    if (!f || !out) return CLM_E_INVALID;
    std::unique_lock<std::mutex> lk(f->mu);
    f->cv.wait(lk, [&] { return f->consumed < f->produced || f->done; });
    if (f->consumed >= f->produced) {
        if (f->failed) return CLM_E_INVALID;
        return 0;
    }
    const int64_t seq = f->consumed++;
    const int slot = (int)(seq % (int64_t)f->ring.size());
    f->busy[(size_t)slot] = 1;
    const Slot& s = f->ring[(size_t)slot];
    out->slot = slot;
    out->n_reads = s.n_reads;
    out->n_tokens = s.n_tokens;
    out->reserved = 0;
    out->row_stride = s.n_tokens;
    out->ids = s.ids;
    out->names = s.names;
    out->first_index = s.first_index;
    return 1;
}

int clm_feeder_release(clm_feeder* f, int32_t slot) {
    if (!f || slot < 0 || slot >= (int32_t)f->ring.size()) return CLM_E_INVALID;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        if (!f->busy[(size_t)slot]) {
            f->err = "clm_feeder_release: slot was not handed out";
            return CLM_E_STATE;
        }
        f->busy[(size_t)slot] = 0;
    }
    f->cv.notify_all();
    return CLM_OK;
}

int clm_feeder_stats(const clm_feeder* f, int64_t* records, int64_t* selected, int64_t* delivered, int64_t* truncated) {
    if (!f) return CLM_E_INVALID;
    if (records) *records = f->n_records.load();
    if (selected) *selected = f->n_selected.load();
    if (delivered) *delivered = f->n_delivered.load();
    if (truncated) *truncated = f->n_truncated.load();
    return CLM_OK;
}

const char* clm_feeder_last_error(const clm_feeder* f) { return f ? f->err.c_str() : g_open_error.c_str(); }

int clm_feeder_close(clm_feeder* f) {
    if (!f) return CLM_OK;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->stop = true;
    }
    f->cv.notify_all();
    if (f->worker.joinable()) f->worker.join();
    if (f->slab_ids) {
        if (f->slab_pinned) (void)hipHostFree(f->slab_ids);
        else std::free(f->slab_ids);
    }
    delete f;
    return CLM_OK;
}

}  // extern "C"
